// Identity of the build: version number and the sha256 over every kernel source, header and compile
// flag (computed by __graft_entry__.source_hash() and passed in as DLPD_SOURCE_HASH_STR).  build()
// compares it with the sources it sees -- in the file before loading, through dlpd_source_hash()
// after -- so a prebuilt libdlpd.so that travelled to another box cannot silently differ from them.
#include "dlpd_internal.h"

#ifndef DLPD_SOURCE_HASH_STR
#define DLPD_SOURCE_HASH_STR "0000000000000000000000000000000000000000000000000000000000000000"
#endif

extern "C" {

static const char dlpd_hash_marker[] = "DLPD_SOURCE_HASH=" DLPD_SOURCE_HASH_STR;

int dlpd_version(void) { return 200; }

const char* dlpd_source_hash(void) { return dlpd_hash_marker + sizeof("DLPD_SOURCE_HASH=") - 1; }

}  // extern "C"
