// Identity of the build: version number and the sha256 over every kernel source, header and compile
// flag (computed by __graft_entry__.source_hash() and passed in as DLPD_SOURCE_HASH_STR).  build()
// compares it with the sources it sees -- in the file before loading, through dlpd_source_hash()
// after -- so a prebuilt libdlpd.so that travelled to another box cannot silently differ from them.
#include "dlpd_internal.h"

#ifndef DLPD_SOURCE_HASH_STR
#define DLPD_SOURCE_HASH_STR "0000000000000000000000000000000000000000000000000000000000000000"
#endif

// ---- test hook: LDS poison (see DLPD_LAUNCH) -------------------------------------------------------------------------
static int g_poison_lds = 0;
__global__ void __launch_bounds__(256) k_poison_lds(unsigned pattern, int words) {
  DLPD_DYN_SHARED(unsigned, S);
  for (int i = threadIdx.x; i < words; i += 256) S[i] = pattern;
  __syncthreads();
  if (S[(threadIdx.x * 7919) % words] != pattern) S[0] = 0;          // (keeps the stores alive)
}
void dlpd_debug_poison_now(hipStream_t st) {
  const int bytes = 160 * 1024;                                       // the whole LDS of a CU: one block per CU at a time
  if (dlpd_set_max_dyn_shared((const void*)k_poison_lds, bytes)) return;
  DLPD_LAUNCH_RAW(k_poison_lds, dim3(2048), dim3(256), bytes, st, 0x7FC00000u | 0x00012345u, bytes / 4);   // quiet NaNs
}

// self-check of the hook: blocks that WRITE NOTHING to their LDS count how many of its words hold the poison pattern
__global__ void __launch_bounds__(256) k_count_poison(unsigned pattern, int words, unsigned long long* hits) {
  DLPD_DYN_SHARED(unsigned, S);
  unsigned n = 0;
  for (int i = threadIdx.x; i < words; i += 256) n += (S[i] == pattern) ? 1u : 0u;
  if (n) atomicAdd(hits, (unsigned long long)n);
}

extern "C" {

// -> per mille of the LDS words that a kernel launched right after the poison finds poisoned (0 .. 1000), or -1
int dlpd_debug_poison_selfcheck(void* counter8, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int bytes = 64 * 1024, nblk = 1024;
  if (!counter8 || hipMemsetAsync(counter8, 0, 8, st) != hipSuccess) return -1;
  (void)hipGetLastError();
  dlpd_debug_poison_now(st);
  if (hipGetLastError() != hipSuccess) return -1;
  if (dlpd_set_max_dyn_shared((const void*)k_count_poison, bytes)) return -1;
  DLPD_LAUNCH_RAW(k_count_poison, dim3(nblk), dim3(256), bytes, st, 0x7FC00000u | 0x00012345u, bytes / 4, (unsigned long long*)counter8);
  if (hipGetLastError() != hipSuccess) return -1;
  unsigned long long h = 0;
  if (hipMemcpyAsync(&h, counter8, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
  return (int)(h * 1000ull / ((unsigned long long)nblk * (bytes / 4)));
}

int dlpd_debug_poison_state(void) { return g_poison_lds; }
// on = 1: poison the LDS before every kernel launch of this library (tests only; never timed), 0: off
int dlpd_debug_poison_lds(int on) {
  g_poison_lds = on ? 1 : 0;
  return 0;
}

static const char dlpd_hash_marker[] = "DLPD_SOURCE_HASH=" DLPD_SOURCE_HASH_STR;

int dlpd_version(void) { return 200; }

const char* dlpd_source_hash(void) { return dlpd_hash_marker + sizeof("DLPD_SOURCE_HASH=") - 1; }

}  // extern "C"
