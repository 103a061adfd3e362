/* dlpd_debug.h -- TEST HOOKS of libdlpd.so.  No reference counterpart, no part of the search path (include/dlpd.h); the
 * test-suite binds them (tests/test_gpu_parity.py: results must not depend on LDS a kernel never wrote). */
#ifndef DLPD_DEBUG_H
#define DLPD_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* on = 1 makes every kernel launch of this library be preceded by a kernel that fills
 * the LDS of every CU with NaNs (a result that depends on LDS the kernel never wrote then fails on every run instead of
 * once in a while beside another stream); 0 switches it off.  Never on in a timed or production run. */
int dlpd_debug_poison_lds(int on);
/* ... and its self-check: poison, then a kernel that writes nothing to its LDS counts the poisoned words it finds -> per mille
 * (0 .. 1000), -1 on a launch error.  counter8: 8 bytes of device memory. */
int dlpd_debug_poison_selfcheck(void* counter8, void* stream);

#ifdef __cplusplus
}
#endif
#endif
