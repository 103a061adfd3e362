// Internal helpers shared by the dlpd kernel translation units.
#pragma once
#include <dlpd_platform.h>
#include <stdio.h>

#define DLPD_OK 0
#define DLPD_ERR_ARG 1          // null pointer / non-positive size
#define DLPD_ERR_UNSUPPORTED 2  // grid size / hidden width / K outside the compiled set
#define DLPD_ERR_LAUNCH 3       // HIP reported a launch error
#define DLPD_MAX_HIDDEN 64

// hipGetLastError() is sticky per thread and also reports benign codes left behind by the
// caller's own runtime use (e.g. torch's event queries): clear it before each launch, latch a
// failure of any launch of the current entry point, report + reset in dlpd_check_launch().
static inline int& dlpd_launch_failed() { static thread_local int f = 0; return f; }
// TEST HOOK (dlpd_version.hip, dlpd_debug_poison_lds): when switched on, every launch is preceded by a kernel that fills the
// LDS of every CU with NaNs, so a kernel whose result depends on LDS it never wrote stops being "right by accident" (the
// left-over of its own previous block) and fails the same way on every run.  Off: one load and one branch per launch.
extern "C" int dlpd_debug_poison_state(void);
void dlpd_debug_poison_now(hipStream_t st);
#define DLPD_LAUNCH(kern, grid, block, shmem, stream, ...)              \
  do {                                                                  \
    if (dlpd_debug_poison_state()) dlpd_debug_poison_now(stream);       \
    (void)hipGetLastError();                                            \
    DLPD_LAUNCH_RAW(kern, grid, block, shmem, stream, __VA_ARGS__);     \
    const hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess) {                                             \
      dlpd_launch_failed() = 1;                                         \
      fprintf(stderr, "dlpd: launch of %s failed: %s\n", #kern, hipGetErrorString(e_)); \
    }                                                                   \
  } while (0)
static inline int dlpd_check_launch() {
  const int f = dlpd_launch_failed();
  dlpd_launch_failed() = 0;
  return f ? DLPD_ERR_LAUNCH : DLPD_OK;
}

static inline int dlpd_set_max_dyn_shared(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
      return DLPD_ERR_LAUNCH;
  }
  return DLPD_OK;
}

// In-kernel cycle stamps for DIAGNOSTIC builds (-DDLPD_STAMPS=<wave>): per-phase s_memtime deltas of
// one wave per block, summed into a __device__ array named `dlpd_stamps`.  Compiles to nothing in
// the product build; a stamped build is never timed (MI355X guide, "In-kernel stamps").
#ifdef DLPD_STAMPS
#define DLPD_STAMP_DECL unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory")
#define DLPD_STAMP(slot) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
  st_sum[slot] += t_ - st_last; st_last = t_; } while (0)
#define DLPD_STAMP_FLUSH(arr, w) do { if (lane == 0 && wave == (w)) { for (int i_ = 0; i_ < 8; i_++) atomicAdd(&arr[i_], st_sum[i_]); \
  atomicAdd(&arr[15], 1ull); } } while (0)
#else
#define DLPD_STAMP_DECL
#define DLPD_STAMP(slot)
#define DLPD_STAMP_FLUSH(arr, w)
#endif
