// Internal helpers shared by the dlpd kernel translation units.
#pragma once
#include <dlpd_platform.h>

#define DLPD_OK 0
#define DLPD_ERR_ARG 1          // null pointer / non-positive size
#define DLPD_ERR_UNSUPPORTED 2  // grid size / hidden width / K outside the compiled set
#define DLPD_ERR_LAUNCH 3       // HIP reported a launch error
#define DLPD_MAX_HIDDEN 64

// hipGetLastError() is sticky per thread and also reports benign codes left behind by the
// caller's own runtime use (e.g. torch's event queries): clear it before each launch, latch a
// failure of any launch of the current entry point, report + reset in dlpd_check_launch().
static inline int& dlpd_launch_failed() { static thread_local int f = 0; return f; }
#define DLPD_LAUNCH(kern, grid, block, shmem, stream, ...)              \
  do {                                                                  \
    (void)hipGetLastError();                                            \
    DLPD_LAUNCH_RAW(kern, grid, block, shmem, stream, __VA_ARGS__);     \
    if (hipGetLastError() != hipSuccess) dlpd_launch_failed() = 1;      \
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
