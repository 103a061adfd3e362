// Platform glue for the gfx950 build: HIP runtime + launch macro.
// (tests/emu/ carries a same-named header that runs the identical kernel sources as
//  cooperative fibers on the CPU -- a debugging/test harness, never part of the product.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DLPD_HD __host__ __device__ __forceinline__
#define DLPD_D __device__ __forceinline__
// kern must be parenthesised when it is a template-id: DLPD_LAUNCH((k<A,B>), grid, block, ...)
#define DLPD_LAUNCH_RAW(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
// dynamic LDS, 16-byte aligned base (guide G17)
#define DLPD_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[]; \
  type* name = reinterpret_cast<type*>(name##_raw_)
