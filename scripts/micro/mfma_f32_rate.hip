// Issue rate of v_mfma_f32_16x16x4_f32 (and 32x32x2) on one SIMD, alone and with two waves per SIMD, and beside a wave of
// vector FMAs (round 6: K3's filter role on the f32 matrix core came out matrix-bound at about twice the cycles per
// instruction the tables give -- is it the instruction or the kernel?).
//   build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITER 2048

// MODE 0: 16 independent 16x16x4 accumulators; 1: 4 independent 32x32x2 accumulators; 2: 64 independent packed FMAs (no matrix work)
// MIX: odd waves of the block run MODE 2 instead (a vector wave beside a matrix wave on every SIMD when the block has 8 waves)
template <int MODE, bool MIX> __global__ void __launch_bounds__(512) k(float* out, float seed, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vec = MODE == 2 || (MIX && (wave >= 4));
  float a = seed + lane, b = seed * 0.5f - lane;
  f4 acc[16];
  f16v big[4];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f va[32];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (f4){seed, a, b, 1.f};
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) big[i][j] = seed + j;
#pragma unroll
  for (int i = 0; i < 32; i++) va[i] = (v2f){seed + i, a};
  const long long t0 = __builtin_readcyclecounter();
  if (vec) {
    const v2f vb = {b, a};
    for (int it = 0; it < ITER; it++)
#pragma unroll
      for (int i = 0; i < 32; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(va[i]) : "v"(vb), "v"(vb));
  } else if (MODE == 0) {
    for (int it = 0; it < ITER; it++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  } else {
    for (int it = 0; it < ITER; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 4; i++) s += big[i][0] + big[i][15];
#pragma unroll
  for (int i = 0; i < 32; i++) s += va[i].x + va[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE, bool MIX> static void run(const char* name, int threads, float* d, long long* dc, int per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256;                         // one block per CU
  hipLaunchKernelGGL((k<MODE, MIX>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f, dc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, MIX>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long c[8] = {0};
  hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-58s %7.3f ms   s_memtime ticks per instruction: wave 0 %.1f", name, ms, (double)c[0] / (ITER * per_iter));
  if (threads > 256) printf(", wave 4 %.1f", (double)c[4] / (ITER * (MIX ? 32 : per_iter)));
  printf("\n");
}

int main() {
  float* d;
  long long* dc;
  hipMalloc(&d, 256 * 512 * 4);
  hipMalloc(&dc, 64);
  run<0, false>("16x16x4 f32, one wave per SIMD", 256, d, dc, 16);
  run<0, false>("16x16x4 f32, two waves per SIMD", 512, d, dc, 16);
  run<1, false>("32x32x2 f32, one wave per SIMD", 256, d, dc, 4);
  run<1, false>("32x32x2 f32, two waves per SIMD", 512, d, dc, 4);
  run<2, false>("v_pk_fma_f32, one wave per SIMD", 256, d, dc, 32);
  run<2, false>("v_pk_fma_f32, two waves per SIMD", 512, d, dc, 32);
  run<0, true>("16x16x4 f32 (waves 0-3) beside v_pk_fma_f32 (waves 4-7)", 512, d, dc, 16);
  return 0;
}
