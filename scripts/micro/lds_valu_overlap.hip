// Microbenchmark: do one wave's LDS round trips overlap another wave's packed-f32 arithmetic on gfx950?
// A block of NW waves; every wave repeats {16 ds_read_b64 -> NPK dependent v_pk_fma_f32 on the 16 values -> 16 ds_write_b64}
// on its own 8 KB of LDS (the shape of one Stockham pass of K2: load a radix-16 butterfly, ~170 packed ops, store it).
// MODE 0: both; 1: LDS only; 2: arithmetic only.  If T(0) ~ T(1) + T(2) the two do not overlap; if ~ max they do.
// build: hipcc --offload-arch=gfx950 -O3 lds_valu_overlap.hip -o lds_valu_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 2000
template <int MODE, int NPK> __global__ void __launch_bounds__(1024) k(float* out, float seed, int skew) {
  extern __shared__ v2f S[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v2f* P = S + wave * 1024;                       // 8 KB per wave: 16 rows x 64 lanes
  for (int r = 0; r < 16; r++) P[r * 64 + lane] = (v2f){seed + r, seed - lane};
  v2f a[16];
#pragma unroll
  for (int r = 0; r < 16; r++) a[r] = (v2f){seed + r, seed};
  const v2f b = {seed * 0.5f, seed * 0.25f};
  __syncthreads();
  if (skew) for (int w = 0; w < wave * skew; w++) asm volatile("s_nop 15");
  for (int it = 0; it < ITER; it++) {
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[r]) : "v"((unsigned)((char*)(P + lane) - (char*)S) + 0u), "n"(r * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < NPK; j++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j % 16]) : "v"(b), "v"(b));
    }
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"((unsigned)((char*)(P + lane) - (char*)S)), "v"(a[r]), "n"(r * 512) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) s += a[r].x + a[r].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NPK> float run(float* d, int nw, int skew) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE, NPK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t shm = (size_t)nw * 8192 > 140 * 1024 ? (size_t)nw * 8192 : 140 * 1024;   // one block per CU, like K2
  hipLaunchKernelGGL((k<MODE, NPK>), dim3(256), dim3(64 * nw), shm, 0, d, 1.0f, skew);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NPK>), dim3(256), dim3(64 * nw), shm, 0, d, 1.0f, skew);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  for (int nw : {4, 8, 12, 16}) {
    const float both = run<0, 170>(d, nw, 0), lds = run<1, 170>(d, nw, 0), alu = run<2, 170>(d, nw, 0), sk = run<0, 170>(d, nw, 7);
    printf("waves/CU %2d: both %.3f ms  LDS only %.3f  arithmetic only %.3f  sum %.3f  max %.3f  (skewed start %.3f) per pass: %.0f ns\n",
           nw, both, lds, alu, lds + alu, lds > alu ? lds : alu, sk, both / ITER * 1e6);
  }
  return 0;
}
