// Microbenchmark: issue rate of scalar f32 add/fma vs packed v_pk_add_f32 / v_pk_fma_f32 on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, float seed) {
  v2f a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = (v2f){seed + i + threadIdx.x, seed - i};
  const v2f b = {seed * 0.5f, seed * 0.25f};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (MODE == 0) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(a[i].x), "v"(b.x)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i].y) : "v"(a[i].y), "v"(b.y)); }
      if (MODE == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (MODE == 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(b.y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.x), "v"(b.y)); }
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(b));
      if (MODE == 4) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(a[i]) : "v"(a[i]), "v"(b));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;           // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double lane_ops = (double)blocks * 256 * ITER * 32;          // f32 results produced
  printf("%-22s waves/SIMD %d: %.3f ms  %.1f T f32-results/s\n", name, waves_per_simd, ms, lane_ops / ms * 1e-9);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4}) {
    run<0>("v_add_f32 x2", d, w); run<1>("v_pk_add_f32", d, w); run<2>("v_fma_f32 x2", d, w); run<3>("v_pk_fma_f32", d, w);
    run<4>("v_pk_add_f32 op_sel", d, w);
  }
  return 0;
}
