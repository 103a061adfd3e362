// Does a wave that feeds v_mfma_f32_16x16x32_bf16 from ds_read_b128 disturb the vector arithmetic of OTHER waves on its CU?
// (EXPERIMENTS.md R5: the plugin's bf16 x 3 convolution beside the coarse grid's K1 changed the low mantissa bits of lanes 48-63.)
//   victim:    every lane runs a long chain of dependent f32 FMAs / multiplies / adds (one variant each) from lane-dependent
//              seeds and stores the result; run alone it gives the reference bits.
//   aggressor: blocks that fill 60 KB of LDS once and then loop { 12 x ds_read_b128 -> 24 x v_mfma_f32_16x16x32_bf16 } (the tap-group
//              loop of k_conv3d_bf16x3), on a second stream; variants: with or without writing the LDS first.
// The victim is relaunched many times beside the aggressor; every result is compared with the reference, and the lanes
// (mod 64) of the mismatches are tallied.   build: hipcc --offload-arch=gfx950 -O3 mfma_valu_interference.hip -o mfma_valu_interference
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OP> __global__ void __launch_bounds__(640) victim(float* out, int iters, float a, float b) {
  // 10 waves per block and 48 KB of LDS: the footprint of k_rotate_zfft_cl<80>
  extern __shared__ float lds[];
  const int tid = threadIdx.x, gid = blockIdx.x * blockDim.x + tid;
  float x = 0.001f * (float)(gid % 977) + 0.5f, y = 1.0f + 0.0001f * (float)(tid % 61);
  lds[tid] = x;
  __syncthreads();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) x = __builtin_fmaf(x, a, y * b);                 // fma + mul
    if (OP == 1) { x = x * a; x = x + b; }                        // mul, add
    if (OP == 2) { x = lds[(tid + i) % 640] * a + x * b; }        // LDS reads feeding the arithmetic, as a transform pass does
    y = y * 0.99999f + 0.00001f;
  }
  out[gid] = x + y;
}

template <bool WRITE_LDS> __global__ void __launch_bounds__(256) aggressor(float* sink, int loops) {
  extern __shared__ f4 cells[];                                    // 3840 cells of 16 bytes = 60 KB
  const int tid = threadIdx.x, lane = tid & 63;
  if (WRITE_LDS)
    for (int v = tid; v < 3840; v += 256) {
      const unsigned h = (unsigned)v * 2654435761u;
      cells[v] = (f4){__uint_as_float(h & 0x3FFF3FFFu), __uint_as_float((h * 3u) & 0x3FFF3FFFu), __uint_as_float((h * 5u) & 0x3FFF3FFFu),
                      __uint_as_float((h * 7u) & 0x3FFF3FFFu)};
    }
  __syncthreads();
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const f4 A = {1.0e-3f * (float)(lane + 1), 2.0e-3f, 3.0e-3f, 4.0e-3f};
  for (int it = 0; it < loops; it++) {
    f4 bfr[3][4];
#pragma unroll
    for (int sp = 0; sp < 3; sp++)
#pragma unroll
      for (int r = 0; r < 4; r++) bfr[sp][r] = cells[(sp * 1280 + r * 160 + (lane & 15) + 20 * ((lane >> 4) + (it & 7))) % 3840];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int sp = 0; sp < 3; sp++) {
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, bfr[sp][r]), acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[(sp + 1) % 3][r]), __builtin_bit_cast(bf16x8, A), acc[r], 0, 0, 0);
      }
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
  if (s == 12345.678f) sink[tid] = s;
}

template <int OP, bool WRITE_LDS> static void run_case(const char* name, int reps) {
  const int vblocks = 768, vthreads = 640, n = vblocks * vthreads, iters = 3000;
  float *out, *sink;
  hipMalloc(&out, n * 4); hipMalloc(&sink, 4096);
  std::vector<float> ref(n), got(n);
  hipFuncSetAttribute((const void*)aggressor<WRITE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
  hipStream_t sv, sa;
  hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipLaunchKernelGGL((victim<OP>), dim3(vblocks), dim3(vthreads), 48256, sv, out, iters, 0.999f, 0.0007f);
  hipStreamSynchronize(sv);
  hipMemcpy(ref.data(), out, n * 4, hipMemcpyDeviceToHost);
  // alone again: must be the same bits
  hipLaunchKernelGGL((victim<OP>), dim3(vblocks), dim3(vthreads), 48256, sv, out, iters, 0.999f, 0.0007f);
  hipStreamSynchronize(sv);
  hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost);
  const bool alone_ok = memcmp(ref.data(), got.data(), n * 4) == 0;
  long bad_runs = 0, bad_vals = 0, lanes[64] = {0};
  double worst = 0;
  for (int rep = 0; rep < reps; rep++) {
    hipLaunchKernelGGL((aggressor<WRITE_LDS>), dim3(4096), dim3(256), 61440, sa, sink, 6000);
    for (int k = 0; k < 12; k++) {
      hipLaunchKernelGGL((victim<OP>), dim3(vblocks), dim3(vthreads), 48256, sv, out, iters, 0.999f, 0.0007f);
      hipStreamSynchronize(sv);
      hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost);
      long b = 0;
      for (int i = 0; i < n; i++)
        if (memcmp(&got[i], &ref[i], 4) != 0) {
          b++; lanes[i % 64]++;
          const double e = fabs((double)got[i] - (double)ref[i]) / fabs((double)ref[i]);
          if (e > worst) worst = e;
        }
      bad_vals += b; bad_runs += b ? 1 : 0;
    }
    hipStreamSynchronize(sa);
  }
  long q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; l++) q[l / 16] += lanes[l];
  printf("%-58s alone identical %d | beside the aggressor: %ld of %d launches differ, %ld values, worst rel. error %.2g, by lane quarter %ld %ld %ld %ld\n",
         name, (int)alone_ok, bad_runs, reps * 12, bad_vals, worst, q[0], q[1], q[2], q[3]);
  hipFree(out); hipFree(sink);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  run_case<0, true>("victim fma+mul chain, aggressor writes its LDS", reps);
  run_case<0, false>("victim fma+mul chain, aggressor reads unwritten LDS", reps);
  run_case<1, false>("victim mul, add chain, aggressor reads unwritten LDS", reps);
  run_case<2, true>("victim LDS-fed arithmetic, aggressor writes its LDS", reps);
  run_case<2, false>("victim LDS-fed arithmetic, aggressor reads unwritten LDS", reps);
  return 0;
}
