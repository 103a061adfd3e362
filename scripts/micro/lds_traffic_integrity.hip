// The three-party mix of EXPERIMENTS.md R5 with synthetic kernels that CHECK their own data:
//   victim     (10 waves, 48 KB): an FFT-pass-like exchange through LDS -- every thread stores 8 complex values, barrier, loads 10
//              values other threads stored (8-byte accesses the compiler pairs into ds_write2_b64 / ds_read2_b64), runs a few
//              FMAs on them and compares with what they must be, bit for bit;
//   aggressor 1 (4 waves, 61 KB): ds_read_b128 -> v_mfma_f32_16x16x32_bf16 loop (the convolution's tap-group loop);
//   aggressor 2 (4 waves, 8 KB):  LDS-atomic histogram of a global array + global atomics (the radix select's first pass).
// build: hipcc --offload-arch=gfx950 -O3 lds_traffic_integrity.hip -o lds_traffic_integrity
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float2 val(unsigned round, int pencil, int e) {
  const unsigned h = (round * 2654435761u) ^ ((unsigned)pencil * 40503u) ^ ((unsigned)e * 2246822519u);
  return make_float2((float)(h & 0xFFFF) * (1.0f / 65536.0f), (float)((h >> 16) & 0xFFFF) * (1.0f / 65536.0f));
}

__global__ void __launch_bounds__(640) victim(int nrounds, unsigned long long* errors, unsigned long long* lanes4) {
  extern __shared__ float2 S[];                          // 64 pencils x 93 complex (+ 80), like k_rotate_zfft_cl<80>
  constexpr int RS = 93;
  const int tid = threadIdx.x, p = tid % 64, t = tid / 64, lane = tid & 63;
  unsigned long long bad = 0;
  for (unsigned r = 1; r <= (unsigned)nrounds; r++) {
    float2* P = S + p * RS;
#pragma unroll
    for (int k = 0; k < 8; k++) P[8 * t + k] = val(r, p, 8 * t + k);             // "pass-1 store": outputs 8 t .. 8 t + 7
    __syncthreads();
    if (t < 8) {
      float2 acc = make_float2(0.f, 0.f), want = make_float2(0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 10; k++) {                                              // "pass-2 load": inputs t + 8 k
        const float2 v = P[t + 8 * k], w = val(r, p, t + 8 * k);
        acc.x = __builtin_fmaf(v.x, 1.0f + 0.01f * k, acc.x); acc.y = __builtin_fmaf(v.y, 0.5f + 0.02f * k, acc.y);
        want.x = __builtin_fmaf(w.x, 1.0f + 0.01f * k, want.x); want.y = __builtin_fmaf(w.y, 0.5f + 0.02f * k, want.y);
      }
      if (__float_as_uint(acc.x) != __float_as_uint(want.x) || __float_as_uint(acc.y) != __float_as_uint(want.y)) { bad++; atomicAdd(&lanes4[lane / 16], 1ull); }
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ void __launch_bounds__(256) mfma_loop(float* sink, int loops) {
  extern __shared__ f4 cells[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int v = tid; v < 3840; v += 256) {
    const unsigned h = (unsigned)v * 2654435761u;
    cells[v] = (f4){__uint_as_float(h & 0x3FFF3FFFu), __uint_as_float((h * 3u) & 0x3FFF3FFFu), __uint_as_float((h * 5u) & 0x3FFF3FFFu), __uint_as_float((h * 7u) & 0x3FFF3FFFu)};
  }
  __syncthreads();
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const f4 A = {1.0e-3f * (float)(lane + 1), 2.0e-3f, 3.0e-3f, 4.0e-3f};
  for (int it = 0; it < loops; it++) {
    f4 b[3][4];
#pragma unroll
    for (int sp = 0; sp < 3; sp++)
#pragma unroll
      for (int r = 0; r < 4; r++) b[sp][r] = cells[(sp * 1280 + r * 160 + (lane & 15) + 20 * ((lane >> 4) + (it & 7))) % 3840];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int sp = 0; sp < 3; sp++) {
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, b[sp][r]), acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[(sp + 1) % 3][r]), __builtin_bit_cast(bf16x8, A), acc[r], 0, 0, 0);
      }
  }
  float s = 0;
  for (int r = 0; r < 4; r++) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
  if (s == 12345.678f) sink[tid] = s;
}

__global__ void __launch_bounds__(256) hist_loop(const unsigned* keys, size_t n, unsigned* ghist, int passes) {
  __shared__ unsigned lh[2048];
  for (int pass = 0; pass < passes; pass++) {
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) atomicAdd(&lh[(keys[i] >> (pass * 5)) & 2047u], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) if (lh[i]) atomicAdd(&ghist[i], lh[i]);
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  const int mode = argc > 2 ? atoi(argv[2]) : 3;           // bit 0: mfma aggressor, bit 1: histogram aggressor
  unsigned long long* err; float* sink; unsigned *keys, *gh;
  const size_t nkeys = 1u << 25;
  hipMalloc(&err, 8 * sizeof(unsigned long long)); hipMalloc(&sink, 4096 * 4); hipMalloc(&keys, nkeys * 4); hipMalloc(&gh, 2048 * 4);
  hipMemset(err, 0, 8 * sizeof(unsigned long long)); hipMemset(keys, 0x5A, nkeys * 4); hipMemset(gh, 0, 2048 * 4);
  hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)mfma_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipStream_t s[3];
  for (int i = 0; i < 3; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  for (int r = 0; r < reps; r++) {
    if (mode & 1) hipLaunchKernelGGL(mfma_loop, dim3(2048), dim3(256), 61440, s[1], sink, 1500);
    if (mode & 2) hipLaunchKernelGGL(hist_loop, dim3(1024), dim3(256), 0, s[2], keys, nkeys, gh, 4);
    for (int k = 0; k < 6; k++) hipLaunchKernelGGL(victim, dim3(1280), dim3(640), 48256, s[0], 40, err, err + 1);
  }
  hipDeviceSynchronize();
  unsigned long long h[5];
  hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost);
  printf("mode %d, %d repetitions: exchange results that differ from what they must be: %llu (by lane quarter %llu %llu %llu %llu)\n", mode, reps, h[0], h[1],
         h[2], h[3], h[4]);
  return h[0] ? 1 : 0;
}
