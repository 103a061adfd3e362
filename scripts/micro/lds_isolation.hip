// Is a workgroup's LDS really its own when workgroups of several LDS sizes share a CU?  (EXPERIMENTS.md R5: three kernels
// -- 48 KB FFT blocks, 61 KB convolution blocks, 8-33 KB top-K blocks -- side by side corrupted the FFT blocks' results.)
// Every block writes a signature into ALL of its LDS words, then re-reads them again and again (with sleeps and barriers in
// between) and counts the words that changed under it.  Kernels with the LDS footprints and block sizes of the real three run
// on three streams at once.   build: hipcc --offload-arch=gfx950 -O3 lds_isolation.hip -o lds_isolation
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NT> __global__ void __launch_bounds__(NT) guard(unsigned salt, int words, int spins, unsigned long long* errors, int atomics) {
  extern __shared__ unsigned w[];
  const unsigned sig = (blockIdx.x * 2654435761u) ^ salt;
  for (int i = threadIdx.x; i < words; i += NT) w[i] = sig ^ (unsigned)i;
  __syncthreads();
  unsigned long long bad = 0;
  for (int r = 0; r < spins; r++) {
    for (int i = threadIdx.x; i < words; i += NT)
      if (w[i] != (sig ^ (unsigned)i)) { bad++; w[i] = sig ^ (unsigned)i; }
    __syncthreads();                                  // (every wave has finished checking before anybody touches a word)
    if (atomics) {                                    // the top-K kernels' habit: LDS atomics on a small table, undone again
      atomicAdd(&w[(threadIdx.x * 7 + r) % words], 1u);
      __syncthreads();
      atomicSub(&w[(threadIdx.x * 7 + r) % words], 1u);
    }
    __builtin_amdgcn_s_sleep(20);
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  unsigned long long* err;
  hipMalloc(&err, 4 * sizeof(unsigned long long));
  hipMemset(err, 0, 4 * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)guard<640>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)guard<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipStream_t s[4];
  for (int i = 0; i < 4; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  for (int r = 0; r < rounds; r++) {
    hipLaunchKernelGGL(guard<640>, dim3(1536), dim3(640), 48256, s[0], 0x11111111u + r, 48256 / 4, 40, err + 0, 0);   // k_rotate_zfft_cl<80>
    hipLaunchKernelGGL(guard<256>, dim3(2048), dim3(256), 61952, s[1], 0x22222222u + r, 61952 / 4, 40, err + 1, 0);   // k_conv3d_bf16x3<5,..>
    hipLaunchKernelGGL(guard<256>, dim3(1024), dim3(256), 8192, s[2], 0x33333333u + r, 8192 / 4, 120, err + 2, 1);    // k_topk_hist
    hipLaunchKernelGGL(guard<256>, dim3(256), dim3(256), 32776, s[3], 0x44444444u + r, 32776 / 4, 120, err + 3, 1);   // k_topk_sort
  }
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost);
  printf("LDS words that changed under their owner after %d rounds of four kernels side by side: 48 KB blocks %llu, 61 KB blocks %llu, "
         "8 KB blocks (LDS atomics) %llu, 33 KB blocks (LDS atomics) %llu\n", rounds, h[0], h[1], h[2], h[3]);
  return (h[0] | h[1] | h[2] | h[3]) ? 1 : 0;
}
