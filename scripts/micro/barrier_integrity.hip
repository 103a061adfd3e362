// Do workgroup barriers hold when workgroups of several shapes (10 / 4 / 4 / 16 waves, 48 / 61 / 8 / 33 KB of LDS) share a CU?
// (EXPERIMENTS.md R5.)  Every round each wave publishes (round, wave) in LDS, the block meets at a barrier, and every wave
// checks that ALL waves of its block have published THIS round; a wave that was let through early sees a stale round.
// Lanes write their slot in four quarters (lane / 16) with arithmetic in between, so that a wave released early is caught with
// only some quarters of a slower wave's stores landed -- the signature of the perturbation in question (lanes 48-63).
// build: hipcc --offload-arch=gfx950 -O3 barrier_integrity.hip -o barrier_integrity
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NT> __global__ void __launch_bounds__(NT) rounds(int nrounds, int lds_words, unsigned long long* errors, float* sink) {
  extern __shared__ unsigned slot[];                       // slot[wave * 64 + lane] = round, the rest of the LDS is ballast
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = NT / 64;
  unsigned long long bad = 0;
  float x = 1.0f + 0.001f * (float)tid;
  for (int i = tid + nw * 64; i < lds_words; i += NT) slot[i] = 0xDEADBEEFu;
  for (unsigned r = 1; r <= (unsigned)nrounds; r++) {
    for (int k = 0; k < 8 + (wave & 3) * 6; k++) x = x * 1.0001f + 0.5f;            // waves of a block arrive at different times
    slot[wave * 64 + lane] = r;
    __syncthreads();
    for (int w = 0; w < nw; w++)
      if (slot[w * 64 + lane] != r) bad++;
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
  if (x == 12345.0f) sink[tid] = x;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  unsigned long long* err; float* sink;
  hipMalloc(&err, 4 * sizeof(unsigned long long)); hipMalloc(&sink, 4096 * 4);
  hipMemset(err, 0, 4 * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)rounds<640>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)rounds<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)rounds<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipStream_t s[4];
  for (int i = 0; i < 4; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  for (int r = 0; r < reps; r++) {
    hipLaunchKernelGGL(rounds<640>, dim3(1536), dim3(640), 48256, s[0], 60, 48256 / 4, err + 0, sink);
    hipLaunchKernelGGL(rounds<256>, dim3(2048), dim3(256), 61952, s[1], 60, 61952 / 4, err + 1, sink);
    hipLaunchKernelGGL(rounds<256>, dim3(1024), dim3(256), 8192, s[2], 200, 8192 / 4, err + 2, sink);
    hipLaunchKernelGGL(rounds<1024>, dim3(256), dim3(1024), 32776, s[3], 200, 32776 / 4, err + 3, sink);
  }
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost);
  printf("stale rounds seen after a barrier, %d repetitions of four kernels side by side: 10-wave / 48 KB blocks %llu, 4-wave / 61 KB %llu, "
         "4-wave / 8 KB %llu, 16-wave / 33 KB %llu\n", reps, h[0], h[1], h[2], h[3]);
  return (h[0] | h[1] | h[2] | h[3]) ? 1 : 0;
}
