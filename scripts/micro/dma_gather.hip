// Microbenchmark: what the memory system gives K3's access pattern on gfx950 -- LDS-DMA gathers of short y-runs
// (P bytes: a tile's rows of ONE (channel, kz) line of Bw[b][ch][kz][x][y]) at a stride of N*N*8 bytes, as a function of
// the run length P, the DMA depth per wave and the number of resident blocks per CU.  Nothing is computed.
// build: hipcc --offload-arch=gfx950 -O3 dma_gather.hip -o dma_gather ; run on the GPU box (scripts/ab_records/gpu_r04_q.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void* g, void* l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)l);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(la), "v"(g) : "memory", "m0");
}
template <int K> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }

// P: bytes per run (64 / 128 / 256 / 1024 = a contiguous stream), D: items in flight per wave
template <int P, int D> __global__ void __launch_bounds__(1024) k_gather(const char* __restrict__ Bw, float* out, int N, int NZ,
                                                                        int CT, int W, int pad_unused) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int LPR = P / 16;                  // lanes per run
  constexpr int RPI = 64 / LPR;                // runs (kz rows) per DMA instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NI = (NZ + RPI - 1) / RPI;         // DMA instructions per item
  const int NYT = (N * 8) / P;                 // tiles along y
  const size_t zstride = (size_t)N * N * 8;    // bytes between kz rows of one channel
  const size_t cstride = zstride * NZ;
  const int plane = blockIdx.x;                // (b, x)
  const int b = plane / N, x = plane % N;
  const char* base = Bw + (size_t)b * CT * cstride + (size_t)x * N * 8;
  unsigned char* mybuf = lds + (size_t)wave * D * NI * 1024;
  const int nitems = NYT * CT;
  auto issue = [&](int j, int buf) {
    const int yt = j / CT, ch = j % CT;
    const char* src = base + (size_t)ch * cstride + (size_t)yt * P;
    int kz = lane / LPR;
    const char* lsrc = src + (size_t)kz * zstride + (size_t)(lane % LPR) * 16;
    for (int it = 0; it < NI; it++) {
      const int k = it * RPI + kz;
      const char* a = (k < NZ) ? lsrc + (size_t)it * RPI * zstride : src + (size_t)(lane % LPR) * 16;
      glds16(a, mybuf + ((size_t)buf * NI + it) * 1024);
    }
  };
  // items of this wave: j = wave, wave + W, ...
  int issued = 0, jn = wave;
  for (; issued < D - 1 && jn < nitems; issued++, jn += W) issue(jn, issued % D);
  int done = 0;
  for (int j = wave; j < nitems; j += W, done++) {
    if (jn < nitems) { issue(jn, issued % D); issued++; jn += W; }
    // all but the newest (issued - done - 1) items have landed: with a full pipeline that is D - 1 items of NI instructions;
    // vmcnt takes an immediate, so the drain at the end simply waits for everything
    if (issued - done == D && D > 1) {
      if (NI == 6) wait_vm<6 * (D - 1)>(); else if (NI == 11) wait_vm<11 * (D - 1) < 63 ? 11 * (D - 1) : 0>();
      else if (NI == 21) wait_vm<21 * (D - 1) < 63 ? 21 * (D - 1) : 0>(); else if (NI == 3) wait_vm<3 * (D - 1)>();
      else if (NI == 5) wait_vm<5 * (D - 1)>(); else if (NI == 9) wait_vm<9 * (D - 1)>(); else if (NI == 17) wait_vm<17 * (D - 1) < 63 ? 17 * (D - 1) : 0>();
      else wait_vm<0>();
    } else wait_vm<0>();
  }
  if (pad_unused == 12345) out[blockIdx.x] = lds[threadIdx.x];
}

static char* buf;
static float* outp;
template <int P, int D> void run(const char* tag, int N, int CT, int nb, int W, int blocks_per_cu) {
  const int NZ = N / 2 + 1, RPI = 64 / (P / 16), NI = (NZ + RPI - 1) / RPI;
  size_t need = (size_t)W * D * NI * 1024;
  // pad the allocation so that exactly `blocks_per_cu` blocks fit into 160 KB
  size_t want = (size_t)160 * 1024 / blocks_per_cu;
  if (want > 64 * 1024 && blocks_per_cu == 2) want = 80 * 1024;
  size_t shmem = need > want ? need : want;
  if (shmem > 160 * 1024) { printf("%-10s P %4d D %d W %2d: needs %zu B of LDS, skipped\n", tag, P, D, W, need); return; }
  hipFuncSetAttribute((const void*)k_gather<P, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = nb * N;
  hipLaunchKernelGGL((k_gather<P, D>), dim3(grid), dim3(64 * W), shmem, 0, buf, outp, N, NZ, CT, W, 0);
  hipEventRecord(e0);
  const int reps = 3;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_gather<P, D>), dim3(grid), dim3(64 * W), shmem, 0, buf, outp, N, NZ, CT, W, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double bytes = (double)nb * CT * NZ * N * N * 8;
  printf("%-10s N %3d CT %2d  run %4d B  depth %d  waves %2d  blocks/CU %d (LDS %3zu KB, in flight/CU %5.1f KB): %.3f ms  %.2f TB/s\n", tag, N,
         CT, P, D, W, (int)(160 * 1024 / shmem), shmem / 1024, (double)(160 * 1024 / shmem) * W * D * NZ * P / 1024.0, ms,
         bytes / ms * 1e-9);
  fflush(stdout);
}

int main() {
  const size_t bytes = (size_t)16 * 49 * 81 * 160 * 160 * 8;   // covers 16 x 17 x 81 x 160^2 and 16 x 49 x 65 x 128^2
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&outp, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  // K3<160> today: 64-byte runs, 5 transform waves x 2 channels = one group in flight, one block per CU
  for (int bpc : {1, 2}) {
    run<64, 1>("N160", 160, 17, 16, 10, bpc);
    run<64, 2>("N160", 160, 17, 16, 10, bpc);
    run<64, 3>("N160", 160, 17, 16, 10, bpc);
    run<128, 1>("N160", 160, 17, 16, 5, bpc);
    run<128, 2>("N160", 160, 17, 16, 5, bpc);
    run<128, 3>("N160", 160, 17, 16, 5, bpc);
    run<128, 1>("N160", 160, 17, 16, 10, bpc);
    run<128, 2>("N160", 160, 17, 16, 10, bpc);
    run<256, 1>("N160", 160, 17, 16, 5, bpc);
    run<256, 2>("N160", 160, 17, 16, 5, bpc);
  }
  // 48 channels x 80^3
  run<64, 1>("c48l80", 160, 49, 16, 10, 1);
  run<64, 2>("c48l80", 160, 49, 16, 10, 1);
  run<128, 1>("c48l80", 160, 49, 16, 5, 1);
  run<128, 2>("c48l80", 160, 49, 16, 5, 1);
  // K3<128>: 128-byte runs, 4 transform waves, two blocks per CU
  for (int bpc : {1, 2}) {
    run<128, 1>("N128", 128, 49, 16, 4, bpc);
    run<128, 2>("N128", 128, 49, 16, 4, bpc);
    run<64, 1>("N128", 128, 49, 16, 8, bpc);
    run<64, 2>("N128", 128, 49, 16, 8, bpc);
    run<256, 1>("N128", 128, 49, 16, 4, bpc);
    run<1024, 1>("N128", 128, 49, 16, 4, bpc);
  }
  return 0;
}
