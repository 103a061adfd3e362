// What does rocprofv3's FETCH_SIZE count for the LOAD SHAPES of K2 (round-5 verdict, weak 3)?  MI355X_MICROARCH.md: on gfx950 a wide
// coalesced streaming read (16 B per lane) is tallied at HALF its bytes; other shapes are uncalibrated.  K2<128> issues two
// kinds of global loads (dlpd_k2.hip): its A rows as 16 bytes per lane, 64 lanes contiguous (1 KB per wave instruction), and
// the receptor values as 8 bytes per lane where the 8 lanes of a row read a contiguous 64-byte run and the 8 rows of the
// instruction lie N * 8 bytes apart.  Each kernel below reads a buffer of known size exactly once in one of those shapes
// (4 GiB: no reuse out of L2 or the 256 MiB Infinity Cache); run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- scripts/micro/fetch_size_shapes
// FETCH_SIZE(kernel) / bytes read = the factor to divide by (scripts/micro/fetch_size_shapes.sh prints it).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_size_shapes.hip -o scripts/micro/fetch_size_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// 16 bytes per lane, a wave reads 1 KB contiguous (K2's A rows, K3's spectra, every "wide" stream)
__global__ void __launch_bounds__(256) k_wide16(const float4* __restrict__ p, size_t n16, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p) + i);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1.2345f) *sink = acc;
}
// 8 bytes per lane; lane = 8 * row + col: 8 lanes of a row read one 64-byte run, the 8 rows are `stride8` 8-byte elements
// apart (K2's receptor loads: rbase[out_index * N + col], N = 128 -> stride 128 elements = 1 KB)
__global__ void __launch_bounds__(256) k_runs64_8B(const float2* __restrict__ p, size_t n8, int stride8, float* sink) {
  // the buffer is cut into tiles of 8 rows x stride8 elements; a wave walks a tile's 64-byte column blocks
  const size_t tile = (size_t)8 * stride8, ntiles = n8 / tile;
  const int lane = threadIdx.x & 63, row = lane >> 3, col = lane & 7;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  for (size_t t = wave; t < ntiles; t += nwaves)
    for (int cb = 0; cb < stride8 / 8; cb++) {
      const f2v v = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p) + t * tile + (size_t)row * stride8 + cb * 8 + col);
      acc += v.x + v.y;
    }
  if (acc == 1.2345f) *sink = acc;
}
// 16 bytes per lane, 4 lanes a 64-byte run, runs scattered (the channels-last K1's gather shape, K3<160>'s 64-byte pieces)
__global__ void __launch_bounds__(256) k_runs64_16B(const float4* __restrict__ p, size_t n16, float* sink) {
  const int lane = threadIdx.x & 63, grp = lane >> 2, q = lane & 3;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t nrun = n16 / 4, per = nrun / 16;             // 16 run streams, 4 MB apart at least
  float acc = 0.f;
  for (size_t r = wave; r < per; r += nwaves) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p) + ((size_t)grp * per + r) * 4 + q);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1.2345f) *sink = acc;
}

int main() {
  const size_t bytes = (size_t)4 << 30;
  void* buf;
  float* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, bytes));
  CK(hipDeviceSynchronize());
  const int nblk = 256 * 16;
  for (int rep = 0; rep < 3; rep++) {
    k_wide16<<<nblk, 256>>>((const float4*)buf, bytes / 16, sink);
    k_runs64_8B<<<nblk, 256>>>((const float2*)buf, bytes / 8, 128, sink);
    k_runs64_16B<<<nblk, 256>>>((const float4*)buf, bytes / 16, sink);
  }
  CK(hipDeviceSynchronize());
  printf("bytes read per launch of every kernel: %zu\n", bytes);
  return 0;
}
