// CPU emulation of the small HIP subset the dlpd kernels use.  TEST HARNESS ONLY.
//
// The kernel sources under deeplocalproteindocking_amd/csrc/ are compiled unchanged with g++
// against this header (it shadows csrc/dlpd_platform.h through the include path).  Every
// thread of a block is a ucontext fiber; __syncthreads() and the wave collectives are
// scheduling points, so the kernels' real barrier structure is exercised.  "Device" pointers
// are plain host pointers.  This lets the -m "not gpu" tests check kernel index logic without a
// GPU; it is never loaded by the product package.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <ucontext.h>
#include <sys/mman.h>
#include <vector>
#include <functional>
#include <algorithm>

#define DLPD_CPU_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define DLPD_HD inline
#define DLPD_D inline

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline float2 make_float2(float a, float b) { float2 r = {a, b}; return r; }
static inline float4 make_float4(float a, float b, float c, float d) { float4 r = {a, b, c, d}; return r; }
static inline int2 make_int2(int a, int b) { int2 r = {a, b}; return r; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
#define hipMemcpyDeviceToDevice 0
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipMemcpyDeviceToHost 2

namespace emu {
struct Fiber {
  ucontext_t ctx;
  void* stack;
  int state;  // 0 ready, 1 at barrier, 2 at wave sync, 3 done
};
struct State {
  uint3 threadIdx, blockIdx;
  dim3 blockDim, gridDim;
  ucontext_t sched;
  std::vector<Fiber> fibers;
  int cur;
  int live, bar_arrived;
  int wave_live[16], wave_arrived[16];
  uint64_t wave_buf[16][64];
  unsigned char wave_wide[16][64][32];
  unsigned char* dyn_smem;
  std::function<void()> body;
};
inline State& S() { static State s; return s; }
static const size_t kStack = 512 * 1024;

inline void yield_to_sched() {
  State& s = S();
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void barrier() {
  State& s = S();
  s.fibers[s.cur].state = 1;
  s.bar_arrived++;
  yield_to_sched();
}
inline void wave_sync() {
  State& s = S();
  s.fibers[s.cur].state = 2;
  s.wave_arrived[s.cur / 64]++;
  yield_to_sched();
}
inline void trampoline() {
  State& s = S();
  s.body();
  s.fibers[s.cur].state = 3;
  s.live--;
  s.wave_live[s.cur / 64]--;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void set_tid(int t) {
  State& s = S();
  s.threadIdx.x = t % s.blockDim.x;
  s.threadIdx.y = (t / s.blockDim.x) % s.blockDim.y;
  s.threadIdx.z = t / (s.blockDim.x * s.blockDim.y);
}
inline void run_block(int nthreads) {
  State& s = S();
  if ((int)s.fibers.size() < nthreads) {
    size_t old = s.fibers.size();
    s.fibers.resize(nthreads);
    for (size_t i = old; i < (size_t)nthreads; i++)
      s.fibers[i].stack = mmap(0, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  }
  s.live = nthreads;
  s.bar_arrived = 0;
  int nw = (nthreads + 63) / 64;
  for (int w = 0; w < nw; w++) { s.wave_live[w] = std::min(64, nthreads - 64 * w); s.wave_arrived[w] = 0; }
  for (int t = 0; t < nthreads; t++) {
    Fiber& f = s.fibers[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
    f.state = 0;
  }
  while (s.live > 0) {
    bool progressed = false;
    for (int t = 0; t < nthreads; t++) {
      Fiber& f = s.fibers[t];
      if (f.state != 0) continue;
      s.cur = t;
      set_tid(t);
      swapcontext(&s.sched, &f.ctx);
      progressed = true;
    }
    // releases
    for (int w = 0; w < nw; w++) {
      if (s.wave_live[w] > 0 && s.wave_arrived[w] == s.wave_live[w]) {
        for (int t = 64 * w; t < std::min(nthreads, 64 * w + 64); t++)
          if (s.fibers[t].state == 2) s.fibers[t].state = 0;
        s.wave_arrived[w] = 0;
        progressed = true;
      }
    }
    if (s.live > 0 && s.bar_arrived == s.live) {
      for (int t = 0; t < nthreads; t++)
        if (s.fibers[t].state == 1) s.fibers[t].state = 0;
      s.bar_arrived = 0;
      progressed = true;
    }
    if (!progressed && s.live > 0) {
      fprintf(stderr, "emu: deadlock (divergent barrier) in block (%u,%u,%u)\n", s.blockIdx.x, s.blockIdx.y, s.blockIdx.z);
      abort();
    }
  }
}
inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  State& s = S();
  s.body = body;
  s.gridDim = grid;
  s.blockDim = block;
  int nthreads = block.x * block.y * block.z;
  std::vector<unsigned char> sm(shmem + 64);
  s.dyn_smem = (unsigned char*)(((uintptr_t)sm.data() + 63) & ~(uintptr_t)63);
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        s.blockIdx.x = bx; s.blockIdx.y = by; s.blockIdx.z = bz;
        memset(s.dyn_smem, 0xCD, shmem);   // poison: catch reads of unwritten LDS
        run_block(nthreads);
      }
}
template <typename T> inline T shfl_generic(T v, int src_lane_of_me /* lane to read */) {
  State& s = S();
  int w = s.cur / 64, lane = s.cur % 64;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  s.wave_buf[w][lane] = bits;
  wave_sync();
  T r;
  uint64_t b = s.wave_buf[w][src_lane_of_me & 63];
  memcpy(&r, &b, sizeof(T));
  wave_sync();
  return r;
}
}  // namespace emu

#define threadIdx (emu::S().threadIdx)
#define blockIdx (emu::S().blockIdx)
#define blockDim (emu::S().blockDim)
#define gridDim (emu::S().gridDim)
#define __syncthreads() emu::barrier()
#define DLPD_LAUNCH_RAW(kern, grid, block, shmem, stream, ...) \
  emu::launch(grid, block, shmem, [=]() { kern(__VA_ARGS__); })
#define DLPD_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu::S().dyn_smem)

template <typename T> static inline T __shfl_xor(T v, int m) { return emu::shfl_generic(v, (emu::S().cur % 64) ^ m); }
template <typename T> static inline T __shfl_down(T v, int d) { int l = emu::S().cur % 64; return emu::shfl_generic(v, l + d < 64 ? l + d : l); }
template <typename T> static inline T __shfl(T v, int src) { return emu::shfl_generic(v, src); }
static inline unsigned long long __ballot(int pred) {
  unsigned long long m = 0;
  int lane = emu::S().cur % 64;
  for (int i = 0; i < 64; i++) { int p = emu::shfl_generic(pred, i); if (p) m |= (1ull << i); }
  (void)lane;
  return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// (the fibers of a wave run one after the other here: a plain increment IS the wave's combined count; the ballot rounds of
//  the device version would cost three wave collectives per voxel and round)
static inline void dlpd_lds_count(unsigned* table, unsigned bin, bool hit, int) { if (hit) table[bin] += 1u; }

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
#define DLPD_GLDS16(g, l) memcpy(reinterpret_cast<char*>(l) + 16 * (emu::S().cur % 64), (const void*)(g), 16)
#define DLPD_GLDS16_SO(base, voff, l) DLPD_GLDS16(reinterpret_cast<const char*>(base) + (voff), l)
typedef char* dlpd_lds_t;
#define DLPD_LDS_ADDR(p) reinterpret_cast<char*>(p)
#define DLPD_GLDS16_SOA(base, voff, la) DLPD_GLDS16(reinterpret_cast<const char*>(base) + (voff), (la))
#define DLPD_LDS_BARRIER() emu::barrier()
#define DLPD_WAIT_VMEM() ((void)0)
#define DLPD_WAVE_SYNC() emu::wave_sync()
#define DLPD_WAIT_LDS() ((void)0)
struct dlpd_pair_t { float x, y; };
#define DLPD_PAIR dlpd_pair_t
static inline dlpd_pair_t dlpd_load_pair(const float* p) { dlpd_pair_t r; r.x = p[0]; r.y = p[1]; return r; }
static inline void dlpd_store_stream_c(float2* p, float2 v) { *p = v; }
static inline float2 dlpd_load_stream_c(const float2* p) { return *p; }
#define DLPD_LOAD_STREAM(p) (*(p))
#define DLPD_STORE_STREAM(p, v) (*(p) = (v))
#define DLPD_CLAMP(v, c) fminf(fmaxf((v), -(c)), (c))
#define DLPD_SCHED_FENCE() ((void)0)
#define DLPD_OPAQUE_V(x) ((void)0)
#define DLPD_UNIFORM(x) (x)
// uses of a scalar / vector register value that emit nothing (the compiler must have the value at this point)
#define DLPD_SINK_S(x) ((void)(x))
#define DLPD_SINK_V(x) ((void)(x))
#define DLPD_OPAQUE_S(x) ((void)0)
#define DLPD_OPAQUE(x) ((void)(x))
#define DLPD_SET_PRIO(n) ((void)0)
struct dlpd_f2v { float x, y; };
static inline dlpd_f2v dlpd_f2_make(float a, float b) { dlpd_f2v r = {a, b}; return r; }
static inline dlpd_f2v dlpd_f2_splat(float a) { dlpd_f2v r = {a, a}; return r; }
static inline float dlpd_f2_get(dlpd_f2v v, int i) { return i ? v.y : v.x; }
static inline dlpd_f2v dlpd_pk_fma(dlpd_f2v a, dlpd_f2v b, dlpd_f2v c) { dlpd_f2v r = {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; return r; }

// f32 MFMA 16x16x4 as a wave collective (same lane maps and k-ordered fmaf chain as the hardware)
struct dlpd_acc4 { float v[4]; };
static inline dlpd_acc4 dlpd_acc4_zero() { dlpd_acc4 z = {{0.f, 0.f, 0.f, 0.f}}; return z; }
static inline float dlpd_acc4_get(const dlpd_acc4& a, int j) { return a.v[j]; }
static inline dlpd_acc4 dlpd_acc4_make(float a, float b, float c, float d) { dlpd_acc4 z = {{a, b, c, d}}; return z; }
static inline dlpd_acc4 dlpd_emu_mfma_16x16x4(float a, float b, dlpd_acc4 acc) {
  emu::State& s = emu::S();
  const int w = s.cur / 64, l = s.cur % 64;
  float ab[2] = {a, b};
  memcpy(&s.wave_buf[w][l], ab, 8);
  emu::wave_sync();
  for (int j = 0; j < 4; j++) {
    const int row = 4 * (l >> 4) + j, col = l & 15;
    for (int k = 0; k < 4; k++) {
      float fa[2], fb[2];
      memcpy(fa, &s.wave_buf[w][row + 16 * k], 8);
      memcpy(fb, &s.wave_buf[w][col + 16 * k], 8);
      acc.v[j] = fmaf(fa[0], fb[1], acc.v[j]);
    }
  }
  emu::wave_sync();
  return acc;
}
#define DLPD_MFMA_16x16x4(a, b, acc) dlpd_emu_mfma_16x16x4((a), (b), (acc))
// bf16 MFMA 16x16x32 as a wave collective: lane l holds A[l&15][8*(l>>4) + j] / B[8*(l>>4) + j][l&15] (8 bf16 = 16 bytes each);
// exact products, summed in double and rounded once (the hardware's internal order is not specified; tests use tolerances)
static inline unsigned dlpd_f2bf(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
static inline float dlpd_bf2f(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline dlpd_acc4 dlpd_emu_mfma_16x16x32_bf16(float4 a, float4 b, dlpd_acc4 acc) {
  emu::State& s = emu::S();
  const int w = s.cur / 64, l = s.cur % 64;
  memcpy(&s.wave_wide[w][l][0], &a, 16);
  memcpy(&s.wave_wide[w][l][16], &b, 16);
  emu::wave_sync();
  for (int j = 0; j < 4; j++) {
    const int row = 4 * (l >> 4) + j, col = l & 15;
    double sum = acc.v[j];
    for (int k = 0; k < 32; k++) {
      unsigned short ha, hb;
      memcpy(&ha, &s.wave_wide[w][row + 16 * (k >> 3)][2 * (k & 7)], 2);
      memcpy(&hb, &s.wave_wide[w][col + 16 * (k >> 3)][16 + 2 * (k & 7)], 2);
      sum += (double)dlpd_bf2f(ha) * (double)dlpd_bf2f(hb);
    }
    acc.v[j] = (float)sum;
  }
  emu::wave_sync();
  return acc;
}
#define DLPD_MFMA_16x16x32_BF16(a, b, acc) dlpd_emu_mfma_16x16x32_bf16((a), (b), (acc))

