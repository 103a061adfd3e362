// Platform glue for the gfx950 build: HIP runtime + launch macro.
// (tests/emu/ carries a same-named header that runs the identical kernel sources as
//  cooperative fibers on the CPU -- a debugging/test harness, never part of the product.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DLPD_HD __host__ __device__ __forceinline__
#define DLPD_D __device__ __forceinline__
// kern must be parenthesised when it is a template-id: DLPD_LAUNCH((k<A,B>), grid, block, ...)
#define DLPD_LAUNCH_RAW(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
// dynamic LDS, 16-byte aligned base (guide G17)
#define DLPD_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[]; \
  type* name = reinterpret_cast<type*>(name##_raw_)

// LDS-DMA: each lane's 16 B at global address g go straight to LDS at (wave-uniform) l + lane*16
// Written as inline asm on purpose: hipcc tracks the builtin form as an LDS store of unknown
// extent and puts `s_waitcnt vmcnt(0)` in front of every later ds_read, which serialises the
// prefetch it is meant to overlap.  The issuer counts it by hand (DLPD_WAIT_VMEM).
__device__ __forceinline__ void dlpd_glds16(const void* g, void* l) {
  const unsigned la =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)l);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(la), "v"(g) : "memory", "m0");
}
#define DLPD_GLDS16(g, l) dlpd_glds16((const void*)(g), (void*)(l))
// the same from a wave-uniform base (scalar register pair) + a 32-bit byte offset per lane: the per-instruction address is
// scalar arithmetic, no 64-bit vector add
__device__ __forceinline__ void dlpd_glds16_so(const void* base, unsigned voff, void* l) {
  const unsigned la =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)l);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(la), "v"(voff), "s"(base) : "memory", "m0");
}
#define DLPD_GLDS16_SO(base, voff, l) dlpd_glds16_so((const void*)(base), (unsigned)(voff), (void*)(l))
// ... and the LDS destination as a 32-bit LDS address taken ONCE (the cast of a generic pointer costs a 64-bit add, two
// readfirstlanes, a null check and a select per use): dlpd_lds_t a = DLPD_LDS_ADDR(p); DLPD_GLDS16_SOA(base, voff, a + bytes)
typedef unsigned dlpd_lds_t;
__device__ __forceinline__ dlpd_lds_t dlpd_lds_addr(void* l) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)l);
}
__device__ __forceinline__ void dlpd_glds16_soa(const void* base, unsigned voff, dlpd_lds_t la) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(la), "v"(voff), "s"(base) : "memory", "m0");
}
#define DLPD_LDS_ADDR(p) dlpd_lds_addr((void*)(p))
#define DLPD_GLDS16_SOA(base, voff, la) dlpd_glds16_soa((const void*)(base), (unsigned)(voff), (la))
// barrier that orders LDS traffic only (leaves global loads / LDS-DMA in flight)
#define DLPD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DLPD_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DLPD_HAS_LDS_ADDRESS_SPACE 1                 // __attribute__((address_space(3))) pointers (not in the CPU emulator)
// wave priority for the instruction arbiter (0 lowest .. 3)
#define DLPD_SET_PRIO(n) __builtin_amdgcn_s_setprio(n)
// ordering point between lanes of ONE wave that exchange data through LDS: the hardware runs a
// wave's LDS instructions in order, so only the compiler must be kept from reordering them
#define DLPD_WAVE_SYNC() asm volatile("" ::: "memory")
#define DLPD_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// two consecutive floats from a 4-byte-aligned address with one global_load_dwordx2
struct __attribute__((packed, aligned(4))) dlpd_pair_t { float x, y; };
#define DLPD_PAIR dlpd_pair_t
__device__ __forceinline__ dlpd_pair_t dlpd_load_pair(const float* p) { return *reinterpret_cast<const dlpd_pair_t*>(p); }
// streaming (touch-once) global accesses: non-temporal, so they do not evict the reused lines
typedef float dlpd_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 dlpd_load_stream(const float4* p) {
  const dlpd_f4v v = __builtin_nontemporal_load(reinterpret_cast<const dlpd_f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void dlpd_store_stream(float4* p, float4 v) {
  const dlpd_f4v q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<dlpd_f4v*>(p));
}
__device__ __forceinline__ void dlpd_store_stream_c(float2* p, float2 v) {
  typedef float dlpd_f2s __attribute__((ext_vector_type(2)));
  const dlpd_f2s q = {v.x, v.y};
  __builtin_nontemporal_store(q, reinterpret_cast<dlpd_f2s*>(p));
}
__device__ __forceinline__ float2 dlpd_load_stream_c(const float2* p) {
  typedef float dlpd_f2s __attribute__((ext_vector_type(2)));
  const dlpd_f2s v = __builtin_nontemporal_load(reinterpret_cast<const dlpd_f2s*>(p));
  return make_float2(v.x, v.y);
}
#define DLPD_LOAD_STREAM(p) dlpd_load_stream(p)
#define DLPD_STORE_STREAM(p, v) dlpd_store_stream((p), (v))
#include <stdlib.h>
// clamp to [-c, c] in one v_med3_f32
#define DLPD_CLAMP(v, c) __builtin_amdgcn_fmed3f((v), -(c), (c))
// keeps the compiler from moving instructions across this point (software-pipelined loops)
#define DLPD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// value barriers for the optimiser: x leaves as "some vector / scalar register value" -- nothing computed from it is
// loop-invariant or shared with code before this point (no instruction is emitted)
#define DLPD_OPAQUE_V(x) asm volatile("" : "+v"(x))
// a wave-uniform integer as a SCALAR register value (addresses built from it use scalar arithmetic)
#define DLPD_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// uses of a scalar / vector register value that emit nothing (the compiler must have the value at this point)
#define DLPD_SINK_S(x) asm volatile("" ::"s"(x))
#define DLPD_SINK_V(x) asm volatile("" ::"v"(x))
#define DLPD_OPAQUE_S(x) asm volatile("" : "+s"(x))
// make a lane-dependent int opaque to the optimiser at this point: stops loop-invariant code motion
// from hoisting dozens of swizzled LDS offsets out of the pencil-set loops (they are cheap to
// recompute and expensive to keep in VGPRs)
#define DLPD_OPAQUE(x) asm volatile("" : "+v"(x))
// explicit 2 x f32 packed math (v_pk_fma_f32), independent of the SLP vectoriser
typedef float dlpd_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dlpd_f2v dlpd_f2_make(float a, float b) { dlpd_f2v r = {a, b}; return r; }
__device__ __forceinline__ dlpd_f2v dlpd_f2_splat(float a) { dlpd_f2v r = {a, a}; return r; }
__device__ __forceinline__ float dlpd_f2_get(dlpd_f2v v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ dlpd_f2v dlpd_pk_fma(dlpd_f2v a, dlpd_f2v b, dlpd_f2v c) { return __builtin_elementwise_fma(a, b, c); }

// table[bin] += 1 for every lane of the wave with `hit`, WITHOUT an LDS atomic: the lanes that name the same bin are found
// with ballots (scalar work only: `v_readlane` of the group's bin, one compare, one population count per DIFFERENT bin in
// the wave), the first lane of every group keeps the group's size, and then all of those lanes -- their bins are different --
// add their sizes with ONE plain read-modify-write.  All 64 lanes must call it together; `table` must be private to the
// wave.  (Why not ds_add_u32: EXPERIMENTS.md R5 / k_topk_hist.)
__device__ __forceinline__ void dlpd_lds_count(unsigned* table, unsigned bin, bool hit, int lane) {
  unsigned long long todo = __ballot(hit);
  unsigned mine = 0;
  while (todo) {
    const int first = __builtin_ctzll(todo);                                   // uniform: an SGPR
    const unsigned fb = (unsigned)__builtin_amdgcn_readlane((int)bin, first);
    const unsigned long long same = __ballot(hit && bin == fb);
    if (lane == first) mine = (unsigned)__popcll(same);
    todo &= ~same;
  }
  if (mine) table[bin] += mine;
}

// ------------------------------------------------------------------------------------------
// Packed complex arithmetic: one complex number = one aligned VGPR pair, one VOP3P instruction
// per complex add / rotate-add and two per complex multiply (FFT butterflies are VALU-issue
// bound: scalar f32 adds run 16 lanes per clock, v_pk_* twice that).  The +-i rotations and
// conjugations ride on the op_sel / neg modifiers -- hipcc does not fold those itself (it emits
// v_xor + v_mov per rotation), hence the inline asm.  Modifier semantics: op_sel[i] / op_sel_hi[i]
// = 1 makes the LOW / HIGH result lane read the HIGH half of source i; neg_lo / neg_hi negate
// source i for that lane.
// ------------------------------------------------------------------------------------------
#ifndef DLPD_PK
#define DLPD_PK 1
#endif
#if DLPD_PK
#define DLPD_PKV(a) dlpd_f2_make((a).x, (a).y)
__device__ __forceinline__ float2 dlpd_pkf(dlpd_f2v v) { return make_float2(v.x, v.y); }
__device__ __forceinline__ float2 dlpd_c_add(float2 a, float2 b) { return dlpd_pkf(DLPD_PKV(a) + DLPD_PKV(b)); }
__device__ __forceinline__ float2 dlpd_c_sub(float2 a, float2 b) { return dlpd_pkf(DLPD_PKV(a) - DLPD_PKV(b)); }
__device__ __forceinline__ float2 dlpd_c_scale(float2 a, float s) { return dlpd_pkf(DLPD_PKV(a) * s); }
// a + s*b (s real)
__device__ __forceinline__ float2 dlpd_c_axpy(float2 a, float s, float2 b) {
  return dlpd_pkf(__builtin_elementwise_fma(dlpd_f2_splat(s), DLPD_PKV(b), DLPD_PKV(a)));
}
// a - i*b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ float2 dlpd_c_add_mi(float2 a, float2 b) {
  dlpd_f2v d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(DLPD_PKV(a)), "v"(DLPD_PKV(b)));
  return dlpd_pkf(d);
}
// a + i*b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ float2 dlpd_c_add_pi(float2 a, float2 b) {
  dlpd_f2v d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(DLPD_PKV(a)), "v"(DLPD_PKV(b)));
  return dlpd_pkf(d);
}
// a * (c - i s) = c*a + s*(a.y, -a.x)   and   a * (c + i s) = c*a + s*(-a.y, a.x); c, s wave-uniform
__device__ __forceinline__ float2 dlpd_c_rotcs_m(float2 a, float c, float s) {
  const dlpd_f2v va = DLPD_PKV(a), t = va * c, vs = dlpd_f2_splat(s);
  dlpd_f2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(va), "s"(vs), "v"(t));
  return dlpd_pkf(d);
}
__device__ __forceinline__ float2 dlpd_c_rotcs_p(float2 a, float c, float s) {
  const dlpd_f2v va = DLPD_PKV(a), t = va * c, vs = dlpd_f2_splat(s);
  dlpd_f2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(va), "s"(vs), "v"(t));
  return dlpd_pkf(d);
}
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ float2 dlpd_c_mul(float2 a, float2 w) {
  dlpd_f2v t, d;
  const dlpd_f2v va = DLPD_PKV(a), vw = DLPD_PKV(w);
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(va), "v"(vw));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(va), "v"(vw), "v"(t));
  return dlpd_pkf(d);
}
// a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
__device__ __forceinline__ float2 dlpd_c_mulc(float2 a, float2 w) {
  dlpd_f2v t, d;
  const dlpd_f2v va = DLPD_PKV(a), vw = DLPD_PKV(w);
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(va), "v"(vw));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(va), "v"(vw), "v"(t));
  return dlpd_pkf(d);
}
#endif

// f32-input matrix core: D(16x16) += A(16x4) * B(4x16); lane l holds A[l&15][l>>4], B[l>>4][l&15] and
// D[4*(l>>4) + j][l&15] in element j of the accumulator
typedef float dlpd_acc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dlpd_acc4 dlpd_acc4_zero() { dlpd_acc4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float dlpd_acc4_get(dlpd_acc4 v, int j) { return v[j]; }
__device__ __forceinline__ dlpd_acc4 dlpd_acc4_make(float a, float b, float c, float d) { dlpd_acc4 z = {a, b, c, d}; return z; }
#define DLPD_MFMA_16x16x4(a, b, acc) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)
// bf16-input matrix core: D(16x16) += A(16x32) * B(32x16), f32 accumulate.  A fragment is 8 consecutive k values
// (16 bytes, passed as a float4): lane l holds A[l&15][8*(l>>4) + j] and B[8*(l>>4) + j][l&15], j = 0..7; D as 16x16x4.
typedef __bf16 dlpd_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ dlpd_acc4 dlpd_mfma_16x16x32_bf16(float4 a, float4 b, dlpd_acc4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dlpd_bf16x8, a), __builtin_bit_cast(dlpd_bf16x8, b), acc, 0, 0, 0);
}
#define DLPD_MFMA_16x16x32_BF16(a, b, acc) dlpd_mfma_16x16x32_bf16((a), (b), (acc))
// f32 -> bf16 bit pattern (round to nearest even: v_cvt_pk_bf16_f32) and back
__device__ __forceinline__ unsigned dlpd_f2bf(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float dlpd_bf2f(unsigned h) { return __uint_as_float(h << 16); }

