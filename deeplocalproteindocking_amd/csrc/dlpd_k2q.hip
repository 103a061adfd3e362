// K2 of the correlation pipeline for the N = 160 grid (box 80, the reference's real shapes): per (channel, kz) slab the
// zero-padded 2-D forward FFT of the rotated ligand's z-spectrum, the multiplication with the receptor spectrum
// (conjugate) and the 2-D inverse -- src/Models/DockingModels.py:70-71 (VolumeConvolution) between K1 and K3, see
// dlpd_corr.hip for the pipeline.
//
// The 160 x 160 complex slab (205 KB) does not fit the 160 KB LDS, so -- as in k_xy_corr_quad (dlpd_k2.hip), which
// this kernel replaces on the main path -- the slab is computed as FOUR dense 80 x 80 problems, decimation in
// frequency along x and y (w = exp(-2 pi i / N), H = N/2, a the L x L input, L = H):
//     F[2m+p][2n+q]         = FFT2_H( a[x][y] w^(p x + q y) )[m][n]
//     out[u + H s][v + H r] = sum_pq (-1)^(p s + q r) conj(w)^(p u + q v) IFFT2_H( rec[2m+p][2n+q] conj(F_pq) )[u][v]
// two sub-problems (q = 0, 1 of one p) side by side in LDS.  What is new (round 3; the quad kernel issued 4.5 x the
// vector instructions of the N = 128 kernel for 1.56 x the elements, half of them address arithmetic and idle lanes):
//   * 4-LANE PENCILS.  An 80-point transform runs on 4 threads as 20 x 4 (forward) / 4 x 20 (inverse): a radix-20
//     butterfly per thread (prime-factor 4 x 5, no internal twiddles) and five radix-4 ones -- every lane busy in every
//     pass, where 10 x 8 on 8 threads ran its radix-8 pass in two rounds with the second a quarter full.  A wave holds
//     16 pencils; the block has TEN waves and every phase (80 rows or 80 columns of two sub-slabs) exactly ten sets.
//   * AFFINE ADDRESSING.  Element (row, col) of sub-slab q lives at q*SUB + row*RS + col with RS = 84, SUB = 84 RS:
//     no XOR swizzle, so every LDS access of a pass is one lane-constant base plus an immediate.  Bank conflicts are
//     avoided by WHICH pencils share a wave instead: a row set is rows 8w..8w+7 of both sub-slabs (84 = 20 mod 32: the
//     eight rows of a 32-lane read group start on eight different multiples of 4 banks), a column set is two 4-column
//     blocks of both sub-slabs (SUB = 16 mod 32 puts the twin blocks 16 eight-byte columns apart); the first-pass
//     intermediates sit in blocks of 21 (forward) and in a rotated natural order (inverse: element r of butterfly j at
//     4 j + ((r + j) & 3)) so that the 16-lane store groups cover 16 distinct 8-byte columns as well.
//   * ROW OWNERSHIP.  Wave w owns rows 8w..8w+7 of both sub-slabs in every row phase: staging (with the pre-twiddles),
//     forward y, inverse y and the q-combination of those rows are wave-local, only the column phase sits between
//     block barriers -- two per sub-problem pair instead of five.
//   The last forward x pass (radix 4, outputs kx = j + 20 r) hands its registers to the first inverse x pass (radix 4
//   over exactly those inputs); H_0 waits in registers (40 VGPRs) for H_1 and the output slab is written once.
// Round 4, measured and not kept (git show f67fb5a): the radix-4 stage of every 80-point transform ACROSS the four lanes of
// its quad (two DPP exchange-and-add stages, lane 3 turning its difference by -+i in between; frequency k1 + 20 kb on the
// lane with kb = bit-reversed t, stored in blocks of 21), which makes a transform one read and one write of its pencil and
// the whole column phase -- butterfly, quad stage, receptor product, quad stage back, butterfly -- a register affair: LDS
// operations per sub-problem pair 390 -> 239, vector instructions 1,448 -> 2,001 (a DPP move per component and stage, the
// lane-3 turn as two selects, 299 hazard no-ops), inside the parity tolerance on emulator and GPU -- and K2
// 2.20 against 2.05-2.09 ms at the real shapes, 6.40 against 6.03 at 48 ch x 80^3.  The kernel's time follows the SUM of its
// vector and LDS work (about 62 % / 38 %); trading the cheaper for the dearer loses.
#include <dlpd_platform.h>
#include "dlpd_fft.h"
#include "dlpd_internal.h"

#ifndef DLPD_K2Q_RV_LATE
#define DLPD_K2Q_RV_LATE 1
#endif
#ifndef DLPD_K2Q_H0_REGS
#define DLPD_K2Q_H0_REGS 3                   // pairs of H_0 kept in registers (of 5); the rest waits in LDS
#endif
template <int N> DLPD_D void init_twiddles_k2q(cplx* tw, int tid, int nthreads) {
  for (int k = tid; k < N; k += nthreads) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    tw[k] = c_make((float)c, (float)s);
  }
}

#ifdef DLPD_STAMPS
__device__ unsigned long long dlpd_stamps_k2q[16];
extern "C" int dlpd_debug_read_stamps_k2q(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(dlpd_stamps_k2q), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k2q), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

// 80-point transforms of one pencil on 4 threads (thread t), element e of the pencil at S[base + e * ES].
// tw80: exp(-2 pi i k / 80).  Every function ends WITHOUT a trailing wave sync.
template <int ES> struct Q4 {
  // rotated natural order of the inverse plan's intermediate: element r of butterfly j at 4 j + ((r + j) & 3)
  struct Rot { int o[4]; };                            // o[k] = (t + k) & 3, times ES
  DLPD_D static Rot rot_of(int t) {
    Rot r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.o[k] = ((t + k) & 3) * ES;
    return r;
  }
  // forward pass A: radix 20 over inputs t + 4 r; outputs of butterfly t left at 21 t + r (blocks of 21)
  DLPD_D static void fwd_a(cplx* S, int base, int t) {
    cplx v[20];
    const cplx* p0 = S + base + t * ES;
#pragma unroll
    for (int r = 0; r < 20; r++) v[r] = lds_ld(p0 + 4 * r * ES);
    SmallDft<20, -1>::run(v);
    DLPD_WAVE_SYNC();
    cplx* p1 = S + base + 21 * t * ES;
#pragma unroll
    for (int r = 0; r < 20; r++) lds_st(p1 + r * ES, v[r]);
  }
  // the same for a zero-padded pencil: only inputs 0 .. 39 are non-zero (r < 10), the others are not even read
  DLPD_D static void fwd_a_pruned(cplx* S, int base, int t) {
    cplx v[20];
    const cplx* p0 = S + base + t * ES;
#pragma unroll
    for (int r = 0; r < 10; r++) v[r] = lds_ld(p0 + 4 * r * ES);
#pragma unroll
    for (int r = 10; r < 20; r++) v[r] = c_make(0.f, 0.f);
    SmallDft<20, -1>::run(v);
    DLPD_WAVE_SYNC();
    cplx* p1 = S + base + 21 * t * ES;
#pragma unroll
    for (int r = 0; r < 20; r++) lds_st(p1 + r * ES, v[r]);
  }
  // forward pass B: five radix-4 butterflies j = t + 4 i over the blocks' offset j, twiddle tw80[j r]; results
  // u[i][r] = X[j + 20 r] stay in registers
  DLPD_D static void fwd_b_load(const cplx* S, int base, int t, const cplx* tw80, cplx (&u)[5][4]) {
    const cplx* p0 = S + base + t * ES;
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) u[i][r] = lds_ld(p0 + (21 * r + 4 * i) * ES);
#pragma unroll
      for (int r = 1; r < 4; r++) u[i][r] = c_mul(u[i][r], tw80[t * r + 4 * i * r]);
      dft4<-1>(u[i][0], u[i][1], u[i][2], u[i][3]);
      DLPD_SCHED_FENCE();
    }
  }
  DLPD_D static void store_nat(cplx* S, int base, int t, const cplx (&u)[5][4]) {       // X[j + 20 r], j = t + 4 i
    cplx* p0 = S + base + t * ES;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) lds_st(p0 + (4 * i + 20 * r) * ES, u[i][r]);
  }
  DLPD_D static void load_nat(const cplx* S, int base, int t, cplx (&u)[5][4]) {
    const cplx* p0 = S + base + t * ES;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) u[i][r] = lds_ld(p0 + (4 * i + 20 * r) * ES);
  }
  // inverse pass A on registers: radix 4 (no twiddles) over u[i][0..3] = inputs j + 20 r of butterfly j = t + 4 i;
  // output r of butterfly j stored at 4 j + ((r + j) & 3)
  DLPD_D static void inv_a_store(cplx* S, int base, int t, const Rot& rt, cplx (&u)[5][4]) {
#pragma unroll
    for (int i = 0; i < 5; i++) dft4<+1>(u[i][0], u[i][1], u[i][2], u[i][3]);
    DLPD_WAVE_SYNC();
    cplx* p4 = S + base + 4 * t * ES;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) lds_st(p4 + rt.o[r] + 16 * i * ES, u[i][r]);
  }
  // inverse pass B: radix 20 of butterfly t over inputs (block r, element t) with twiddle conj(tw80[t r]); v[r] = x[t + 4 r]
  DLPD_D static void inv_b_load(const cplx* S, int base, int t, const Rot& rt, const cplx* tw80, cplx (&v)[20]) {
    const cplx* pb = S + base;
    // in chunks of five: with all 20 values and 19 twiddles requested at once the pass needs 80 registers before the
    // first multiplication
#pragma unroll
    for (int r0 = 0; r0 < 20; r0 += 5) {
#pragma unroll
      for (int r = r0; r < r0 + 5; r++) v[r] = lds_ld(pb + rt.o[r & 3] + 4 * r * ES);
#pragma unroll
      for (int r = r0; r < r0 + 5; r++)
        if (r) v[r] = c_mulc(v[r], tw80[t * r]);
      DLPD_SCHED_FENCE();
    }
    SmallDft<20, +1>::run(v);
  }
  DLPD_D static void store_a(cplx* S, int base, int t, const cplx (&v)[20]) {          // x[t + 4 r]
    cplx* p0 = S + base + t * ES;
#pragma unroll
    for (int r = 0; r < 20; r++) lds_st(p0 + 4 * r * ES, v[r]);
  }
  // whole transforms, in place
  DLPD_D static void forward(cplx* S, int base, int t, const cplx* tw80) {
    fwd_a(S, base, t);
    DLPD_WAVE_SYNC();
    cplx u[5][4];
    fwd_b_load(S, base, t, tw80, u);
    DLPD_WAVE_SYNC();
    store_nat(S, base, t, u);
  }
  DLPD_D static void inverse(cplx* S, int base, int t, const Rot& rt, const cplx* tw80) {
    {
      cplx u[5][4];
      load_nat(S, base, t, u);
      inv_a_store(S, base, t, rt, u);
    }
    DLPD_WAVE_SYNC();
    cplx v[20];
    inv_b_load(S, base, t, rt, tw80, v);
    DLPD_WAVE_SYNC();
    store_a(S, base, t, v);
  }
};

//   A    (nb, CT, NZ, L, L) complex [kz][x][y]   (K1's output)
//   rec  (CT, NZ, N, N) complex [kz][kx][ky], times 1/N^3; rec_bstride: element stride between batch entries (0: shared)
//   out  (nb, CT, NZ, N, N) complex [kz][x'][y'] = IFFT2( rec * conj(FFT2(pad(A))) )   (unnormalised inverse)
//   grid NZ*CT*nsplit (XCD-aware decode as in k_xy_corr), block 640 threads, persistent over its part of the batch
//   PACKED: rec is the receptor spectrum re-ordered by k_pack_rec_q4 (below) -- every wave's 20 values per lane as twenty
//   contiguous half-kilobytes per (slab, p) -- instead of the natural [kz][kx][ky]
template <int N, bool PACKED> __global__ void __launch_bounds__(640)
k_xy_corr_q4(const cplx* __restrict__ A, const cplx* __restrict__ rec, cplx* __restrict__ out,
             int CT, int nb, int nsplit, long long rec_bstride, const unsigned char* __restrict__ pmap, int nmasked) {
  constexpr int L = N / 2, H = N / 2, NZ = N / 2 + 1;
  constexpr int RS = H + 4, HR = H + 4, SUB = HR * RS;   // 84 x 84: rows / columns 80..82 take the blocks-of-21 intermediates
  constexpr int NT = 640, W = NT / 64;
  static_assert(H == 80 && 8 * W == H, "ten waves: eight rows and two 4-column blocks of both sub-slabs each");
  static_assert(RS % 32 == 20 && SUB % 32 == 16 && RS % 2 == 0, "bank spreading of the row / column sets");
  constexpr int NF4 = 8 * L / 2;                       // float4 (two complex) in a wave's 8 rows of A
  constexpr int NP = NF4 / 64;                         // per lane
  static_assert(NF4 % 64 == 0, "whole waves per row set");
  DLPD_DYN_SHARED(cplx, S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int part = (bid >> 3) % nsplit;
  const int slab = (bid / (8 * nsplit)) * 8 + (bid & 7);
  if (slab >= NZ * CT) return;
  const int kz = slab % NZ, c = slab / NZ;
  const int b_beg = (int)(((long long)nb * part) / nsplit), b_end = (int)(((long long)nb * (part + 1)) / nsplit);
  if (b_beg >= b_end) return;
  cplx* tw = S + 2 * SUB;                              // exp(-2 pi i k / N)
  cplx* twh = tw + N;                                  // exp(-2 pi i k / H)
  init_twiddles_k2q<N>(tw, tid, NT);
  init_twiddles_k2q<H>(twh, tid, NT);

  const int t = lane & 3, pidx = lane >> 2;            // thread of the pencil, pencil of the set
  // row set: rows 8w..8w+7 of sub-slab 0 (lanes 0-31) and of sub-slab 1 (lanes 32-63)
  const int rbase = (pidx >> 3) * SUB + (8 * wave + (pidx & 7)) * RS;
  // column set: 4-column blocks 2w (lanes 0-31) and 2w+1 (lanes 32-63), each of sub-slab 0 (first 16 lanes) and 1
  const int cq = (pidx >> 2) & 1, ccol = 4 * (2 * wave + (pidx >> 3)) + (pidx & 3);
  const int cbase = cq * SUB + ccol;
  const typename Q4<1>::Rot rot_r = Q4<1>::rot_of(t);
  const typename Q4<RS>::Rot rot_c = Q4<RS>::rot_of(t);

  // this wave's 8 rows of a rotation's A slab (8 L contiguous complex): lane = 8 * row + piece reads float4 number
  // piece + 8 k of its row -- eight 128-byte lines per instruction and no division anywhere (staging and combination
  // use the same dealing: their (row, column) are a shift, a mask and an immediate)
  static_assert(L / 2 == 8 * NP, "row = 8 lanes x NP float4");
  float4 apref[NP];
  // PENCIL MAP (round 6; pmap != null, channels c < nmasked): bit (y cell) of the 32-bit word pmap[b][x cell] = 0 says that pencil (x, y) of rotation
  // b's rotated ligand is all zero (dlpd_rotated_occupancy) -- K1 did not write it (dlpd_zfft_channels_last_occ, skip_empty) and
  // it is NOT read here: the pair is the zero it stands for.  A float4 is two neighbouring y of one row: one cell.
  constexpr int NC = (L + 3) / 4;
  const bool by_map = pmap != nullptr && c < nmasked;          // (block-uniform)
  auto fetch_A = [&](int b) {
    const int srow = 8 * wave + (lane >> 3);
    const float4* a4 = reinterpret_cast<const float4*>(A + (((size_t)b * CT + c) * NZ + kz) * L * L + (size_t)srow * L) + (lane & 7);
    if (by_map) {
      // one 32-bit word per (rotation, x cell): bit = y cell.  The wave's 8 rows are x cells 2w (lanes 0-31) and 2w + 1: two
      // words at a wave-uniform address (scalar loads: no vector load waits for a vector load)
      const unsigned* pw = reinterpret_cast<const unsigned*>(pmap) + (size_t)b * NC + 2 * DLPD_UNIFORM(wave);
      const unsigned m0 = pw[0], m1 = pw[1];
      const unsigned m = ((lane & 32) ? m1 : m0) >> ((lane & 7) >> 1);                               // y cell = ((lane & 7) + 8 k) / 2
#pragma unroll
      for (int k = 0; k < NP; k++) apref[k] = ((m >> (4 * k)) & 1u) ? DLPD_LOAD_STREAM(a4 + 8 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int k = 0; k < NP; k++) apref[k] = DLPD_LOAD_STREAM(a4 + 8 * k);
    }
  };
  // H_0[u][v..v+1], H_0[u][v+H..] wait for H_1: NPR of the NP pairs in registers, the others parked in the LDS the
  // sub-slabs leave free (with all 40 registers held the column phase of p = 1 spills)
  constexpr int NPR = DLPD_K2Q_H0_REGS;
  float4 h0[NPR > 0 ? NPR : 1][2];
  float4* park = reinterpret_cast<float4*>(twh + H) + tid;     // [(k - NPR) * 2 + j][NT]
  fetch_A(b_beg);
  DLPD_LDS_BARRIER();                                  // twiddle tables visible
  DLPD_STAMP_DECL;
  for (int b = b_beg; b < b_end; b++) {
    float4* o = reinterpret_cast<float4*>(out + (((size_t)b * CT + c) * NZ + kz) * N * N);
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
      // ---- own rows of the A slab (registers) times w^(p x) -> sub-slab q = 0, times w^y -> sub-slab q = 1
      {
        int lq = lane;
        DLPD_OPAQUE(lq);                               // (row and column are recomputed, not kept in VGPRs)
        const int x = 8 * wave + (lq >> 3), scol = 2 * (lq & 7);
        cplx wx = c_make(1.f, 0.f);
        if (p) wx = tw[x];
#pragma unroll
        for (int k = 0; k < NP; k++) {
          const int y = scol + 16 * k;
          cplx u = c_make(apref[k].x, apref[k].y), v = c_make(apref[k].z, apref[k].w);
          if (p) { u = c_mul(u, wx); v = c_mul(v, wx); }
          cplx* d = S + x * RS + y;
          const float4 wy = *reinterpret_cast<const float4*>(tw + y);
          const cplx u1 = c_mul(u, c_make(wy.x, wy.y)), v1 = c_mul(v, c_make(wy.z, wy.w));
          *reinterpret_cast<float4*>(d) = make_float4(u.x, u.y, v.x, v.y);
          *reinterpret_cast<float4*>(d + SUB) = make_float4(u1.x, u1.y, v1.x, v1.y);
        }
      }
      DLPD_WAVE_SYNC();
      DLPD_STAMP(0);
      // ---- forward along y: the wave's own rows of both sub-slabs
      Q4<1>::forward(S, rbase, t, twh);
      DLPD_STAMP(1);
      // receptor values of this wave's column set: rec[2 m + p][2 n + q] for m = t + 4 i + 20 r, in flight over the barrier
      // (DLPD_K2Q_RV_LATE: requested behind the first column pass instead -- 40 registers less under the radix-20 butterfly)
      cplx rv[5][4];
      auto fetch_rec = [&]() {
        if constexpr (PACKED) {
          // natural layout: lane (t, column 2 ccol + cq) reads rows 2 (t + 4 i + 20 r) + p -- eight lanes share a 64-byte
          // run, every instruction touches eight half lines (measured: the loads cost 0.45 of K2's 2.40 ms at the real
          // shapes); packed: 512 contiguous bytes per instruction
          const cplx* rp = rec + ((((size_t)c * NZ + kz) * 2 + p) * W + DLPD_UNIFORM(wave)) * 1280;
#pragma unroll
          for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) rv[i][r] = rp[(unsigned)((4 * i + r) * 64 + lane)];
        } else {
          const cplx* rb = rec + (size_t)b * rec_bstride + (((size_t)c * NZ + kz) * N + p) * N + (2 * ccol + cq) +
                           (size_t)(2 * N) * t;
#pragma unroll
          for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) rv[i][r] = rb[(unsigned)(2 * N * (4 * i + 20 * r))];
        }
      };
      if (!DLPD_K2Q_RV_LATE) fetch_rec();
      DLPD_LDS_BARRIER();                              // all rows y-transformed
      DLPD_STAMP(2);
      // ---- columns: forward x, receptor multiply, inverse x
      {
        Q4<RS>::fwd_a(S, cbase, t);
        if (DLPD_K2Q_RV_LATE) fetch_rec();
        DLPD_WAVE_SYNC();
        // forward pass B, the receptor multiplication and the inverse pass A butterfly by butterfly (the radix-4
        // butterfly j = t + 4 i of both passes works on the same four elements kx = j + 20 r), so that the receptor
        // values are released as the results pile up
        cplx u[5][4];
        {
          const cplx* p0 = S + cbase + t * RS;
#pragma unroll
          for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) u[i][r] = lds_ld(p0 + (21 * r + 4 * i) * RS);
#pragma unroll
            for (int r = 1; r < 4; r++) u[i][r] = c_mul(u[i][r], twh[t * r + 4 * i * r]);
            dft4<-1>(u[i][0], u[i][1], u[i][2], u[i][3]);
#pragma unroll
            for (int r = 0; r < 4; r++) u[i][r] = c_mulc(rv[i][r], u[i][r]);
            dft4<+1>(u[i][0], u[i][1], u[i][2], u[i][3]);
            DLPD_SCHED_FENCE();
          }
        }
        DLPD_WAVE_SYNC();
        {
          cplx* p4 = S + cbase + 4 * t * RS;
#pragma unroll
          for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) lds_st(p4 + rot_c.o[r] + 16 * i * RS, u[i][r]);
        }
      }
      DLPD_WAVE_SYNC();
      {
        cplx v[20];
        Q4<RS>::inv_b_load(S, cbase, t, rot_c, twh, v);
        DLPD_WAVE_SYNC();
        Q4<RS>::store_a(S, cbase, t, v);
      }
      DLPD_STAMP(3);
      DLPD_LDS_BARRIER();                              // all columns done
      DLPD_STAMP(4);
      if (p == 1 && b + 1 < b_end) fetch_A(b + 1);     // next rotation's rows, in flight over the rest of this slab
      // ---- inverse along y: own rows -> G_p0, G_p1
      Q4<1>::inverse(S, rbase, t, rot_r, twh);
      DLPD_WAVE_SYNC();
      DLPD_STAMP(5);
      // ---- H_p[u][v + H r] = G_p0[u][v] + (-1)^r conj(w^v) G_p1[u][v];  out[u + H s][y'] = H_0 + (-1)^s conj(w^u) H_1
      {
        int lq = lane;
        DLPD_OPAQUE(lq);
        const int u = 8 * wave + (lq >> 3), scol = 2 * (lq & 7);
        cplx wu = c_make(1.f, 0.f);
        if (p) wu = tw[u];
#pragma unroll
        for (int k = 0; k < NP; k++) {
          const int v = scol + 16 * k;
          const cplx* g = S + u * RS + v;
          const float4 ga = *reinterpret_cast<const float4*>(g), gb = *reinterpret_cast<const float4*>(g + SUB);
          const float4 wv = *reinterpret_cast<const float4*>(tw + v);
          const cplx a0 = c_make(ga.x, ga.y), a1 = c_make(ga.z, ga.w);
          const cplx b0 = c_mulc(c_make(gb.x, gb.y), c_make(wv.x, wv.y)), b1 = c_mulc(c_make(gb.z, gb.w), c_make(wv.z, wv.w));
          const cplx lo0 = c_add(a0, b0), lo1 = c_add(a1, b1), hi0 = c_sub(a0, b0), hi1 = c_sub(a1, b1);
          if (p == 0) {
            const float4 lo = make_float4(lo0.x, lo0.y, lo1.x, lo1.y), hi = make_float4(hi0.x, hi0.y, hi1.x, hi1.y);
            if (k < NPR) {
              h0[k][0] = lo;
              h0[k][1] = hi;
            } else {
              park[((k - NPR) * 2) * NT] = lo;
              park[((k - NPR) * 2 + 1) * NT] = hi;
            }
          } else {
            const cplx l0 = c_mulc(lo0, wu), l1 = c_mulc(lo1, wu), k0 = c_mulc(hi0, wu), k1 = c_mulc(hi1, wu);
            const float4 fa = k < NPR ? h0[k < NPR ? k : 0][0] : park[((k - NPR) * 2) * NT];
            const float4 fb = k < NPR ? h0[k < NPR ? k : 0][1] : park[((k - NPR) * 2 + 1) * NT];
            const cplx fa0 = c_make(fa.x, fa.y), fa1 = c_make(fa.z, fa.w), fb0 = c_make(fb.x, fb.y), fb1 = c_make(fb.z, fb.w);
            const cplx o00 = c_add(fa0, l0), o01 = c_add(fa1, l1), o10 = c_add(fb0, k0), o11 = c_add(fb1, k1);
            const cplx o20 = c_sub(fa0, l0), o21 = c_sub(fa1, l1), o30 = c_sub(fb0, k0), o31 = c_sub(fb1, k1);
            DLPD_STORE_STREAM(o + (u * N + v) / 2, make_float4(o00.x, o00.y, o01.x, o01.y));
            DLPD_STORE_STREAM(o + (u * N + v + H) / 2, make_float4(o10.x, o10.y, o11.x, o11.y));
            DLPD_STORE_STREAM(o + ((u + H) * N + v) / 2, make_float4(o20.x, o20.y, o21.x, o21.y));
            DLPD_STORE_STREAM(o + ((u + H) * N + v + H) / 2, make_float4(o30.x, o30.y, o31.x, o31.y));
          }
        }
      }
      DLPD_WAVE_SYNC();                                // own rows fully read before they are refilled
      DLPD_STAMP(6);
    }
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k2q, DLPD_STAMPS);
}

// ------------------------------------------------------------------------------------------------------------------
// N = 80 (box 40: the coarse grid of the reference's real shapes): the 80 x 80 slab FITS (56 KB: two blocks per CU), so
// no decimation -- the zero-padded forward transforms (40 non-zero rows / columns: pruned first passes), the receptor
// product and the inverse -- on the same 4-lane pencils and affine addressing.  Five waves: wave w stages and
// y-transforms input rows 8w..8w+7 (its upper 32 lanes run the same code on rows 40 + 8w.., which nobody reads before the
// column passes overwrite them), owns 16 columns of the column phase, and rows 16w..16w+15 of the inverse y transform,
// which it also copies out; two block barriers per slab.
// ------------------------------------------------------------------------------------------------------------------
// column of lane group `pidx` of wave `wave` in the column phase of k_xy_corr_s4 (shared with the receptor packing)
DLPD_HD int k2s4_column(int wave, int pidx) {
  // pairs of 4-column blocks 16 eight-byte columns apart (mod 32) share a 32-lane read group; the last wave's
  // blocks 16..19 have no such partner (a 2-way conflict on an eighth of its loads)
  const int pr = 2 * wave + (pidx >> 3);
  const int blk = (pr < 8) ? ((pr & 3) + 8 * (pr >> 2) + 4 * ((pidx >> 2) & 1)) : (16 + 2 * (pr - 8) + ((pidx >> 2) & 1));
  return 4 * blk + (pidx & 3);
}
template <int N, bool PACKED> __global__ void __launch_bounds__(320)
k_xy_corr_s4(const cplx* __restrict__ A, const cplx* __restrict__ rec, cplx* __restrict__ out,
             int CT, int nb, int nsplit, long long rec_bstride, const unsigned char* __restrict__ pmap, int nmasked) {
  constexpr int L = N / 2, NZ = N / 2 + 1, RS = N + 4, HR = N + 4;
  constexpr int NT = 320, W = NT / 64;
  static_assert(N == 80 && 16 * W == N && 8 * W == L, "five waves: 8 input rows, 16 columns and 16 output rows each");
  static_assert(RS % 32 == 20 && RS % 2 == 0, "bank spreading of the row / column sets");
  constexpr int NIN = 8 * L / 2;                       // float4 in a wave's 8 input rows (160)
  constexpr int NPI = (NIN + 63) / 64;
  constexpr int NOUT = 16 * N / 2 / 64;                // float4 per lane of a wave's 16 output rows (10)
  DLPD_DYN_SHARED(cplx, S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int part = (bid >> 3) % nsplit;
  const int slab = (bid / (8 * nsplit)) * 8 + (bid & 7);
  if (slab >= NZ * CT) return;
  const int kz = slab % NZ, c = slab / NZ;
  const int b_beg = (int)(((long long)nb * part) / nsplit), b_end = (int)(((long long)nb * (part + 1)) / nsplit);
  if (b_beg >= b_end) return;
  cplx* twh = S + HR * RS;                             // exp(-2 pi i k / N)
  init_twiddles_k2q<N>(twh, tid, NT);
  const int t = lane & 3, pidx = lane >> 2;
  // forward rows: lanes 0-31 the wave's 8 input rows, lanes 32-63 dummy rows 40 + 8w..
  const int fbase = ((pidx >> 3) * L + 8 * wave + (pidx & 7)) * RS;
  // inverse rows: 16w .. 16w + 15
  const int ibase = (16 * wave + pidx) * RS;
  const int ccol = k2s4_column(wave, pidx);
  const int cbase = ccol;
  const typename Q4<1>::Rot rot_r = Q4<1>::rot_of(t);
  const typename Q4<RS>::Rot rot_c = Q4<RS>::rot_of(t);

  float4 apref[NPI];
  constexpr int NC = (L + 3) / 4;                       // pencil map (k_xy_corr_q4): pmap[b][x cell][y cell]
  const bool by_map = pmap != nullptr && c < nmasked;
  auto fetch_A = [&](int b) {
    const float4* a4 = reinterpret_cast<const float4*>(A + (((size_t)b * CT + c) * NZ + kz) * L * L + (size_t)wave * 8 * L);
    unsigned m0 = ~0u, m1 = ~0u;                         // bit (y cell) of word pmap[b][x cell]; the wave's rows are x cells 2w, 2w + 1
    if (by_map) {
      const unsigned* pw = reinterpret_cast<const unsigned*>(pmap) + (size_t)b * NC + 2 * DLPD_UNIFORM(wave);
      m0 = pw[0];
      m1 = pw[1];
    }
#pragma unroll
    for (int k = 0; k < NPI; k++)
      if (lane + 64 * k < NIN) {
        const int f = lane + 64 * k, r = f / (L / 2), yc = (f % (L / 2)) >> 1;
        const bool live = (((r & 4) ? m1 : m0) >> yc) & 1u;
        apref[k] = live ? DLPD_LOAD_STREAM(a4 + f) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  };
  fetch_A(b_beg);
  DLPD_LDS_BARRIER();                                  // twiddle table visible
  for (int b = b_beg; b < b_end; b++) {
    // ---- own input rows -> slab rows 8w..8w+7, columns 0..L-1
#pragma unroll
    for (int k = 0; k < NPI; k++) {
      const int f = lane + 64 * k, r = f / (L / 2), y = 2 * (f % (L / 2));
      if (f < NIN) *reinterpret_cast<float4*>(S + (8 * wave + r) * RS + y) = apref[k];
    }
    DLPD_WAVE_SYNC();
    // ---- forward along y (zero-padded: pruned first pass)
    Q4<1>::fwd_a_pruned(S, fbase, t);
    DLPD_WAVE_SYNC();
    {
      cplx u[5][4];
      Q4<1>::fwd_b_load(S, fbase, t, twh, u);
      DLPD_WAVE_SYNC();
      Q4<1>::store_nat(S, fbase, t, u);
    }
    cplx rv[5][4];
    if constexpr (PACKED) {
      // (natural layout: four lanes share a 32-byte run)
      const cplx* rp = rec + (((size_t)c * NZ + kz) * W + DLPD_UNIFORM(wave)) * 1280;
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) rv[i][r] = rp[(unsigned)((4 * i + r) * 64 + lane)];
    } else {
      const cplx* rb = rec + (size_t)b * rec_bstride + ((size_t)c * NZ + kz) * N * N + ccol + (size_t)N * t;
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) rv[i][r] = rb[(unsigned)(N * (4 * i + 20 * r))];
    }
    if (b + 1 < b_end) fetch_A(b + 1);
    DLPD_LDS_BARRIER();                                // all rows y-transformed
    // ---- columns: forward x (pruned), receptor product, inverse x
    {
      Q4<RS>::fwd_a_pruned(S, cbase, t);
      DLPD_WAVE_SYNC();
      cplx u[5][4];
      {
        const cplx* p0 = S + cbase + t * RS;
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
          for (int r = 0; r < 4; r++) u[i][r] = lds_ld(p0 + (21 * r + 4 * i) * RS);
#pragma unroll
          for (int r = 1; r < 4; r++) u[i][r] = c_mul(u[i][r], twh[t * r + 4 * i * r]);
          dft4<-1>(u[i][0], u[i][1], u[i][2], u[i][3]);
#pragma unroll
          for (int r = 0; r < 4; r++) u[i][r] = c_mulc(rv[i][r], u[i][r]);
          dft4<+1>(u[i][0], u[i][1], u[i][2], u[i][3]);
          DLPD_SCHED_FENCE();
        }
      }
      DLPD_WAVE_SYNC();
      cplx* p4 = S + cbase + 4 * t * RS;
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) lds_st(p4 + rot_c.o[r] + 16 * i * RS, u[i][r]);
    }
    DLPD_WAVE_SYNC();
    {
      cplx v[20];
      Q4<RS>::inv_b_load(S, cbase, t, rot_c, twh, v);
      DLPD_WAVE_SYNC();
      Q4<RS>::store_a(S, cbase, t, v);
    }
    DLPD_LDS_BARRIER();                                // all columns done
    // ---- inverse along y of the wave's 16 rows, then those rows to global memory (contiguous 10 KB)
    Q4<1>::inverse(S, ibase, t, rot_r, twh);
    DLPD_WAVE_SYNC();
    {
      float4* o = reinterpret_cast<float4*>(out + (((size_t)b * CT + c) * NZ + kz) * N * N + (size_t)wave * 16 * N);
#pragma unroll
      for (int k = 0; k < NOUT; k++) {
        const int f = lane + 64 * k, r = f / (N / 2), y = 2 * (f % (N / 2));
        DLPD_STORE_STREAM(o + f, *reinterpret_cast<const float4*>(S + (16 * wave + r) * RS + y));
      }
    }
    DLPD_WAVE_SYNC();                                  // own rows read before the next rotation's staging refills rows 8w..
    // (rows 8w..8w+7 and 40+8w.. of the next staging / dummy transform belong to OTHER waves' output rows when
    //  w > 0: the block barrier below keeps them from being refilled while still being copied out)
    DLPD_LDS_BARRIER();
  }
}

int dlpd_k2q_supported(int L) { return (L == 80 || L == 40) ? 1 : 0; }

// The receptor spectrum in the order the column phases read it: complex number
//   N = 160:  (((slab * 2 + p) * 10 + wave) * 20 + k) * 64 + lane  =  rec[slab][2 m + p][2 ccol + cq],
//   N =  80:  ((slab * 5 + wave) * 20 + k) * 64 + lane             =  rec[slab][m][ccol]
// with k = 4 i + r, m = t + 4 i + 20 r and (t, ccol, cq) the lane's pencil thread and column as in the kernels above: a
// wave's load number k is 512 contiguous bytes.  Same size as the natural layout; written once per receptor.
template <int N> __global__ void __launch_bounds__(256) k_pack_rec_k2q(const cplx* __restrict__ rec, cplx* __restrict__ packed,
                                                                      long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx % 64), k = (int)((idx / 64) % 20);
  const int t = lane & 3, pidx = lane >> 2;
  const int m = t + 4 * (k / 4) + 20 * (k % 4);
  if constexpr (N == 160) {
    const int wave = (int)((idx / 1280) % 10), p = (int)((idx / 12800) % 2);
    const long long slab = idx / 25600;
    const int cq = (pidx >> 2) & 1, ccol = 4 * (2 * wave + (pidx >> 3)) + (pidx & 3);
    packed[idx] = rec[(slab * N + (2 * m + p)) * N + 2 * ccol + cq];
  } else {
    const int wave = (int)((idx / 1280) % 5);
    const long long slab = idx / 6400;
    packed[idx] = rec[(slab * N + m) * N + k2s4_column(wave, pidx)];
  }
}

int dlpd_k2q_pack_receptor(const cplx* rec, void* packed, int CT, int L, hipStream_t st) {
  if (L != 80 && L != 40) return DLPD_ERR_UNSUPPORTED;
  const int N = 2 * L, NZ = N / 2 + 1;
  const long long total = (long long)CT * NZ * N * N;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (L == 80) DLPD_LAUNCH((k_pack_rec_k2q<160>), grid, dim3(256), 0, st, rec, (cplx*)packed, total);
  else DLPD_LAUNCH((k_pack_rec_k2q<80>), grid, dim3(256), 0, st, rec, (cplx*)packed, total);
  return dlpd_check_launch();
}

template <bool PACKED> static int k2q_launch(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, int L, long long rbs,
                                             int nsplit, hipStream_t st, const unsigned char* pmap = nullptr, int nmasked = 0) {
  if (L == 40) {
    constexpr int N = 80, NZ = N / 2 + 1, RS = N + 4;
    const size_t shmem = (size_t)(RS * RS + N) * sizeof(cplx);
    int rc = dlpd_set_max_dyn_shared((const void*)k_xy_corr_s4<N, PACKED>, shmem);
    if (rc) return rc;
    const int slabs8 = ((NZ * CT + 7) / 8) * 8;
    DLPD_LAUNCH((k_xy_corr_s4<N, PACKED>), dim3(slabs8 * nsplit), dim3(320), shmem, st, A, rec, out, CT, nb, nsplit, rbs, pmap, nmasked);
    return dlpd_check_launch();
  }
  if (L != 80) return DLPD_ERR_UNSUPPORTED;
  constexpr int N = 160, NZ = N / 2 + 1, H = N / 2, RS = H + 4;
  const size_t shmem = (size_t)(2 * RS * RS + N + H) * sizeof(cplx) + (size_t)(5 - DLPD_K2Q_H0_REGS) * 2 * 640 * sizeof(float4);
  int rc = dlpd_set_max_dyn_shared((const void*)k_xy_corr_q4<N, PACKED>, shmem);
  if (rc) return rc;
  const int slabs8 = ((NZ * CT + 7) / 8) * 8;
  DLPD_LAUNCH((k_xy_corr_q4<N, PACKED>), dim3(slabs8 * nsplit), dim3(640), shmem, st, A, rec, out, CT, nb, nsplit, rbs, pmap, nmasked);
  return dlpd_check_launch();
}

// packed: rec is dlpd_k2q_pack_receptor's output (one receptor shared by the batch)
int dlpd_k2q_correlate(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, int L, long long rbs, int nsplit_override,
                       hipStream_t st, int packed, const unsigned char* pmap, int nmasked) {
  int nsplit = nb >= 8 ? 2 : 1;
  if (nsplit_override > 0) nsplit = nsplit_override;
  if (pmap && !packed) return DLPD_ERR_UNSUPPORTED;
  if (packed) return rbs ? DLPD_ERR_ARG : k2q_launch<true>(A, rec, out, CT, nb, L, 0, nsplit, st, pmap, nmasked);
  return k2q_launch<false>(A, rec, out, CT, nb, L, rbs, nsplit, st);
}
