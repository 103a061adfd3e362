// K2 of the correlation pipeline (see dlpd_corr.hip for the overview): the per-slab 2-D FFT /
// receptor multiply / 2-D inverse.  Separate translation unit because this kernel runs ~13 %
// faster built with -fno-slp-vectorize (at 248 VGPRs the SLP vectoriser's register pairing costs
// more moves than its packed adds save), while K3's MLP wants the vectoriser's v_pk_fma_f32.
#include <dlpd_platform.h>
#include <type_traits>
#include "dlpd_fft.h"
#include "dlpd_internal.h"

template <int N> DLPD_D void init_twiddles(cplx* tw, int tid, int nthreads) {
  for (int k = tid; k < N; k += nthreads) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    tw[k] = c_make((float)c, (float)s);
  }
}

// ------------------------------------------------------------------------------------------
// K2: one block per (c, kz), looping over the nb rotations of the batch (persistent over b).
//   1-D grid NZ*CT*nsplit, block 4N threads (W = N/16 waves), dynamic LDS N*(N+8)*8 B (one swizzled N x N slab).
//   MODE 0: forward only -> out[(b*CT+c)][kz][kx][ky] = scale * FFT2(pad(A))     (receptor prep)
//   MODE 1: correlate    -> out = IFFT2( rec * conj(FFT2(pad(A))) )  (unnormalised inverse;
//                           the 1/N^3 lives in rec)
//   rec_bstride: element stride between batch entries of rec (0: shared receptor)
// Per slab: y-forward on the L non-zero rows, x-forward on all columns (pruned first passes), the
// receptor multiply in registers, and -- because the Stockham output of the last forward x pass
// leaves thread t with exactly the elements {t + 8m} that the first inverse x pass needs -- the
// inverse x transform starts from those registers without a trip through LDS; then y-inverse on all rows.
//
// ROW OWNERSHIP.  Wave w owns rows 8w..8w+7 and 8(w+W)..8(w+W)+7 of the slab in every row phase: it stages
// rows 8w.. of the next rotation's A slab, runs their y-forward, later the y-inverse of both row sets, and copies
// exactly those rows out to global memory.  Everything between the two column phases of consecutive rotations is
// therefore wave-local (the FFT passes are wave-local anyway, dlpd_fft.h): TWO block barriers per slab -- before
// and after the column phase, which needs all rows -- instead of five, and the waves of a block drift apart through
// the row phases, one wave's global stores and LDS traffic overlapping another's butterflies.
// The next rotation's A rows and this slab's receptor values are prefetched into registers while the current
// passes run (plain global loads stay in flight across barriers).
//
// What bounds it (round-2 measurements, N = 128, 16 rotations x 49 channels): the kernel with the FFT phases
// removed (staging, barriers, copy-out only) streams its 8.8 GB in 1.35 ms = 6.5 TB/s; the row phases alone add
// 0.07 ms to that (wave-local, hidden behind the memory stream), the column phase alone 0.67 ms, both together
// 1.43 ms.  With every global access removed the kernel still takes 2.24 ms: the butterflies (about 2.0 MFLOP per
// slab on a VALU that retires ~110 f32 results per clock and CU, packed or not) and the LDS exchanges (~13 k LDS
// cycles per slab) add up rather than overlap; the memory stream hides behind them.
// DIRECT OUT (N = 128): the last inverse-y pass deals its butterflies 2t, 2t+1 to thread t, so a lane ends up with
// adjacent output pairs (16 bytes) and the 8 lanes of a pencil with a full 128-byte line: the rows go to global
// memory from the butterfly registers, without the copy-out trip through the slab (2.84 -> 2.74 ms).  The same
// idea with the standard dealing and a DPP lane-pair exchange was slower (3.02 vs 2.78 ms: the exchange is VALU).
// ------------------------------------------------------------------------------------------
#ifdef DLPD_STAMPS
__device__ unsigned long long dlpd_stamps_k2[16];
extern "C" int dlpd_debug_read_stamps_k2(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(dlpd_stamps_k2), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k2), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
#ifndef DLPD_K2_DIRECT_OUT
#define DLPD_K2_DIRECT_OUT 1
#endif
#define DLPD_K2_THREADS(N) ((N) * 4)               // N/16 waves; each owns 8 pencils per step (wave-local FFT passes)
// ROW-GROUP SHIFT (round 6).  Row `row` of the slab starts at row*RS + 8 * ((row / R1) & 1): the rows of every other group
// of R1 rows sit 8 elements to the right, in the 8 spare elements of the row stride.  Why: the first column pass stores
// rows R1*t + r, and the two t of a 16-lane ds_write_b64 group are R1 rows apart -- any row stride times 16 is 0 (mod 16
// slots), so unshifted both t hit the same 8 slots (the kernel's one bank conflict: SQ_LDS_BANK_CONFLICT = 18 % of its
// LDS-active cycles through round 5); shifted, the odd t use the other 8.  Every other access of the kernel touches rows
// of ONE group parity per instruction (row phases: a wave's 8 rows lie in one group; column passes: rows t + 8r and
// j + R1*r differ below bit log2(R1) only), so its conflict-free pattern moves as a whole.  The shift is part of the row
// base (row phases) or of the compile-time offset (column passes): no instruction, no register.  64 bytes: 16-byte pair
// accesses stay aligned.  Off: -DDLPD_K2_ROWSHIFT=0 (variant builds).
#ifndef DLPD_K2_ROWSHIFT
#define DLPD_K2_ROWSHIFT 1
#endif
template <int RS, int GS, int GX> struct ColAddrG {
  static constexpr bool IS_ROW = false;
  int base;      // swz(col)
  DLPD_HD int operator()(int e) const { return e * RS + ((e >> GS) & 1) * GX + base; }
};
template <int N, int MODE> __global__ void __launch_bounds__(DLPD_K2_THREADS(N))
k_xy_corr(const cplx* __restrict__ A, const cplx* __restrict__ rec, cplx* __restrict__ out,
          int CT, int nb, int nsplit, long long rec_bstride, float scale, int transposed) {
  constexpr int L = N / 2, NZ = N / 2 + 1, RS = N + 8;
  constexpr int T = 8, R1 = FftPlanW<N>::R1, R2 = FftPlanW<N>::R2;
  static_assert(RS % 16 == 8, "row stride must be an odd multiple of 8 elements (bank spreading)");
  constexpr int NT = DLPD_K2_THREADS(N), W = NT / 64;
  constexpr int NSET = N / 8;                      // pencil sets (8 pencils) per direction
  static_assert(NSET == 2 * W && L / 8 == W, "row ownership: one forward and two inverse row sets per wave");
  constexpr int NA4 = 8 * L / 2;                   // float4 (2 complex) in a wave's 8 A rows
  constexpr int NLD = (NA4 + 63) / 64;             // ... per lane (the last round partly idle when L = 40)
  constexpr int NST = (8 * N / 2) / 64;            // float4 per lane of 8 output rows
  static_assert((8 * N / 2) % 64 == 0, "whole waves per output row set");
  typedef FftPassW<N, R1, 1, -1, T, L> FwdP1;      // pruned: only the first L inputs are non-zero
  typedef FftPassW<N, R2, R1, -1, T> FwdP2;
  typedef FftPassW<N, R1, 1, +1, T> InvP1;
  typedef FftPassW<N, R2, R1, +1, T> InvP2;
  typedef FftPassW<N, R2, R1, +1, T, N, 1> InvP2A;
  constexpr bool DIRECT_OUT = DLPD_K2_DIRECT_OUT && MODE == 1 && InvP2::FULL && InvP2::PER == 2;
  // register hand-over forward-x pass 2 -> inverse-x pass 1 (power-of-two plans only)
  constexpr bool HANDOVER = InvP1::PER == 1 && InvP1::NBF == T && (R1 % T == 0) && FwdP2::NBF <= R1;
  // row-group shift (above): power-of-two plans only (N = 80 keeps its blocked intermediates instead)
  constexpr int GX = (DLPD_K2_ROWSHIFT && N != 80) ? 8 : 0, GS = (R1 == 16) ? 4 : 3;
  static_assert(GX == 0 || ((1 << GS) == R1 && R1 >= 8 && RS >= N + GX), "groups of R1 rows, a wave's 8 rows in one group");
  DLPD_DYN_SHARED(cplx, S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid of NZ*CT*nsplit blocks.  The batch is cut into nsplit parts handled by blocks whose
  // ids differ by 8 (same XCD under round-robin dispatch, close in time), so the receptor slab
  // they share is served by that XCD's L2 instead of crossing the fabric once per rotation.
  // Speed only: any placement gives the same result.
  const int bid = blockIdx.x;
  const int part = (bid >> 3) % nsplit;
  const int slab = (bid / (8 * nsplit)) * 8 + (bid & 7);
  if (slab >= NZ * CT) return;
  const int kz = slab % NZ, c = slab / NZ;
  const int b_beg = (int)(((long long)nb * part) / nsplit), b_end = (int)(((long long)nb * (part + 1)) / nsplit);
  if (b_beg >= b_end) return;
  // row phase: lane = 8*q + t  (pencil q of the set, thread t); column phase: lane = 8*t + c8
  const int qr = lane >> 3, c8 = lane & 7;
  auto rb = [&](int row) { return row * RS + ((row >> GS) & 1) * GX; };       // start of a slab row
  // Blocked intermediate of the two-pass transforms (FftPassW::store_blk) where it pays: the 10 x 8 plan of N = 80
  // (first-pass stores 104 / 80 -> 40 LDS-array cycles per pencil set); at N = 128 the column stores would go
  // 128 -> 64 by the same model, but the kernel measured 0.1 ms SLOWER with it (2.46 -> 2.56 ms), so it keeps the
  // natural order there.  The slab has 8 spare rows for the column intermediates.
  constexpr bool BLK = (N == 80);
#define K2_P1_STORE(ps, t) do { if constexpr (BLK) (ps).store_blk(S, md, t); else (ps).store(S, ad, t); } while (0)
#define K2_P2_LOAD(ps, t) do { if constexpr (BLK) (ps).load_blk(S, md, t, tw); else (ps).load(S, ad, t, tw); } while (0)
  cplx* tw = S + (N + (BLK ? 8 : 0)) * RS;
  init_twiddles<N>(tw, tid, NT);

  // this wave's 8 rows of a rotation's A slab (L x L complex, [x][y], or [y][x] when K1 stored it transposed:
  // dlpd_corr.hip, slab orientation): NLD float4 per lane
  float4 apref[NLD];
  auto fetch_rows = [&](int bb) {
    const cplx* a = A + (((size_t)bb * CT + c) * NZ + kz) * L * L;
    if (!transposed) {
      // rows 8w..8w+7 are 8*L contiguous complex: lane reads float4 number lane + 64 j of that run
      const float4* a4 = reinterpret_cast<const float4*>(a + (size_t)wave * 8 * L);
#pragma unroll
      for (int j = 0; j < NLD; j++)
        if (NA4 % 64 == 0 || lane + 64 * j < NA4) apref[j] = DLPD_LOAD_STREAM(a4 + lane + 64 * j);
    } else {
      // stored [y][x]: the wave's rows x = 8w..8w+7 are a 64-byte run in every y line
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int f = lane + 64 * j, y = f >> 2, p = f & 3;
        if (NA4 % 64 == 0 || f < NA4)
          apref[j] = DLPD_LOAD_STREAM(reinterpret_cast<const float4*>(a + (size_t)y * L + wave * 8 + 2 * p));
      }
    }
  };
  auto stage_rows = [&]() {                        // apref -> LDS rows 8w..8w+7, columns 0..L-1
    if (!transposed) {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int f = lane + 64 * j, row = wave * 8 + f / (L / 2), col = 2 * (f % (L / 2));
        if (NA4 % 64 == 0 || f < NA4) slab_store_pair(S + rb(row), col, apref[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int f = lane + 64 * j, y = f >> 2, row = wave * 8 + 2 * (f & 3);
        if (NA4 % 64 == 0 || f < NA4) {
          S[rb(row) + slab_swz(y)] = c_make(apref[j].x, apref[j].y);
          S[rb(row + 1) + slab_swz(y)] = c_make(apref[j].z, apref[j].w);
        }
      }
    }
  };
  auto forward_rows = [&]() {                      // y-forward of the wave's rows 8w..8w+7 (wave-local)
    const RowAddr<RS> ad = {rb(wave * 8 + qr)};
    const RowMid<R1 + 1> md = {ad.base};           // blocked intermediate (FftPassW::store_blk)
    const int tr = lane & 7;
    {
      FwdP1 ps;
      ps.load(S, ad, tr, nullptr);
      DLPD_WAVE_SYNC();
      K2_P1_STORE(ps, tr);
      DLPD_WAVE_SYNC();
    }
    {
      FwdP2 ps;
      K2_P2_LOAD(ps, tr);
      DLPD_WAVE_SYNC();
      ps.store(S, ad, tr);
    }
  };
  // rows 8*set..8*set+7 of the slab -> global (N contiguous complex each, the 8 rows one contiguous run)
  auto copy_rows_out = [&](int set, int bb) {
    float4* o = reinterpret_cast<float4*>(out + (((size_t)bb * CT + c) * NZ + kz) * N * N + (size_t)set * 8 * N);
    const float sc = (MODE == 0) ? scale : 1.0f;
#pragma unroll
    for (int j = 0; j < NST; j++) {
      const int f = lane + 64 * j, row = set * 8 + f / (N / 2), col = 2 * (f % (N / 2));
      const float4 v = slab_load_pair(S + rb(row), col);
      DLPD_STORE_STREAM(o + f, make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc));
    }
  };

  cplx rv[FwdP2::PER][R2];
  auto fetch_rec = [&](int bb, int set) {
    const cplx* rbase = rec + (size_t)bb * rec_bstride + ((size_t)c * NZ + kz) * N * N;
    int col = set * 8 + c8;
    DLPD_OPAQUE(col);                  // keeps the 16 load addresses from being hoisted out of the batch loop (spills)
    const int tc = lane >> 3;
    FwdP2 idx;
#pragma unroll
    for (int i = 0; i < FwdP2::PER; i++)
      if (idx.active(i, tc)) {
#pragma unroll
        for (int q = 0; q < R2; q++) rv[i][q] = rbase[(unsigned)(idx.out_index(i, q, tc) * N) + (unsigned)col];
      }
  };
  fetch_rows(b_beg);
  if (MODE == 1) fetch_rec(b_beg, wave);
  __syncthreads();                                         // twiddle table visible
  stage_rows();
  DLPD_WAVE_SYNC();
  forward_rows();
  DLPD_STAMP_DECL;
  for (int b = b_beg; b < b_end; b++) {
    DLPD_STAMP(7);
    __syncthreads();                                       // all rows y-transformed (and the previous slab copied out)
    DLPD_STAMP(1);
    // ---- columns: forward x, receptor multiply, inverse x -- all inside one wave per set
#pragma unroll 1
    for (int set = wave; set < NSET; set += W) {
      const int col = set * 8 + c8;
      const ColAddrG<RS, GS, GX> ad = {slab_swz(col)};
      const int tc = lane >> 3;
      // receptor values: the first set's were requested in the row phase (ahead of the second row set's output
      // stores), the second set's are requested here, before its first x pass
      if (MODE == 1 && set != wave) fetch_rec(b, set);
      // first passes leave their outputs in blocks R1 + 1 rows apart (FftPassW::store_blk), the second passes read them there
      const ColMid<RS, R1 + 1> md = {ad.base};
      {
        FwdP1 ps;
        ps.load(S, ad, tc, nullptr);
        DLPD_WAVE_SYNC();
        K2_P1_STORE(ps, tc);
        DLPD_WAVE_SYNC();
      }
      if (MODE == 0) {
        FwdP2 ps;
        K2_P2_LOAD(ps, tc);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      } else if (!HANDOVER) {
        {
          FwdP2 ps;
          K2_P2_LOAD(ps, tc);
#pragma unroll
          for (int i = 0; i < FwdP2::PER; i++)
#pragma unroll
            for (int q = 0; q < R2; q++) ps.v[i][q] = c_mulc(rv[i][q], ps.v[i][q]);
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tc);
          DLPD_WAVE_SYNC();
        }
        {
          InvP1 ps;
          ps.load(S, ad, tc, nullptr);
          DLPD_WAVE_SYNC();
          K2_P1_STORE(ps, tc);
          DLPD_WAVE_SYNC();
        }
        InvP2 ps;
        K2_P2_LOAD(ps, tc);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      } else {
        InvP1 qs;
        {
          FwdP2 ps;
          K2_P2_LOAD(ps, tc);
          // thread t owns kx = t + i*T + q*R1; the inverse radix-R1 butterfly j = t wants input r1
          // at kx = t + r1*T  ->  r1 = (i*T + q*R1) / T : a pure register renaming
#pragma unroll
          for (int i = 0; i < FwdP2::PER; i++)
#pragma unroll
            for (int q = 0; q < R2; q++) qs.v[0][(i * T + q * R1) / T] = c_mulc(rv[i][q], ps.v[i][q]);
        }
        SmallDft<R1, +1>::run(qs.v[0]);
        DLPD_WAVE_SYNC();
        K2_P1_STORE(qs, tc);
        DLPD_WAVE_SYNC();
        InvP2 ps;
        K2_P2_LOAD(ps, tc);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      }
    }
    DLPD_STAMP(3);
    __syncthreads();                                       // all columns done
    DLPD_STAMP(1);
    // ---- the wave's own rows from here to the next column phase.  Order (vmcnt counts loads and stores together, in
    // issue order): next rotation's A rows requested; y-inverse + output of row set w; staging + y-forward of the next
    // slab's rows 8w.. (the wait for the A rows has only set w's stores behind it); the receptor values of the next
    // column phase's first set requested; y-inverse + output of row set w + W.  No load is ever waited for behind
    // stores younger than one transform phase: with all stores ahead of the receptor loads the column phase used to
    // wait for the write-back of the whole slab (A loads 0.25 ms, receptor loads 0.15 ms of 2.56: variant builds).
    auto inverse_rows_out = [&](int set) {
      if (MODE == 1) {
        const RowAddr<RS> ad = {rb(set * 8 + qr)};
        const int tr = lane & 7;
        const RowMid<R1 + 1> md = {ad.base};
        {
          InvP1 ps;
          ps.load(S, ad, tr, nullptr);
          DLPD_WAVE_SYNC();
          K2_P1_STORE(ps, tr);
          DLPD_WAVE_SYNC();
        }
        if constexpr (DIRECT_OUT) {
          // last pass straight to global memory: with the butterflies dealt 2t, 2t+1 a lane holds adjacent pairs
          // (16 bytes) and the 8 lanes of a pencil a full 128-byte line of the output row -- no copy-out trip
          // through the slab.  The pencils of the set are dealt over the lanes so that the 16-byte reads are
          // bank-conflict free (lane groups of ds_read_b128, MI355X_MICROARCH.md).
          const int t2 = lane & 7, q2 = ((lane >> 5) & 1) | (((lane >> 3) & 3) << 1);
          InvP2A ps;
          ps.load_pairs(S + rb(set * 8 + q2), t2, tw);
          float4* o = reinterpret_cast<float4*>(out + (((size_t)b * CT + c) * NZ + kz) * N * N + (size_t)(set * 8 + q2) * N);
#pragma unroll
          for (int r = 0; r < R2; r++)
            DLPD_STORE_STREAM(o + t2 + r * (InvP2A::NBF / 2),
                              make_float4(ps.v[0][r].x, ps.v[0][r].y, ps.v[1][r].x, ps.v[1][r].y));
        } else {
          InvP2 ps;
          K2_P2_LOAD(ps, tr);
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tr);
        }
        DLPD_WAVE_SYNC();
      }
      DLPD_STAMP(4);
      if constexpr (!DIRECT_OUT) {
        copy_rows_out(set, b);
        DLPD_WAVE_SYNC();
      }
      DLPD_STAMP(5);
    };
    if (b + 1 < b_end) fetch_rows(b + 1);
    inverse_rows_out(wave);
    if (b + 1 < b_end) {
      stage_rows();
      DLPD_WAVE_SYNC();
      DLPD_STAMP(0);
      forward_rows();
      DLPD_STAMP(2);
      if (MODE == 1) fetch_rec(b + 1, wave);
    }
    inverse_rows_out(wave + W);
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k2, DLPD_STAMPS);
}
#undef K2_P1_STORE
#undef K2_P2_LOAD

// batch split of the persistent K2 blocks (variant builds: -DDLPD_K2_NSPLIT=n)
#ifndef DLPD_K2_NSPLIT
#define DLPD_K2_NSPLIT 0
#endif
static constexpr int k2_nsplit_override() { return DLPD_K2_NSPLIT; }

template <int N, int MODE> static int launch_k2(const cplx* A, const cplx* rec, cplx* out, int CT, int nb,
                                                long long rbs, float scale, hipStream_t st, int transposed = 0) {
  constexpr int NZ = N / 2 + 1, RS = N + 8;
  const size_t shmem = (size_t)((N + (N == 80 ? 8 : 0)) * RS + N) * sizeof(cplx);
  int rc = dlpd_set_max_dyn_shared((const void*)k_xy_corr<N, MODE>, shmem);
  if (rc) return rc;
  int nsplit = (MODE == 1 && nb >= 8) ? 2 : 1;
  if (k2_nsplit_override()) nsplit = k2_nsplit_override();
  const int slabs8 = ((NZ * CT + 7) / 8) * 8;
  dim3 grid(slabs8 * nsplit), block(DLPD_K2_THREADS(N));
  DLPD_LAUNCH((k_xy_corr<N, MODE>), grid, block, shmem, st, A, rec, out, CT, nb, nsplit, rbs, scale, transposed);
  return dlpd_check_launch();
}


// ------------------------------------------------------------------------------------------
// Split path for grids whose N x N complex slab does not fit the 160 KB LDS (N = 160, the
// reference's box_size 80): the same per-slab transform as k_xy_corr in three tile kernels that
// work IN PLACE on the output slab (rows 0..L-1 hold the y-forward result in between), so no
// extra workspace is needed.  Plain barrier-separated passes (FftPass); correctness path for the
// reference's real model shapes, not tuned.
// ------------------------------------------------------------------------------------------
template <int N, int RT> __global__ void __launch_bounds__(RT * FftPlan<N>::T)
k2s_fwd_y(const cplx* __restrict__ A, cplx* __restrict__ B) {
  constexpr int L = N / 2, RS = N + 1;
  constexpr int T = FftPlan<N>::T, R1 = FftPlan<N>::R1, R2 = FftPlan<N>::R2, NT = RT * T;
  __shared__ cplx S[RT * RS + N];
  cplx* tw = S + RT * RS;
  const int tid = threadIdx.x, r0 = blockIdx.y * RT;
  const size_t slab = blockIdx.x;
  init_twiddles<N>(tw, tid, NT);
  const cplx* a = A + slab * L * L + (size_t)r0 * L;
  for (int i = tid; i < RT * L; i += NT) S[(i / L) * RS + (i % L)] = a[i];
  __syncthreads();
  const int p = tid % RT, t = tid / RT;
  {
    FftPass<N, R1, 1, -1, T, L> ps;
    ps.load(S + p * RS, 1, t, tw);
    __syncthreads();
    ps.store(S + p * RS, 1, t);
    __syncthreads();
  }
  {
    FftPass<N, R2, R1, -1, T> ps;
    ps.load(S + p * RS, 1, t, tw);
    __syncthreads();
    ps.store(S + p * RS, 1, t);
    __syncthreads();
  }
  cplx* o = B + slab * N * N + (size_t)r0 * N;
  for (int i = tid; i < RT * N; i += NT) o[i] = S[(i / N) * RS + (i % N)];
}

template <int N, int MODE, int CW> __global__ void __launch_bounds__(CW * FftPlan<N>::T)
k2s_cols(cplx* __restrict__ B, const cplx* __restrict__ rec, int CT, long long rec_bstride, float scale) {
  constexpr int L = N / 2, NZ = N / 2 + 1, RSC = CW + 1;
  constexpr int T = FftPlan<N>::T, R1 = FftPlan<N>::R1, R2 = FftPlan<N>::R2, NT = CW * T;
  DLPD_DYN_SHARED(cplx, S);
  cplx* tw = S + N * RSC;
  const int tid = threadIdx.x, c0 = blockIdx.y * CW;
  const size_t slab = blockIdx.x;
  const int kz = (int)(slab % NZ), c = (int)((slab / NZ) % CT), b = (int)(slab / ((size_t)NZ * CT));
  init_twiddles<N>(tw, tid, NT);
  cplx* bs = B + slab * N * N + c0;
  for (int i = tid; i < L * CW; i += NT) S[(i / CW) * RSC + (i % CW)] = bs[(size_t)(i / CW) * N + (i % CW)];
  __syncthreads();
  const int p = tid % CW, t = tid / CW;
  {
    FftPass<N, R1, 1, -1, T, L> ps;
    ps.load(S + p, RSC, t, tw);
    __syncthreads();
    ps.store(S + p, RSC, t);
    __syncthreads();
  }
  {
    FftPass<N, R2, R1, -1, T> ps;
    ps.load(S + p, RSC, t, tw);
    __syncthreads();
    if (MODE == 1) {
      const cplx* r = rec + (size_t)b * rec_bstride + ((size_t)c * NZ + kz) * N * N + c0 + p;
#pragma unroll
      for (int i = 0; i < ps.PER; i++)
        if (ps.active(i, t)) {
#pragma unroll
          for (int q = 0; q < R2; q++) ps.v[i][q] = c_mulc(r[(size_t)ps.out_index(i, q, t) * N], ps.v[i][q]);
        }
    }
    ps.store(S + p, RSC, t);
    __syncthreads();
  }
  if (MODE == 1) {
    {
      FftPass<N, R1, 1, +1, T> ps;
      ps.load(S + p, RSC, t, tw);
      __syncthreads();
      ps.store(S + p, RSC, t);
      __syncthreads();
    }
    {
      FftPass<N, R2, R1, +1, T> ps;
      ps.load(S + p, RSC, t, tw);
      __syncthreads();
      ps.store(S + p, RSC, t);
      __syncthreads();
    }
  }
  const float sc = (MODE == 0) ? scale : 1.0f;
  for (int i = tid; i < N * CW; i += NT) {
    const cplx u = S[(i / CW) * RSC + (i % CW)];
    bs[(size_t)(i / CW) * N + (i % CW)] = c_make(u.x * sc, u.y * sc);
  }
}

template <int N, int RT> __global__ void __launch_bounds__(RT * FftPlan<N>::T)
k2s_inv_y(cplx* __restrict__ B) {
  constexpr int RS = N + 1;
  constexpr int T = FftPlan<N>::T, R1 = FftPlan<N>::R1, R2 = FftPlan<N>::R2, NT = RT * T;
  __shared__ cplx S[RT * RS + N];
  cplx* tw = S + RT * RS;
  const int tid = threadIdx.x, r0 = blockIdx.y * RT;
  const size_t slab = blockIdx.x;
  init_twiddles<N>(tw, tid, NT);
  cplx* o = B + slab * N * N + (size_t)r0 * N;
  for (int i = tid; i < RT * N; i += NT) S[(i / N) * RS + (i % N)] = o[i];
  __syncthreads();
  const int p = tid % RT, t = tid / RT;
  {
    FftPass<N, R1, 1, +1, T> ps;
    ps.load(S + p * RS, 1, t, tw);
    __syncthreads();
    ps.store(S + p * RS, 1, t);
    __syncthreads();
  }
  {
    FftPass<N, R2, R1, +1, T> ps;
    ps.load(S + p * RS, 1, t, tw);
    __syncthreads();
    ps.store(S + p * RS, 1, t);
    __syncthreads();
  }
  for (int i = tid; i < RT * N; i += NT) o[i] = S[(i / N) * RS + (i % N)];
}

template <int N, int MODE> static int launch_k2_split(const cplx* A, const cplx* rec, cplx* out, int CT, int nb,
                                                      long long rbs, float scale, hipStream_t st) {
  constexpr int L = N / 2, NZ = N / 2 + 1, RT = 16, CW = 32, T = FftPlan<N>::T;
  static_assert(L % RT == 0 && N % RT == 0 && N % CW == 0, "tile sizes must divide the grid");
  const unsigned nslab = (unsigned)nb * CT * NZ;
  DLPD_LAUNCH((k2s_fwd_y<N, RT>), dim3(nslab, L / RT), dim3(RT * T), 0, st, A, out);
  const size_t shmem = (size_t)(N * (CW + 1) + N) * sizeof(cplx);
  int rc = dlpd_set_max_dyn_shared((const void*)k2s_cols<N, MODE, CW>, shmem);
  if (rc) return rc;
  DLPD_LAUNCH((k2s_cols<N, MODE, CW>), dim3(nslab, N / CW), dim3(CW * T), shmem, st, out, rec, CT, rbs, scale);
  if (MODE == 1) DLPD_LAUNCH((k2s_inv_y<N, RT>), dim3(nslab, N / RT), dim3(RT * T), 0, st, out);
  return dlpd_check_launch();
}

// ------------------------------------------------------------------------------------------
// K2 for grids whose N x N slab does not fit LDS (N = 160, 205 KB), as FOUR N/2 x N/2 problems: decimation in
// frequency along x AND y.  With kx = 2m + p, ky = 2n + q
//     F[2m+p][2n+q] = FFT2_{N/2}( a[x][y] w^(p x + q y) )[m][n]                       (w = exp(-2 pi i / N); a is L x L, L = N/2)
//     out[u + L s][v + L r] = sum_pq (-1)^(p s + q r) conj(w)^(p u + q v) G_pq[u][v],  G_pq = IFFT2_{N/2}( rec[2m+p][2n+q] conj(F_pq) )
// so every transform is a dense 80-point one (two wave-local passes, one LDS exchange; the length-160 columns of the
// round-2 half-slab kernel -- decimation along y only, G_0 parked in the output slab -- needed three passes: 3.86 ms
// against 3.63 ms for 17 channels x 16 rotations) and two sub-problems (q = 0, 1 of one p) sit in LDS side by side: every
// phase has 20 pencil sets for the block's waves instead of 10.  The q-combination H_p[u][y'] is formed when the pair
// is done; H_0 waits in registers (L*N/NT complex per thread) for H_1 and the output slab is written once.  The last
// forward x pass (radix 8, butterfly j holds kx = j + 10 r) hands its registers to the first inverse x pass, which is
// a radix-8 pass over exactly those inputs when the inverse runs the plan 8 x 10.
//   grid NZ*CT*nsplit (XCD-aware decode as above), block 64*WV threads, persistent over its part of the batch.
// ------------------------------------------------------------------------------------------
template <int N, int WV> __global__ void __launch_bounds__(64 * WV)
k_xy_corr_quad(const cplx* __restrict__ A, const cplx* __restrict__ rec, cplx* __restrict__ out,
               int CT, int nb, int nsplit, long long rec_bstride, int transposed) {
  constexpr int L = N / 2, H = N / 2, NZ = N / 2 + 1, RS = H + 8;
  constexpr int HR = H + 10, SUB = HR * RS;            // rows per sub-slab (10 spare: blocked intermediates of the column
                                                       // passes, FftPassW::store_blk: 8 x 11 forward, 10 x 9 inverse)
  static_assert(RS % 16 == 8, "row stride must be an odd multiple of 8 elements (bank spreading)");
  constexpr int NT = 64 * WV, W = WV;
  constexpr int NP = (L * L / 2 + NT - 1) / NT;        // element pairs (float4) per thread of an L x L slab
  constexpr int NSET = 2 * (H / 8);                    // pencil sets per phase: both sub-problems
  typedef FftPlanW<H> P;
  typedef FftPassW<H, P::R1, 1, -1, 8> FwdP1;
  typedef FftPassW<H, P::R2, P::R1, -1, 8> FwdP2;
  typedef FftPassW<H, P::R2, 1, +1, 8> InvP1;          // inverse columns run R2 x R1: pass 1 consumes FwdP2's registers
  typedef FftPassW<H, P::R1, P::R2, +1, 8> InvP2;
  static_assert(FwdP2::PER == InvP1::PER && FwdP2::NBF == InvP1::NBF, "register hand-over forward -> inverse");
  DLPD_DYN_SHARED(cplx, S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int part = (bid >> 3) % nsplit;
  const int slab = (bid / (8 * nsplit)) * 8 + (bid & 7);
  if (slab >= NZ * CT) return;
  const int kz = slab % NZ, c = slab / NZ;
  const int b_beg = (int)(((long long)nb * part) / nsplit), b_end = (int)(((long long)nb * (part + 1)) / nsplit);
  if (b_beg >= b_end) return;
  const int tr = lane & 7, qr = lane >> 3;             // row phase: lane = 8*pencil + thread
  const int c8 = lane & 7;                             // column phase: lane = 8*thread + column
  cplx* tw = S + 2 * SUB;                              // exp(-2 pi i k / N)
  cplx* twh = tw + N;                                  // exp(-2 pi i k / H)
  init_twiddles<N>(tw, tid, NT);
  init_twiddles<H>(twh, tid, NT);

  float4 apref[NP];
  auto fetch_A = [&](int b) {
    const float4* a = reinterpret_cast<const float4*>(A + (((size_t)b * CT + c) * NZ + kz) * L * L);
#pragma unroll
    for (int i = 0; i < NP; i++)
      if (tid + i * NT < L * L / 2) apref[i] = DLPD_LOAD_STREAM(a + tid + i * NT);
  };
  float4 h0[NP][2];                                    // H_0[u][v..v+1], H_0[u][v+L..]
  fetch_A(b_beg);
  __syncthreads();                                     // twiddle tables visible
  DLPD_STAMP_DECL;
  for (int b = b_beg; b < b_end; b++) {
    float4* o = reinterpret_cast<float4*>(out + (((size_t)b * CT + c) * NZ + kz) * N * N);
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
      // ---- A slab (registers) times w^(p x) -> sub-slab q = 0, times w^y -> sub-slab q = 1
      DLPD_STAMP(7);
      int tq = tid;
      DLPD_OPAQUE(tq);               // keeps the slab offsets below from being hoisted out of the loops and spilled
#pragma unroll
      for (int i = 0; i < NP; i++) {
        const int e = 2 * (tq + i * NT), r0 = e / L, c0 = e % L;       // memory row / column of the pair
        if (e < L * L) {
          cplx u = c_make(apref[i].x, apref[i].y), v = c_make(apref[i].z, apref[i].w);
          if (transposed) {                                  // stored [y][x]: the pair is (x, y) = (c0, r0), (c0 + 1, r0)
            if (p) { u = c_mul(u, tw[c0]); v = c_mul(v, tw[c0 + 1]); }
            const cplx wy = tw[r0];
            cplx* d = S + c0 * RS + slab_swz(r0);
            d[0] = u;
            d[RS] = v;
            d[SUB] = c_mul(u, wy);
            d[SUB + RS] = c_mul(v, wy);
          } else {
            if (p) { const cplx wx = tw[r0]; u = c_mul(u, wx); v = c_mul(v, wx); }
            cplx* d = S + r0 * RS;
            d[slab_swz(c0)] = u;
            d[slab_swz(c0 + 1)] = v;
            d[SUB + slab_swz(c0)] = c_mul(u, tw[c0]);
            d[SUB + slab_swz(c0 + 1)] = c_mul(v, tw[c0 + 1]);
          }
        }
      }
      if (p == 1 && b + 1 < b_end) fetch_A(b + 1);     // next rotation's slab, in flight over this pair
      DLPD_STAMP(0);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- forward along y: all rows of both sub-slabs
#pragma unroll 1
      for (int set = wave; set < NSET; set += W) {
        const RowAddr<RS> ad = {(set / (H / 8)) * SUB + ((set % (H / 8)) * 8 + qr) * RS};
        int t = tr;
        DLPD_OPAQUE(t);
        fft_wave<H, -1, H>(S, ad, t, twh);
      }
      DLPD_STAMP(2);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- columns: forward x, receptor multiply, inverse x (first inverse pass in registers)
#pragma unroll 1
      for (int set = wave; set < NSET; set += W) {
        const int q = set / (H / 8), col = (set % (H / 8)) * 8 + c8;
        const ColAddr<RS> ad = {q * SUB + slab_swz(col)};
        const ColMid<RS, P::R1 + 1> mdf = {ad.base};   // forward 10 x 8: blocks of 11 rows
        const ColMid<RS, P::R2 + 1> mdi = {ad.base};   // inverse 8 x 10: blocks of 9 rows
        int tc = lane >> 3;
        DLPD_OPAQUE(tc);
        const cplx* rbase = rec + (size_t)b * rec_bstride + (((size_t)c * NZ + kz) * N + p) * N + (2 * col + q);
        cplx rv[FwdP2::PER][P::R2];
        {
          FwdP2 idx;                                   // receptor values: requested first, in flight during the first pass
#pragma unroll
          for (int i = 0; i < FwdP2::PER; i++)
            if (idx.active(i, tc)) {
#pragma unroll
              for (int r = 0; r < P::R2; r++) rv[i][r] = rbase[(unsigned)(idx.out_index(i, r, tc) * 2 * N)];
            }
        }
        {
          FwdP1 ps;
          ps.load(S, ad, tc, twh);
          DLPD_WAVE_SYNC();
          ps.store_blk(S, mdf, tc);
          DLPD_WAVE_SYNC();
        }
        InvP1 qs;
        {
          FwdP2 ps;
          ps.load_blk(S, mdf, tc, twh);
#pragma unroll
          for (int i = 0; i < FwdP2::PER; i++)
            if (ps.active(i, tc)) {
#pragma unroll
              for (int r = 0; r < P::R2; r++) qs.v[i][r] = c_mulc(rv[i][r], ps.v[i][r]);
              SmallDft<P::R2, +1>::run(qs.v[i]);
            }
        }
        DLPD_WAVE_SYNC();
        qs.store_blk(S, mdi, tc);
        DLPD_WAVE_SYNC();
        InvP2 ps;
        ps.load_blk(S, mdi, tc, twh);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      }
      DLPD_STAMP(3);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- inverse along y -> G_p0, G_p1
#pragma unroll 1
      for (int set = wave; set < NSET; set += W) {
        const RowAddr<RS> ad = {(set / (H / 8)) * SUB + ((set % (H / 8)) * 8 + qr) * RS};
        int t = tr;
        DLPD_OPAQUE(t);
        fft_wave<H, +1, H>(S, ad, t, twh);
      }
      DLPD_STAMP(4);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- H_p[u][v + L r] = G_p0[u][v] + (-1)^r conj(w^v) G_p1[u][v];  out[u + L s][y'] = H_0 + (-1)^s conj(w^u) H_1
      DLPD_OPAQUE(tq);
#pragma unroll
      for (int i = 0; i < NP; i++) {
        const int e = 2 * (tq + i * NT), u = e / H, v = e % H;
        if (e < H * H) {
          const cplx* g = S + u * RS;
          const cplx a0 = g[slab_swz(v)], a1 = g[slab_swz(v + 1)];
          const cplx b0 = c_mulc(g[SUB + slab_swz(v)], tw[v]), b1 = c_mulc(g[SUB + slab_swz(v + 1)], tw[v + 1]);
          const float4 lo = make_float4(a0.x + b0.x, a0.y + b0.y, a1.x + b1.x, a1.y + b1.y);
          const float4 hi = make_float4(a0.x - b0.x, a0.y - b0.y, a1.x - b1.x, a1.y - b1.y);
          if (p == 0) {
            h0[i][0] = lo;
            h0[i][1] = hi;
          } else {
            const cplx wu = tw[u];
            const cplx l0 = c_mulc(c_make(lo.x, lo.y), wu), l1 = c_mulc(c_make(lo.z, lo.w), wu);
            const cplx k0 = c_mulc(c_make(hi.x, hi.y), wu), k1 = c_mulc(c_make(hi.z, hi.w), wu);
            const float4 f = h0[i][0], k = h0[i][1];
            DLPD_STORE_STREAM(o + (u * N + v) / 2, make_float4(f.x + l0.x, f.y + l0.y, f.z + l1.x, f.w + l1.y));
            DLPD_STORE_STREAM(o + (u * N + v + H) / 2, make_float4(k.x + k0.x, k.y + k0.y, k.z + k1.x, k.w + k1.y));
            DLPD_STORE_STREAM(o + ((u + H) * N + v) / 2, make_float4(f.x - l0.x, f.y - l0.y, f.z - l1.x, f.w - l1.y));
            DLPD_STORE_STREAM(o + ((u + H) * N + v + H) / 2, make_float4(k.x - k0.x, k.y - k0.y, k.z - k1.x, k.w - k1.y));
          }
        }
      }
      DLPD_STAMP(5);
      __syncthreads();                                 // sub-slabs fully read before they are refilled
      DLPD_STAMP(1);
    }
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k2, DLPD_STAMPS);
}

#ifndef DLPD_K2Q_WAVES
#define DLPD_K2Q_WAVES 8
#endif
template <int N, int WV> static int launch_k2_quad(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, long long rbs,
                                                   hipStream_t st, int transposed = 0) {
  constexpr int NZ = N / 2 + 1, H = N / 2, RS = H + 8;
  const size_t shmem = (size_t)(2 * (H + 10) * RS + N + H) * sizeof(cplx);
  int rc = dlpd_set_max_dyn_shared((const void*)k_xy_corr_quad<N, WV>, shmem);
  if (rc) return rc;
  int nsplit = nb >= 8 ? 2 : 1;
  if (k2_nsplit_override()) nsplit = k2_nsplit_override();
  const int slabs8 = ((NZ * CT + 7) / 8) * 8;
  DLPD_LAUNCH((k_xy_corr_quad<N, WV>), dim3(slabs8 * nsplit), dim3(64 * WV), shmem, st, A, rec, out, CT, nb, nsplit, rbs, transposed);
  return dlpd_check_launch();
}

int dlpd_k2_forward(const cplx* A, cplx* out, int CT, int nb, int L, float scale, hipStream_t st) {
  switch (L) {
    case 32: return launch_k2<64, 0>(A, nullptr, out, CT, nb, 0, scale, st);
    case 40: return launch_k2<80, 0>(A, nullptr, out, CT, nb, 0, scale, st);
    case 64: return launch_k2<128, 0>(A, nullptr, out, CT, nb, 0, scale, st);
    case 80: return launch_k2_split<160, 0>(A, nullptr, out, CT, nb, 0, scale, st);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// N = 160 with 4-lane pencils and affine LDS addressing: dlpd_k2q.hip (untransposed slabs)
int dlpd_k2q_correlate(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, int L, long long rbs, int nsplit_override,
                       hipStream_t st, int packed, const unsigned char* pmap = nullptr, int nmasked = 0);
int dlpd_k2q_pack_receptor(const cplx* rec, void* packed, int CT, int L, hipStream_t st);
#ifndef DLPD_K2_Q4
#define DLPD_K2_Q4 1
#endif
#ifndef DLPD_K2_S4
#define DLPD_K2_S4 1
#endif

int dlpd_k2_correlate(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, int L, long long rbs, hipStream_t st,
                      int transposed) {
  if (((DLPD_K2_Q4 && L == 80) || (DLPD_K2_S4 && L == 40)) && !transposed)
    return dlpd_k2q_correlate(A, rec, out, CT, nb, L, rbs, k2_nsplit_override(), st, 0);
  switch (L) {
    case 32: return launch_k2<64, 1>(A, rec, out, CT, nb, rbs, 1.f, st, transposed);
    case 40: return launch_k2<80, 1>(A, rec, out, CT, nb, rbs, 1.f, st, transposed);
    case 64: return launch_k2<128, 1>(A, rec, out, CT, nb, rbs, 1.f, st, transposed);
#ifdef DLPD_TEST_VARIANTS
    case 80: return launch_k2_quad<160, DLPD_K2Q_WAVES>(A, rec, out, CT, nb, rbs, st, transposed);
#endif
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// The receptor spectrum re-ordered for the grids whose K2 re-reads it for every rotation (the 4-lane-pencil kernels of
// dlpd_k2q.hip; the N = 64 / 128 kernels hold their receptor values in registers across the batch and take the natural layout)
long long dlpd_k2_packed_receptor_floats(int CT, int L) {
  if (!((DLPD_K2_Q4 && L == 80) || (DLPD_K2_S4 && L == 40)) || CT <= 0) return 0;
  return (long long)CT * (L + 1) * (2 * L) * (2 * L) * 2;
}
int dlpd_k2_pack_receptor(const cplx* rec, void* packed, int CT, int L, hipStream_t st) {
  if (!dlpd_k2_packed_receptor_floats(CT, L)) return DLPD_ERR_UNSUPPORTED;
  return dlpd_k2q_pack_receptor(rec, packed, CT, L, st);
}
int dlpd_k2_correlate_packed(const cplx* A, const cplx* packed, cplx* out, int CT, int nb, int L, hipStream_t st,
                             const unsigned char* pmap, int nmasked) {
  if (!dlpd_k2_packed_receptor_floats(CT, L)) return DLPD_ERR_UNSUPPORTED;
  return dlpd_k2q_correlate(A, packed, out, CT, nb, L, 0, k2_nsplit_override(), st, 1, pmap, nmasked);
}

// Slabs that K1 stored transposed (dlpd_zfft_oriented, the per-channel K1 of ligands with fewer than 8 channels): every
// compiled grid except N = 160, whose transposed reader is round 2's quad kernel -- kept as a TEST VARIANT
// (-DDLPD_TEST_VARIANTS: tests/variants, never in libdlpd.so); the engine asks and visits box 80 in one orientation.
int dlpd_k2_orientation_supported(int L) {
#ifdef DLPD_TEST_VARIANTS
  return (L == 32 || L == 40 || L == 64 || L == 80) ? 1 : 0;
#else
  return (L == 32 || L == 40 || L == 64) ? 1 : 0;
#endif
}
