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
// 1.43 ms: LDS instruction throughput at 2 waves per SIMD (reads cost about as much as writes there).  Measured
// and rejected: the last inverse-y pass written straight to global memory with a DPP lane-pair exchange so that
// every lane stores 16 bytes and 8 lanes a full 128-byte line (a sixth fewer LDS instructions): 3.02 vs 2.78 ms.
// ------------------------------------------------------------------------------------------
#ifdef DLPD_STAMPS
__device__ unsigned long long dlpd_stamps_k2[16];
extern "C" int dlpd_debug_read_stamps_k2(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(dlpd_stamps_k2), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k2), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
#define DLPD_K2_THREADS(N) ((N) * 4)               // N/16 waves; each owns 8 pencils per step (wave-local FFT passes)
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
  // register hand-over forward-x pass 2 -> inverse-x pass 1 (power-of-two plans only)
  constexpr bool HANDOVER = InvP1::PER == 1 && InvP1::NBF == T && (R1 % T == 0) && FwdP2::NBF <= R1;
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
  cplx* tw = S + N * RS;
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
        if (NA4 % 64 == 0 || f < NA4) slab_store_pair(S + row * RS, col, apref[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int f = lane + 64 * j, y = f >> 2, row = wave * 8 + 2 * (f & 3);
        if (NA4 % 64 == 0 || f < NA4) {
          S[row * RS + slab_swz(y)] = c_make(apref[j].x, apref[j].y);
          S[(row + 1) * RS + slab_swz(y)] = c_make(apref[j].z, apref[j].w);
        }
      }
    }
  };
  auto forward_rows = [&]() {                      // y-forward of the wave's rows 8w..8w+7 (wave-local)
    const RowAddr<RS> ad = {(wave * 8 + qr) * RS};
    const int tr = lane & 7;
    {
      FwdP1 ps;
      ps.load(S, ad, tr, nullptr);
      DLPD_WAVE_SYNC();
      ps.store(S, ad, tr);
      DLPD_WAVE_SYNC();
    }
    {
      FwdP2 ps;
      ps.load(S, ad, tr, tw);
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
      const float4 v = slab_load_pair(S + row * RS, col);
      DLPD_STORE_STREAM(o + f, make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc));
    }
  };

  fetch_rows(b_beg);
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
      const ColAddr<RS> ad = {slab_swz(col)};
      const int tc = lane >> 3;
      cplx rv[FwdP2::PER][R2];
      // receptor values: requested before the first x pass, in flight during it
      if (MODE == 1) {
        const cplx* rbase = rec + (size_t)b * rec_bstride + ((size_t)c * NZ + kz) * N * N;
        FwdP2 idx;
#pragma unroll
        for (int i = 0; i < FwdP2::PER; i++)
          if (idx.active(i, tc)) {
#pragma unroll
            for (int q = 0; q < R2; q++) rv[i][q] = rbase[(unsigned)(idx.out_index(i, q, tc) * N) + (unsigned)col];
          }
      }
      {
        FwdP1 ps;
        ps.load(S, ad, tc, nullptr);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
        DLPD_WAVE_SYNC();
      }
      if (MODE == 0) {
        FwdP2 ps;
        ps.load(S, ad, tc, tw);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      } else if (!HANDOVER) {
        {
          FwdP2 ps;
          ps.load(S, ad, tc, tw);
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
          ps.store(S, ad, tc);
          DLPD_WAVE_SYNC();
        }
        InvP2 ps;
        ps.load(S, ad, tc, tw);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      } else {
        InvP1 qs;
        {
          FwdP2 ps;
          ps.load(S, ad, tc, tw);
          // thread t owns kx = t + i*T + q*R1; the inverse radix-R1 butterfly j = t wants input r1
          // at kx = t + r1*T  ->  r1 = (i*T + q*R1) / T : a pure register renaming
#pragma unroll
          for (int i = 0; i < FwdP2::PER; i++)
#pragma unroll
            for (int q = 0; q < R2; q++) qs.v[0][(i * T + q * R1) / T] = c_mulc(rv[i][q], ps.v[i][q]);
        }
        SmallDft<R1, +1>::run(qs.v[0]);
        DLPD_WAVE_SYNC();
        qs.store(S, ad, tc);
        DLPD_WAVE_SYNC();
        InvP2 ps;
        ps.load(S, ad, tc, tw);
        DLPD_WAVE_SYNC();
        ps.store(S, ad, tc);
      }
    }
    DLPD_STAMP(3);
    __syncthreads();                                       // all columns done
    DLPD_STAMP(1);
    // ---- the wave's own rows from here to the next column phase: next rotation's A rows requested now,
    // y-inverse + copy-out of row sets w and w + W, then staging + y-forward of the next slab's rows
    if (b + 1 < b_end) fetch_rows(b + 1);
#pragma unroll 1
    for (int set = wave; set < NSET; set += W) {
      if (MODE == 1) {
        const RowAddr<RS> ad = {(set * 8 + qr) * RS};
        const int tr = lane & 7;
        {
          InvP1 ps;
          ps.load(S, ad, tr, nullptr);
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tr);
          DLPD_WAVE_SYNC();
        }
        {
          InvP2 ps;
          ps.load(S, ad, tr, tw);
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tr);
        }
        DLPD_WAVE_SYNC();
      }
      DLPD_STAMP(4);
      copy_rows_out(set, b);
      DLPD_WAVE_SYNC();
      DLPD_STAMP(5);
    }
    if (b + 1 < b_end) {
      stage_rows();
      DLPD_WAVE_SYNC();
      DLPD_STAMP(0);
      forward_rows();
      DLPD_STAMP(2);
    }
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k2, DLPD_STAMPS);
}

// DLPD_K2_NSPLIT (diagnostic override of the batch split): read once, not on every launch
static int k2_nsplit_override() {
  static const int v = [] { const char* e = getenv("DLPD_K2_NSPLIT"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();
  return v;
}

template <int N, int MODE> static int launch_k2(const cplx* A, const cplx* rec, cplx* out, int CT, int nb,
                                                long long rbs, float scale, hipStream_t st, int transposed = 0) {
  constexpr int NZ = N / 2 + 1, RS = N + 8;
  const size_t shmem = (size_t)(N * RS + N) * sizeof(cplx);
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
// K2 for grids whose N x N slab does not fit LDS (N = 160): decimation in frequency along y.
// With only the first N/2 inputs of a y row non-zero, its even outputs are the N/2-point FFT of the
// row and its odd outputs the N/2-point FFT of the row times w_N^y; likewise the inverse along y is
//     out[n'] = G0[n'] + conj(w_N^n') G1[n'],  out[n' + N/2] = G0[n'] - conj(w_N^n') G1[n'],
// G_p = N/2-point inverse over the parity-p columns.  So one block runs the whole slab as two
// half-width passes over an N x (N/2) LDS slab (110 KB at N = 160): G0 waits in registers
// (N*N/4/NT complex per thread) while parity 1 runs, the A slab stays in registers for both
// parities, and the output is written once, fully coalesced.  Same in/out layout as k_xy_corr.
//   grid NZ*CT*nsplit (XCD-aware decode as above), block 4N = 640 threads (10 waves: one column
//   pencil set each), persistent over the rotations of its part of the batch.
// ------------------------------------------------------------------------------------------
// Measured and rejected: a 10 x 16 two-pass column plan (8 % slower than the wave-local 8 x 4 x 5 despite one LDS round
// trip less); G0 kept in registers instead of parked in the output slab (spills at 8 waves); 5 or 10 waves per block
// (5.6 / 6.1 ms against 4.0 ms at 8: fewer waves hide less latency, ten spill -- also with the receptor values loaded at
// their use instead of prefetched: 5.75 ms at ten waves, 7.2 ms at five).
#define DLPD_K2D_WAVES 8
template <int N, int WV> __global__ void __launch_bounds__(64 * WV)
k_xy_corr_dif(const cplx* __restrict__ A, const cplx* __restrict__ rec, cplx* __restrict__ out,
              int CT, int nb, int nsplit, long long rec_bstride, int transposed) {
  constexpr int L = N / 2, H = N / 2, NZ = N / 2 + 1, RS = H + 8;
  static_assert(RS % 16 == 8, "row stride must be an odd multiple of 8 elements (bank spreading)");
  constexpr int NT = 64 * WV, W = WV;
  constexpr int NLOAD = (L * L / 2 + NT - 1) / NT;     // float4 (2 complex) per thread of an A slab
  constexpr int NG = (N * H / 2 + NT - 1) / NT;        // float4 per thread of a G slab
  typedef FftPlanW<N> P;                          // column (length-N) plan; rows use FftPlanW<H>
  constexpr bool THREE = P::R3 > 1;
  typedef FftPassW<N, P::R1, 1, -1, 8, L> FwdP1;
  typedef FftPassW<N, P::R2, P::R1, -1, 8> FwdP2;
  typedef FftPassW<N, (THREE ? P::R3 : 2), P::R1 * P::R2, -1, 8> FwdP3;
  typedef typename std::conditional<THREE, FwdP3, FwdP2>::type FwdLast;   // the pass that meets the receptor
  constexpr int RL = THREE ? P::R3 : P::R2;
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
  cplx* tw = S + N * RS;                               // exp(-2 pi i k / N)
  cplx* twh = tw + N;                                  // exp(-2 pi i k / H)
  init_twiddles<N>(tw, tid, NT);
  init_twiddles<H>(twh, tid, NT);

  float4 apref[NLOAD];
  auto fetch_A = [&](int b) {
    const float4* a = reinterpret_cast<const float4*>(A + (((size_t)b * CT + c) * NZ + kz) * L * L);
#pragma unroll
    for (int i = 0; i < NLOAD; i++)
      if (tid + i * NT < L * L / 2) apref[i] = DLPD_LOAD_STREAM(a + tid + i * NT);
  };
  fetch_A(b_beg);
  __syncthreads();                                     // twiddle tables visible
  DLPD_STAMP_DECL;
  for (int b = b_beg; b < b_end; b++) {
    const int tr_flag = transposed;                          // slabs stored transposed by K1 (dlpd_corr.hip)
    float4* o = reinterpret_cast<float4*>(out + (((size_t)b * CT + c) * NZ + kz) * N * N);
#pragma unroll 1
    for (int par = 0; par < 2; par++) {
      // ---- A slab (registers) -> rows 0..L-1, times w_N^y for the odd outputs
      DLPD_STAMP(7);
      int tq = tid;
      DLPD_OPAQUE(tq);               // keeps the ~60 slab offsets below from being hoisted out of the loops and spilled
#pragma unroll
      for (int i = 0; i < NLOAD; i++) {
        const int e = 2 * (tq + i * NT), x = e / L, y = e % L;
        if (e < L * L) {
          cplx u = c_make(apref[i].x, apref[i].y), v = c_make(apref[i].z, apref[i].w);
          if (tr_flag) {                                     // stored [y][x]: this pair is (x = y, y = x), (x = y + 1, ..)
            if (par) { u = c_mul(u, tw[x]); v = c_mul(v, tw[x]); }
            S[y * RS + slab_swz(x)] = u;
            S[(y + 1) * RS + slab_swz(x)] = v;
          } else {
            if (par) { u = c_mul(u, tw[y]); v = c_mul(v, tw[y + 1]); }
            S[x * RS + slab_swz(y)] = u;
            S[x * RS + slab_swz(y + 1)] = v;
          }
        }
      }
      if (par == 1 && b + 1 < b_end) fetch_A(b + 1);   // next rotation's slab, in flight over this parity
      DLPD_STAMP(0);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- forward along y: H-point transforms of the L non-zero rows
#pragma unroll 1
      for (int set = wave; set < L / 8; set += W) {
        const RowAddr<RS> ad = {(set * 8 + qr) * RS};
        int t = tr;
        DLPD_OPAQUE(t);
        fft_wave<H, -1, H>(S, ad, t, twh);
      }
      DLPD_STAMP(2);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- columns ky = 2m + par: forward x (pruned), receptor multiply, inverse x
#pragma unroll 1
      for (int set = wave; set < H / 8; set += W) {
        const int col = set * 8 + c8;
        const ColAddr<RS> ad = {slab_swz(col)};
        int tc = lane >> 3;
        DLPD_OPAQUE(tc);
        // receptor values of this pencil set: requested first, in flight during the forward passes
        // receptor values of this pencil set: requested first (in flight during the forward passes) when
        // they fit (three-pass plan: 40 VGPRs), else one butterfly at a time right before their use
        const cplx* rbase = rec + (size_t)b * rec_bstride + ((size_t)c * NZ + kz) * N * N + (2 * col + par);
        cplx rv[THREE ? FwdLast::PER : 1][RL];
        if (THREE) {
          FwdLast idx;
#pragma unroll
          for (int i = 0; i < FwdLast::PER; i++)
            if (idx.active(i, tc)) {
#pragma unroll
              for (int q = 0; q < RL; q++) rv[i][q] = rbase[(unsigned)(idx.out_index(i, q, tc) * N)];
            }
        }
        {
          FwdP1 ps;
          ps.load(S, ad, tc, tw);
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tc);
          DLPD_WAVE_SYNC();
        }
        if (THREE) {
          FwdP2 ps;
          ps.load(S, ad, tc, tw);
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tc);
          DLPD_WAVE_SYNC();
        }
        {
          FwdLast ps;
          ps.load(S, ad, tc, tw);
#pragma unroll
          for (int i = 0; i < FwdLast::PER; i++)
            if (ps.active(i, tc)) {
              if (!THREE) {
#pragma unroll
                for (int q = 0; q < RL; q++) rv[0][q] = rbase[(unsigned)(ps.out_index(i, q, tc) * N)];
              }
#pragma unroll
              for (int q = 0; q < RL; q++) ps.v[i][q] = c_mulc(rv[THREE ? i : 0][q], ps.v[i][q]);
            }
          DLPD_WAVE_SYNC();
          ps.store(S, ad, tc);
          DLPD_WAVE_SYNC();
        }
        fft_wave<N, +1, N, ColAddr<RS>, P>(S, ad, tc, tw);
      }
      DLPD_STAMP(3);
      __syncthreads();
      DLPD_STAMP(1);
      // ---- inverse along y: H-point transforms of all N rows -> G_par
#pragma unroll 1
      for (int set = wave; set < N / 8; set += W) {
        const RowAddr<RS> ad = {(set * 8 + qr) * RS};
        int t = tr;
        DLPD_OPAQUE(t);
        fft_wave<H, +1, H>(S, ad, t, twh);
      }
      DLPD_STAMP(4);
      __syncthreads();
      DLPD_STAMP(1);
      DLPD_OPAQUE(tq);
      if (par == 0) {
#pragma unroll
        for (int i = 0; i < NG; i++) {
          const int e = 2 * (tq + i * NT), x = e / H, y = e % H;
          if (e < N * H) {
            const cplx u = S[x * RS + slab_swz(y)], v = S[x * RS + slab_swz(y + 1)];
            o[(x * N + y) / 2] = make_float4(u.x, u.y, v.x, v.y);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NG; i++) {
          const int e = 2 * (tq + i * NT), x = e / H, y = e % H;
          if (e < N * H) {
            const cplx u = c_mulc(S[x * RS + slab_swz(y)], tw[y]);          // conj(w^y) * G1
            const cplx v = c_mulc(S[x * RS + slab_swz(y + 1)], tw[y + 1]);
            const float4 g = o[(x * N + y) / 2];
            DLPD_STORE_STREAM(o + (x * N + y) / 2, make_float4(g.x + u.x, g.y + u.y, g.z + v.x, g.w + v.y));
            DLPD_STORE_STREAM(o + (x * N + y + H) / 2, make_float4(g.x - u.x, g.y - u.y, g.z - v.x, g.w - v.y));
          }
        }
      }
      DLPD_STAMP(5);
      __syncthreads();                                 // slab fully read before it is refilled
      DLPD_STAMP(1);
    }
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k2, DLPD_STAMPS);
}

template <int N, int WV> static int launch_k2_dif(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, long long rbs,
                                                  hipStream_t st, int transposed = 0) {
  constexpr int NZ = N / 2 + 1, H = N / 2, RS = H + 8;
  const size_t shmem = (size_t)(N * RS + N + H) * sizeof(cplx);
  int rc = dlpd_set_max_dyn_shared((const void*)k_xy_corr_dif<N, WV>, shmem);
  if (rc) return rc;
  int nsplit = nb >= 8 ? 2 : 1;
  if (k2_nsplit_override()) nsplit = k2_nsplit_override();
  const int slabs8 = ((NZ * CT + 7) / 8) * 8;
  DLPD_LAUNCH((k_xy_corr_dif<N, WV>), dim3(slabs8 * nsplit), dim3(64 * WV), shmem, st, A, rec, out, CT, nb, nsplit, rbs, transposed);
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

int dlpd_k2_correlate(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, int L, long long rbs, hipStream_t st,
                      int transposed) {
  switch (L) {
    case 32: return launch_k2<64, 1>(A, rec, out, CT, nb, rbs, 1.f, st, transposed);
    case 40: return launch_k2<80, 1>(A, rec, out, CT, nb, rbs, 1.f, st, transposed);
    case 64: return launch_k2<128, 1>(A, rec, out, CT, nb, rbs, 1.f, st, transposed);
    case 80: return launch_k2_dif<160, DLPD_K2D_WAVES>(A, rec, out, CT, nb, rbs, st, transposed);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}
