// K1, role-split formulation: trilinear rotation of the channels-last ligand + forward z transform (gfx950 / CDNA4).
//
// Reference path being replaced (file:line in /root/reference):
//   src/Docker/Docker.py:218            VolumeRotation of the ligand representation volumes
//   src/Models/DockingModels.py:70-71   the z pass of the per-channel correlation's forward transform
//
// Same inputs, outputs, samples and butterflies as k_rotate_zfft_cl of dlpd_corr.hip (the sample is ONE function,
// dlpd_k1.h; the passes are the same radix plan run wave-locally): the spectra are bit-identical.  There every wave
// gathers, then transforms, then stores, block barrier between the phases, and the three phases overlap only as far as
// the two resident blocks of a CU happen to be out of step.  Here ONE block per CU holds the CU for a whole range of
// work items (rotation b, plane x, YG rows, CC channels) and its waves have FIXED ROLES:
//   * G gather waves: sample addresses, the eight 16-byte channels-last loads of a voxel, the weighted sums, 8-byte LDS
//     stores into the compact INPUT buffer (two real rows per complex pencil, L elements each) -- nothing else, no
//     transform registers;
//   * X transform / store waves: a wave owns 8 pencils (8 lanes each): first pass from the input buffer into the WORK
//     buffer, second pass in place, Hermitian untangle of the wave's OWN pencils and the 128-byte (N = 128) / 64-byte
//     (N = 160) global stores -- no address arithmetic of the gather, no cross-wave hand-over (wave-level ordering only).
// The roles meet at block barriers: INBUF = 2 (N = 128: 2 x 36 KB + 70 KB): one per item -- item i + 1 is gathered into
// the other input buffer while item i is transformed and stored; INBUF = 1 (N = 160: 53 KB + 86 KB, a second input buffer
// does not fit): two per item -- the gather of item i + 1 starts as soon as the first pass of item i has its inputs in
// registers.
//
// MEASURED (round 5, one MI355X, bench.py --k1_form 1 / 2, twice each): 48 ch x 64^3 K1 1.14 against 0.82-0.85 ms,
// the reference's real shapes 0.645 against 0.60-0.61, 48 ch x 80^3 1.91 against 1.91.  In-kernel stamps (48 x 64^3): a
// gather wave spends 62 % gathering and 37 % waiting at the item barrier; a transform / store wave 13 % taking its inputs,
// 76 % in stores + second pass + untangle + global stores, 10 % waiting -- the transform / store waves are the longer role,
// and most of their time is waiting for vector-memory issue behind the gather's loads: both roles go through the same
// address unit, which the split cannot duplicate.  NOT the default; kept as a TEST VARIANT (-DDLPD_TEST_VARIANTS:
// tests/variants/libdlpd_variants.so and the emulated library) because it is the bit-exactness cross-check of K1.
#include <dlpd_platform.h>
#include "dlpd_fft.h"
#include "dlpd_internal.h"
#include "dlpd_k1.h"

extern "C" int dlpd_grid_supported(int L);
#ifndef DLPD_TEST_VARIANTS
int dlpd_k1_role_split(const float4*, const float*, cplx*, int, int, float, hipStream_t, int, int, int, int) {
  return DLPD_ERR_UNSUPPORTED;               // the product library ships one K1 formulation (k_rotate_zfft_cl)
}
extern "C" int dlpd_k1_form_supported(int L, int form) {
  return (form == 0 || form == 1) ? dlpd_grid_supported(L) : 0;
}
#else
extern "C" int dlpd_k1_form_supported(int L, int form) {
  if (form == 0 || form == 1) return dlpd_grid_supported(L);
  return (form == 2 && (L == 64 || L == 80)) ? 1 : 0;
}

template <int N> struct K1RsCfg {
  static constexpr int L = N / 2, NZ = N / 2 + 1;
  static constexpr int NP = K1ClCfg<N>::NP, CC = K1ClCfg<N>::CC, YG = K1ClCfg<N>::YG, NPR = YG / 2, LPV = CC / 4;
  static constexpr int GW = 8, XW = NP / 8;             // gather waves; transform waves (8 pencils each)
  static constexpr int GT = 64 * GW, NT = 64 * (GW + XW);
  static constexpr int INBUF = (N == 128) ? 2 : 1;
  // row strides (complex elements), both = 8 (mod 32): four pencils of a half-wave sit 16 banks apart
  static constexpr int RSI = (L % 32 == 0) ? L + 8 : ((L + 31) / 32) * 32 + 8, RSW = N + 8;
  static constexpr size_t LDS_BYTES = (size_t)(INBUF * NP * RSI + NP * RSW + N) * sizeof(cplx);
  static_assert(NP % 8 == 0 && RSI % 32 == 8 && RSW % 32 == 8 && RSI >= L, "pencil sets of 8, conflict-free strides");
};

// compact input pencils: element e at base + e (natural order)
struct PlainAddr {
  static constexpr bool IS_ROW = true;
  int base;
  DLPD_HD int operator()(int e) const { return base + e; }
};

// First pass of the z transform in two halves: the L non-zero inputs of this thread's butterflies from the input buffer into
// registers (only those: RNZ of R per butterfly), and -- once the buffer has been handed back -- zero fill, butterfly and the
// stores into the work buffer, one butterfly at a time.  Same loads, same SmallDft, same stores as FftPassW::load / store.
template <class Pass, class Addr> DLPD_D void k1r_first_pass_inputs(Pass& ps, const cplx* in, const Addr& iad, int t) {
#pragma unroll
  for (int i = 0; i < Pass::PER; i++)
    if (ps.active(i, t)) {
#pragma unroll
      for (int r = 0; r < Pass::RNZ; r++) ps.v[i][r] = lds_ld(in + iad(Pass::bf(i, t) + r * Pass::NBF));
    }
}
template <class Pass, int R, class Addr> DLPD_D void k1r_first_pass_run_store(Pass& ps, cplx* work, const Addr& wad, int t,
                                                                               const cplx* tw) {
#pragma unroll
  for (int i = 0; i < Pass::PER; i++)
    if (ps.active(i, t)) {
#pragma unroll
      for (int r = Pass::RNZ; r < R; r++) ps.v[i][r] = c_make(0.f, 0.f);
      ps.twiddle_and_run(i, Pass::bf(i, t), tw);
#pragma unroll
      for (int r = 0; r < R; r++) lds_st(work + wad(ps.out_index(i, r, t)), ps.v[i][r]);
    }
}

#ifdef DLPD_STAMPS
__device__ unsigned long long dlpd_stamps_k1r[32];
extern "C" int dlpd_debug_read_stamps_k1r(unsigned long long* host32) {
  if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(dlpd_stamps_k1r), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[32] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k1r), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

// work item `seq` of XCD `xcd` (the order of k_rotate_zfft_cl's grid: every XCD a contiguous range of (b, x), the
// (row group, channel chunk) of a plane innermost, so that neighbouring planes -- which share source lines -- meet in one L2)
struct K1Item { int b, x, yg, chunk; bool live; };
DLPD_D K1Item k1r_item(int xcd, int seq, int per, int gper, int groups, int nchunk, int L) {
  K1Item it;
  const int g = xcd * gper + seq / per, inner = seq % per;
  it.live = (seq / per < gper) && (g < groups);
  it.b = g / L;
  it.x = g % L;
  it.yg = inner / nchunk;
  it.chunk = inner % nchunk;
  return it;
}

template <int N> __global__ void __launch_bounds__(K1RsCfg<N>::NT)
k_rotate_zfft_cl_rs(const float4* __restrict__ cl, const float* __restrict__ R, cplx* __restrict__ A,
                    int C, int Cq, int nb, float c0, int CT_out, int c_base, int ext, int items_per_block) {
  typedef K1RsCfg<N> G;
  constexpr int L = G::L, NZ = G::NZ, NP = G::NP, CC = G::CC, YG = G::YG, NPR = G::NPR, LPV = G::LPV;
  constexpr int RSI = G::RSI, RSW = G::RSW, INBUF = G::INBUF, GT = G::GT;
  constexpr int R1 = FftPlan<N>::R1, R2 = FftPlan<N>::R2;      // the radix plan of k_rotate_zfft_cl, run wave-locally
  DLPD_DYN_SHARED(cplx, S);
  cplx* in0 = S;
  cplx* work = S + INBUF * NP * RSI;
  cplx* tw = work + NP * RSW;
  const int tid = threadIdx.x, wave = DLPD_UNIFORM(tid >> 6), lane = tid & 63;      // (the role is a SCALAR condition)
  const bool gatherer = wave < G::GW;
  const int nchunk = Cq / (CC / 4), per = (L / YG) * nchunk;
  const int groups = nb * L, gper = (groups + 7) / 8;
  const int xcd = blockIdx.x & 7, first = (blockIdx.x >> 3) * items_per_block;
  const int total = gper * per;
  const int n_items = min(items_per_block, total - first);
  if (n_items <= 0) return;
  for (int k = tid; k < N; k += G::NT) {                // the table of k_rotate_zfft_cl (init_twiddles, dlpd_corr.hip)
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    tw[k] = c_make((float)c, (float)s);
  }
  DLPD_STAMP_DECL;

  // ---- gather role: item `it` into input buffer `buf`
  auto gather = [&](int it, cplx* buf) {
    const K1Item w = k1r_item(xcd, first + it, per, gper, groups, nchunk, L);
    if (!w.live) return;
    const K1ClRot rot = k1cl_load_rotation(R + (size_t)w.b * 9);
    int task0 = tid;
    DLPD_OPAQUE(task0);                                  // nothing derived from the lane is hoisted out of the item loop
    for (int task = task0; task < NPR * L * LPV; task += GT) {
      const int q = task % LPV, z = (task / LPV) % L, m = (task / LPV) / L;
      const float4* src = cl + w.chunk * (CC / 4) + q;
      float4 acc[2];
      k1cl_sample_rows(src, Cq, L, ext, c0, rot, w.x, w.yg * YG + 2 * m, z, acc);
      // rows 2m (real part) and 2m+1 (imaginary part) of the four channels' pencils
      cplx* P = buf + ((4 * q) * NPR + m) * RSI + z;
      P[0] = c_make(acc[0].x, acc[1].x);
      P[NPR * RSI] = c_make(acc[0].y, acc[1].y);
      P[2 * NPR * RSI] = c_make(acc[0].z, acc[1].z);
      P[3 * NPR * RSI] = c_make(acc[0].w, acc[1].w);
    }
  };

  // ---- transform / store role (wave xw of XW owns pencils 8 xw .. 8 xw + 7; lane = 8 * pencil + thread)
  const int xw = wave - G::GW;
  typedef FftPassW<N, R1, 1, -1, 8, L> Pass1;
  auto second_pass_and_store = [&](int it, int p, int t) {
    const RowAddr<0> wad = {p * RSW};
    {
      FftPassW<N, R2, R1, -1, 8> ps;
      ps.load(work, wad, t, tw);
      DLPD_WAVE_SYNC();
      ps.store(work, wad, t);
      DLPD_WAVE_SYNC();
    }
    const K1Item w = k1r_item(xcd, first + it, per, gper, groups, nchunk, L);
    // untangle the two real rows packed in each complex pencil of this wave; write [kz][x][y]
    const int pl = t, kk = p & 7;                       // pencil of the set, bin offset (lane % 8, lane / 8)
    const int pm = 8 * xw + pl;
    const int c = w.chunk * CC + pm / NPR, m = pm % NPR;
    const cplx* Z = work + pm * RSW;
    if (c < C) {
      cplx* a0 = A + (((size_t)w.b * CT_out + c_base + c) * NZ) * L * L + (size_t)w.x * L + w.yg * YG + 2 * m;
      for (int k = kk; k < NZ; k += 8) {
        const cplx zk = lds_ld(Z + slab_swz(k));
        const cplx zn = lds_ld(Z + slab_swz((N - k) % N));
        float4 o;
        o.x = 0.5f * (zk.x + zn.x);
        o.y = 0.5f * (zk.y - zn.y);
        o.z = 0.5f * (zk.y + zn.y);
        o.w = 0.5f * (zn.x - zk.x);
        DLPD_STORE_STREAM(reinterpret_cast<float4*>(a0 + (size_t)k * L * L), o);
      }
    }
  };

  __syncthreads();                                       // twiddles
  if (gatherer) gather(0, in0);
  __syncthreads();                                       // item 0 gathered
  DLPD_STAMP(0);
  for (int it = 0; it < n_items; it++) {
    cplx* cur = in0 + (INBUF == 2 ? (it & 1) * NP * RSI : 0);
    if (INBUF == 2) {
      if (gatherer) {
        if (it + 1 < n_items) gather(it + 1, in0 + ((it + 1) & 1) * NP * RSI);
        DLPD_STAMP(1);
      } else {
        const K1Item w = k1r_item(xcd, first + it, per, gper, groups, nchunk, L);
        if (w.live) {
          int ln = lane;
          DLPD_OPAQUE(ln);                               // (the swizzled LDS offsets are recomputed per item, not kept in registers)
          const int p = 8 * xw + (ln >> 3), t = ln & 7;
          {
            Pass1 ps;
            const PlainAddr iad = {p * RSI};
            const RowAddr<0> wad = {p * RSW};
            k1r_first_pass_inputs(ps, cur, iad, t);
            DLPD_WAVE_SYNC();
            k1r_first_pass_run_store<Pass1, R1>(ps, work, wad, t, tw);
            DLPD_WAVE_SYNC();
          }
          DLPD_STAMP(2);
          second_pass_and_store(it, p, t);
        }
        DLPD_STAMP(3);
      }
      __syncthreads();
      DLPD_STAMP(4);
    } else {
      // one input buffer: the first pass takes its inputs into registers, THEN the buffer goes back to the gather waves.
      // Each role meets the "input buffer free" barrier inside its OWN branch (whole waves take one branch, and the
      // barrier counts waves): the first pass's registers are then live in the transform branch only.
      if (gatherer) {
        DLPD_LDS_BARRIER();                              // input buffer free
        DLPD_STAMP(5);
        if (it + 1 < n_items) gather(it + 1, in0);
        DLPD_STAMP(1);
      } else {
        int ln = lane;
        DLPD_OPAQUE(ln);
        const int p = 8 * xw + (ln >> 3), t = ln & 7;
        const bool live = k1r_item(xcd, first + it, per, gper, groups, nchunk, L).live;
        Pass1 ps;
        if (live) {
          const PlainAddr iad = {p * RSI};
          k1r_first_pass_inputs(ps, cur, iad, t);
        }
        DLPD_STAMP(2);
        DLPD_LDS_BARRIER();                              // input buffer free
        DLPD_STAMP(5);
        if (live) {
          const RowAddr<0> wad = {p * RSW};
          k1r_first_pass_run_store<Pass1, R1>(ps, work, wad, t, tw);
          DLPD_WAVE_SYNC();
          second_pass_and_store(it, p, t);
        }
        DLPD_STAMP(3);
      }
      __syncthreads();                                   // item it + 1 gathered
      DLPD_STAMP(4);
    }
  }
#ifdef DLPD_STAMPS
  DLPD_STAMP_FLUSH(dlpd_stamps_k1r, 0);                  // a gather wave: slots 0 (prologue) 1 (gather) 4/5 (barrier waits)
  if (lane == 0 && wave == G::GW) {                      // a transform wave: slots 2 (first pass) 3 (rest + stores) 4/5 (waits)
    for (int i_ = 0; i_ < 8; i_++) atomicAdd(&dlpd_stamps_k1r[16 + i_], st_sum[i_]);
    atomicAdd(&dlpd_stamps_k1r[31], 1ull);
  }
#endif
}

template <int N> static int launch_k1_rs(const float4* cl, const float* R, cplx* A, int C, int nb, float c0, hipStream_t st,
                                         int CT_out, int c_base, int ext) {
  typedef K1RsCfg<N> G;
  constexpr int L = N / 2;
  const int Cq = ((C + DLPD_K1CL_CC - 1) / DLPD_K1CL_CC) * (DLPD_K1CL_CC / 4);
  int rc = dlpd_set_max_dyn_shared((const void*)k_rotate_zfft_cl_rs<N>, G::LDS_BYTES);
  if (rc) return rc;
  const int per = (L / G::YG) * (Cq / (G::CC / 4));
  const int gper = (nb * L + 7) / 8;
  const int total = gper * per;                          // items per XCD
  // one block per CU: 32 blocks per XCD share the XCD's items in contiguous ranges
  const int bpx = 32, ipb = (total + bpx - 1) / bpx;
  dim3 grid((unsigned)(8 * ((total + ipb - 1) / ipb))), block(G::NT);
  DLPD_LAUNCH((k_rotate_zfft_cl_rs<N>), grid, block, G::LDS_BYTES, st, cl, R, A, C, Cq, nb, c0, CT_out, c_base,
              (ext > 0 && ext < L) ? ext : L, ipb);
  return dlpd_check_launch();
}

int dlpd_k1_role_split(const float4* cl, const float* R, cplx* A, int C, int nb, float c0, hipStream_t st, int CT_out, int c_base,
                       int ext, int L) {
  switch (L) {
    case 64: return launch_k1_rs<128>(cl, R, A, C, nb, c0, st, CT_out, c_base, ext);
    case 80: return launch_k1_rs<160>(cl, R, A, C, nb, c0, st, CT_out, c_base, ext);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}
#endif  // DLPD_TEST_VARIANTS
