// K3 with the filter MLP on the matrix cores (N = 128, hidden width 17..32): z-axis C2R + clip + SimpleFilter MLP
// + clash mask, same inputs, outputs and arithmetic as k_zifft_filter<N, HP, 1> (dlpd_corr.hip).
//
// Reference path being replaced (file:line in /root/reference):
//   src/Models/DockingModels.py:70-83   per-channel correlation (its z-inverse), concat, SimpleFilter MLP
//   src/Docker/Docker.py:226,232        clash threshold, mask multiply
//
// Why a second formulation.  In k_zifft_filter a wave owns a CHANNEL of the current group (all 8 row pairs of the tile),
// so the MLP -- which needs all channels of a voxel -- sits behind a block barrier: pack + FFT (LDS + VALU) and MLP
// (VALU: 4.0e10 f32 FMAs per launch, ~80 % of the measured vector issue rate) alternate in lock step, 27 % of the kernel
// is barrier wait, and its DMA-only skeleton runs 1.36 ms against 2.5 ms for the whole.  Here a wave owns a ROW PAIR
// of the tile for all channels:
//   * the DMA is unchanged (wave w streams channel 8g + w of group g, 128-byte runs), but after the hand-off barrier
//     every wave packs ITS pair of all 8 channels into 8 private pencils, runs the wave-local z C2R passes on them
//     and folds them into the hidden units of its own 256 voxels -- no barrier between transform and MLP, none
//     around the MLP; the two block barriers per group only fence the short raw hand-off;
//   * the MLP runs on the matrix pipe (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered fmaf chain, the same
//     sums as the vector form), so a wave's multiplications proceed beside its SIMD partner's butterflies and LDS
//     traffic instead of competing for the vector issue slots.
// Fragment maps (dlpd_platform.h): A[m = hidden][k = channel] in lane (k << 4 | m), B[k][n = voxel] in lane
// (k << 4 | n), D[4 * (lane >> 4) + j][lane & 15]; a wave's 256 voxels = 2 rows x 8 z tiles of 16, times two 16-wide
// hidden tiles: 32 accumulators of 4 registers.
// (Measured and rejected on the way: two teams of 4 waves alternating transform / MFMA roles every phase over three
// 4-channel pencil buffers -- 3.0 ms: only half the waves transform at a time and a single wave cannot keep the LDS
// busy; an L2 warm-up of the next-but-one DMA changed nothing, the DMA was not what the phases waited for.)
#include <dlpd_platform.h>
#include "dlpd_fft.h"
#include "dlpd_internal.h"

template <int N> DLPD_D void init_twiddles(cplx* tw, int tid, int nthreads) {
  for (int k = tid; k < N; k += nthreads) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    tw[k] = c_make((float)c, (float)s);
  }
}

#ifdef DLPD_STAMPS   // diagnostic build only (scripts/stamps.py): where a wave of each team spends its cycles
__device__ unsigned long long dlpd_stamps_k3m[16];
extern "C" int dlpd_debug_read_stamps_k3m(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(dlpd_stamps_k3m), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k3m), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

template <int N> struct K3mCfg {
  static constexpr int NT = 512, NW = 8, G = 8, TY = 16, NPAIR = 8;
  static constexpr int NZ = N / 2 + 1, RS = N + 8;
  static constexpr int RAWC = ((NZ * NPAIR + 63) / 64) * 64 + 1;   // float4 slots per raw channel, + 1: consecutive
                                                       // channels start 16 bytes apart (mod 256), so the 8 channels x 8 k
                                                       // that a pack instruction reads cover 1 KB of distinct banks
  static constexpr size_t LDS_BYTES = (size_t)(NW * 8 * RS + N) * sizeof(cplx) + (size_t)G * RAWC * 16;
};

// MT: 16-wide hidden tiles (HP <= 16 * MT)
template <int N, int MT> __global__ void __launch_bounds__(512)
k_zifft_mlp_mfma(const cplx* __restrict__ Bw, float* __restrict__ out, int CT, int C, int has_clash,
                 const float* __restrict__ W1t, int HP, const float* __restrict__ b1, const float* __restrict__ W2,
                 float b2, int has_clip, float clip, float thr) {
  typedef K3mCfg<N> Cfg;
  constexpr int NZ = Cfg::NZ, RS = Cfg::RS, G = Cfg::G, NPAIR = Cfg::NPAIR, RAWC = Cfg::RAWC;
  constexpr int LPK = 64 / NPAIR, NFULL = (N / 2) / LPK, NZT = N / 16;
  static_assert(NZ * NPAIR == NFULL * 64 + NPAIR, "raw channel = NFULL full DMA instructions + one short one");
  DLPD_DYN_SHARED(cplx, S);                              // [wave][8 pencils][RS]: the wave's row pair of 8 channels
  cplx* tw = S + Cfg::NW * 8 * RS;
  float4* raw = reinterpret_cast<float4*>(tw + N);       // [G][RAWC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = blockIdx.x * Cfg::TY, xo = blockIdx.y, b = blockIdx.z;
  const int kq = lane >> 4, n = lane & 15;               // MFMA fragment coordinates
  const int ng = (CT + G - 1) / G;
  cplx* Pw = S + wave * 8 * RS;
  // pencil slot of channel j of the group: the channels of one MFMA K-step (4 consecutive j) sit 0, 128, 64, 192 bytes
  // (mod 256) apart, so the B-fragment read of channels k, k + 1 (one 32-lane half) uses disjoint banks
  auto slot_of = [](int j) { return (j & 4) | ((j & 1) << 1) | ((j >> 1) & 1); };

  // hidden pre-activations: D[hidden][voxel] tiles, bias in every column
  dlpd_acc4 acc[NZT][2][MT];
  {
    dlpd_acc4 bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float bj[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int h = 16 * mt + 4 * kq + j;
        bj[j] = h < HP ? b1[h] : 0.f;
      }
      bias[mt] = dlpd_acc4_make(bj[0], bj[1], bj[2], bj[3]);
    }
#pragma unroll
    for (int zt = 0; zt < NZT; zt++)
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[zt][u][mt] = bias[mt];
  }

  // raw[w][k][m] <- Bw[b][8g + w][k][xo][y0 + 2m .. +1]: wave w streams channel w of the group (lane = 8 (k % 8) + m:
  // 8 runs of 128 bytes per DMA instruction; k = N/2 is the last, short one)
  auto issue_dma = [&](int g) {
    const int c = G * g + wave;
    if (c < CT) {
      float4* rawg = raw + wave * RAWC;
      const cplx* src = Bw + (((size_t)b * CT + c) * NZ * N + xo) * N + y0;
      const cplx* lane_src = src + (size_t)(lane / NPAIR) * N * N + 2 * (lane % NPAIR);
#pragma unroll
      for (int it = 0; it < NFULL; it++) DLPD_GLDS16(lane_src + (size_t)it * LPK * N * N, rawg + it * 64);
      DLPD_GLDS16(src + (size_t)(N / 2) * N * N + 2 * (lane % NPAIR), rawg + NFULL * 64);
    }
  };
  DLPD_STAMP_DECL;
  // this wave's row pair of all (<= 8) channels of the group: raw -> 8 private pencils.  Z[k] = A[k] + i B[k],
  // Z[N-k] = conj(A[k]) + i conj(B[k]) (two real rows per complex pencil).  Lanes dealt over (channel j, k row) like
  // k_zifft_filter's pack deals them over (pair, k row): conflict-free raw reads, 5-cycle stores.
  auto pack = [&](int gn) {
    int lq = lane;
    DLPD_OPAQUE(lq);                 // lane-dependent offsets recomputed per group: the 128 accumulators leave no room to hoist them
    const int j = (lq & 3) | (((lq >> 4) & 1) << 2);
    const int kk = ((lq >> 2) & 1) | (((lq >> 5) & 1) << 1) | (((lq >> 3) & 1) << 2);
    const int rot = 2 * slot_of(j);                      // (with the slot permutation the rotation that keeps the stores at 5 cycles)
    const float4* rj = raw + j * RAWC + wave;            // slot 8 k + (pair = wave) of channel j
    cplx* P = Pw + slot_of(j) * RS;
    const bool live = j < gn;
    constexpr int PCH = 4;                               // raw elements in flight per lane
#pragma unroll
    for (int u0 = 0; u0 < NFULL; u0 += PCH) {
      float4 q[PCH];
#pragma unroll
      for (int u = 0; u < PCH; u++) q[u] = rj[(((u0 + u + rot) % NFULL) * LPK + kk) * NPAIR];
#pragma unroll
      for (int u = 0; u < PCH; u++) {
        const int k = ((u0 + u + rot) % NFULL) * LPK + kk;
        cplx lo = (k == 0) ? c_make(q[u].x, q[u].z) : c_make(q[u].x - q[u].w, q[u].y + q[u].z);
        cplx hi = (k == 0) ? c_make(q[u].x, q[u].z) : c_make(q[u].x + q[u].w, q[u].z - q[u].y);
        if (!live) { lo = c_make(0.f, 0.f); hi = lo; }   // channels beyond the group: zero pencils (never stale LDS)
        P[slab_swz(k)] = lo;
        P[slab_swz((N - k) & (k == 0 ? 0 : ~0))] = hi;
      }
    }
    if (kk == 0) {
      const float4 qh = rj[(N / 2) * NPAIR];
      P[slab_swz(N / 2)] = live ? c_make(qh.x, qh.z) : c_make(0.f, 0.f);
    }
  };
  // the wave's 256 voxels x the 8 channels of the group (two K-steps of 4)
  auto accumulate = [&](int g) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      float a[MT];
      const int c = G * g + 4 * s + kq;
      const bool live = c < C;                           // score channels only (the clash channel is not an MLP input)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const int h = 16 * mt + n;                       // A fragment: hidden unit = lane & 15
        a[mt] = (live && h < HP) ? W1t[(size_t)c * HP + h] : 0.f;
      }
      const cplx* Pb = Pw + slot_of(4 * s + kq) * RS;
#pragma unroll
      for (int zt = 0; zt < NZT; zt++) {
        cplx v = Pb[slab_swz(16 * zt + n)];
        if (!live) v = c_make(0.f, 0.f);
        if (has_clip) { v.x = DLPD_CLAMP(v.x, clip); v.y = DLPD_CLAMP(v.y, clip); }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          acc[zt][0][mt] = DLPD_MFMA_16x16x4(a[mt], v.x, acc[zt][0][mt]);
          acc[zt][1][mt] = DLPD_MFMA_16x16x4(a[mt], v.y, acc[zt][1][mt]);
        }
      }
    }
  };

  issue_dma(0);
  init_twiddles<N>(tw, tid, Cfg::NT);
  float nrm0[NZT / 4], nrm1[NZT / 4];                    // clash correlation of the voxels this lane will write
#pragma unroll
  for (int i = 0; i < NZT / 4; i++) nrm0[i] = nrm1[i] = 0.f;
#pragma unroll 1
  for (int g = 0; g < ng; g++) {
    const int gn = (CT - G * g) < G ? (CT - G * g) : G;
    DLPD_WAIT_VMEM();                                    // this wave's channel has landed
    __syncthreads();                                     // ... and everybody else's (first pass: twiddle table too)
    DLPD_STAMP(5);
    pack(gn);
    DLPD_WAIT_LDS();
    DLPD_STAMP(0);
    __syncthreads();                                     // raw drained by all waves
    DLPD_STAMP(5);
    if (g + 1 < ng) issue_dma(g + 1);
    DLPD_STAMP(1);
    {
      int lq = lane;
      DLPD_OPAQUE(lq);
      const RowAddr<RS> ad = {(int)(Pw - S) + (lq >> 3) * RS};
      fft_wave<N, +1, N>(S, ad, lq & 7, tw);
    }
    DLPD_WAVE_SYNC();
    DLPD_STAMP(2);
    accumulate(g);
    if (has_clash && C >= G * g && C < G * g + G) {      // the clash channel lives in this group: keep its values
      const cplx* Pn = Pw + slot_of(C - G * g) * RS;
#pragma unroll
      for (int i = 0; i < NZT / 4; i++) {
        const cplx nv = Pn[slab_swz(16 * (4 * i + kq) + n)];
        nrm0[i] = nv.x;
        nrm1[i] = nv.y;
      }
    }
    DLPD_WAIT_LDS();                                     // pencils read before the next pack overwrites them
    DLPD_STAMP(4);
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k3m, DLPD_STAMPS);

  // ---- second layer, clash mask, store.  Lane (kq, n) holds hidden units 4 kq + j (+ 16 mt) of voxel n of every tile.
  float w2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int h = 16 * mt + 4 * kq + j;
      w2[mt][j] = h < HP ? W2[h] : 0.f;
    }
  // lane group kq writes the z tiles zt == kq (mod 4): a store instruction covers 64 consecutive z of one row
#pragma unroll
  for (int i = 0; i < NZT / 4; i++) {
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      float part[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        part[u] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            part[u] = fmaf(w2[mt][j], fmaxf(dlpd_acc4_get(acc[4 * i + g][u][mt], j), 0.f), part[u]);
        part[u] += __shfl_xor(part[u], 16);
        part[u] += __shfl_xor(part[u], 32);
      }
      v0 = (kq == g) ? part[0] + b2 : v0;
      v1 = (kq == g) ? part[1] + b2 : v1;
    }
    const int z = 16 * (4 * i + kq) + n;
    if (has_clash) {
      v0 = v0 * ((nrm0[i] < thr) ? 1.0f : 0.0f);
      v1 = v1 * ((nrm1[i] < thr) ? 1.0f : 0.0f);
    }
    float* o = out + (((size_t)b * N + xo) * N + y0 + 2 * wave) * N + z;
    o[0] = v0;
    o[N] = v1;
  }
}

int dlpd_k3_mfma_supported(int L, int HP) { return (L == 64 && HP > 16 && HP <= 32) ? 1 : 0; }

int dlpd_k3_mfma(const cplx* Bw, float* V, int CT, int C, int has_clash, int nb, int L, const float* W1t, int HP,
                 const float* b1, const float* W2, float b2, int has_clip, float clip, float thr, hipStream_t st) {
  if (!dlpd_k3_mfma_supported(L, HP)) return DLPD_ERR_UNSUPPORTED;
  constexpr int N = 128;
  typedef K3mCfg<N> Cfg;
  int rc = dlpd_set_max_dyn_shared((const void*)k_zifft_mlp_mfma<N, 2>, Cfg::LDS_BYTES);
  if (rc) return rc;
  dim3 grid(N / Cfg::TY, N, nb), block(Cfg::NT);
  DLPD_LAUNCH((k_zifft_mlp_mfma<N, 2>), grid, block, Cfg::LDS_BYTES, st, Bw, V, CT, C, has_clash, W1t, HP, b1, W2, b2,
              has_clip, clip, thr);
  return dlpd_check_launch();
}
