// Representation-plugin convolution: Conv3d(cin, cout, k, padding = k/2, stride 1 or 2,
// bias = False) [+ ReLU] as the reference's plugins stack them
//   /root/reference/src/Models/ProteinRepresentationModels.py:85-114   E3MultiResRepr4x4 (stride 1)
//   /root/reference/src/Models/ProteinRepresentationModels.py:38-61    SE3MultiResReprScalar (one stride-2 layer, :51)
// -- the per-batch cost of Docker.dockE3 (Docker.py:166-167) -- as an implicit GEMM on the f32-input
// matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered fmaf chain):
//     Y[co][voxel] = sum over (tap, ci)  W[co][ci][tap] * X[ci][voxel + tap - k/2]
//   M = 16 output channels, N = 16 z-consecutive voxels, K = 4 input channels of one tap per MFMA.
// One block = a 4 x 4 (x, y) patch of full-z rows for one volume; input channels are staged through
// LDS four at a time (halo tile + that chunk's weights), each wave owns two rows x NZT z-tiles x
// COUT/16 accumulator tiles.  A fragment (weights) is read once per tap and reused by all 2*NZT
// voxel tiles of the wave; the B fragment is one conflict-free ds_read_b32 per MFMA.
// STRIDE = 2 (one layer per protein, never per rotation): the output voxel o is the stride-1 result at 2o, so the
// same tile is walked with the rows of odd x or y skipped (wave-uniform) and only the even z written -- 2x the
// minimal matrix work on that layer instead of a second staging scheme for strided fragments.
#include <dlpd_platform.h>
#include "dlpd_internal.h"

template <int KS, int COUT, int NZT> struct ConvCfg {
  static constexpr int TX = 4, TY = 4, NW = 8;                 // patch, waves (2 rows per wave)
  static constexpr int H = KS / 2;
  static constexpr int XS = TX + KS - 1, YS = TY + KS - 1;
  static constexpr int ZS = NZT * 16 + KS - 1;                 // row stride (floats)
  static constexpr int PLANE0 = XS * YS * ZS;
  static constexpr int PLANE = PLANE0 + ((16 - PLANE0 % 64) + 64) % 64;   // == 16 (mod 64): the 4 channel
                                                               // planes of a B fragment hit disjoint banks
  static constexpr int NTAP = KS * KS * KS;
  static constexpr int WCHUNK = NTAP * 4 * COUT;               // floats of weights per channel chunk
  static constexpr size_t LDS_BYTES = (size_t)(4 * PLANE + WCHUNK) * sizeof(float);
};

template <int KS, int COUT, int NZT, int RELU, int STRIDE> __global__ void __launch_bounds__(512)
k_conv3d_mfma(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int CIN, int D,
              int cout_total, int co_base) {
  typedef ConvCfg<KS, COUT, NZT> C;
  constexpr int MT = COUT / 16, H = C::H;
  DLPD_DYN_SHARED(float, S);
  float* Xs = S;                         // [4][XS][YS][ZS] (+ plane padding)
  float* Ws = S + 4 * C::PLANE;          // [tap][4][COUT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * C::TX, y0 = blockIdx.y * C::TY, b = blockIdx.z;
  const size_t D3 = (size_t)D * D * D;
  const float* Xb = X + (size_t)b * CIN * D3;
  const int kq = lane >> 4, n = lane & 15;                     // fragment coordinates of this lane
  dlpd_acc4 acc[2][NZT][MT];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int zt = 0; zt < NZT; zt++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[r][zt][mt] = dlpd_acc4_zero();
  const int nchunk = (CIN + 3) / 4;
  // Staging.  One z row per wave and step: the row decode is wave-uniform, the lanes only add their z.
  // KS = 3 (short compute per chunk): the next chunk's tile and weights are fetched into registers
  // before the MFMAs of the current one and committed to LDS after them; KS = 5 stages directly (its
  // 125-tap compute phase dwarfs the fetch, and it has no registers to spare).
  constexpr int NROW = 4 * C::XS * C::YS, RPW = (NROW + C::NW - 1) / C::NW, ZL = (C::ZS + 63) / 64;
  constexpr int NWV = (C::WCHUNK / 4 + 511) / 512;
  constexpr bool PREFETCH = (KS == 3);
  float treg[PREFETCH ? RPW : 1][PREFETCH ? ZL : 1];
  float4 wreg[PREFETCH ? NWV : 1];
  auto row_src = [&](int row, int ch, bool& ok) -> const float* {
    const int yy = row % C::YS, xx = (row / C::YS) % C::XS, k = row / (C::YS * C::XS);
    const int gx = x0 + xx - H, gy = y0 + yy - H, ci = 4 * ch + k;
    ok = row < NROW && ci < CIN && gx >= 0 && gx < D && gy >= 0 && gy < D;
    return Xb + (size_t)(ok ? ci : 0) * D3 + ((size_t)(ok ? gx : 0) * D + (ok ? gy : 0)) * D - H;
  };
  auto row_dst = [&](int row) -> float* {
    const int yy = row % C::YS, xx = (row / C::YS) % C::XS, k = row / (C::YS * C::XS);
    return Xs + k * C::PLANE + (xx * C::YS + yy) * C::ZS;
  };
  auto fetch = [&](int ch) {                                   // global -> registers
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      bool ok;
      const float* src = row_src(wave + j * C::NW, ch, ok);
#pragma unroll
      for (int q = 0; q < ZL; q++) {
        const int zz = q * 64 + lane;
        treg[j][q] = (ok && zz >= H && zz < D + H) ? src[zz] : 0.f;
      }
    }
    const float4* wsrc = reinterpret_cast<const float4*>(W + (size_t)ch * C::WCHUNK);
#pragma unroll
    for (int j = 0; j < NWV; j++)
      if (tid + j * 512 < C::WCHUNK / 4) wreg[j] = wsrc[tid + j * 512];
  };
  auto commit = [&]() {                                        // registers -> LDS
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int row = wave + j * C::NW;
      if (row < NROW) {
        float* dst = row_dst(row);
#pragma unroll
        for (int q = 0; q < ZL; q++)
          if (q * 64 + lane < C::ZS) dst[q * 64 + lane] = treg[j][q];
      }
    }
#pragma unroll
    for (int j = 0; j < NWV; j++)
      if (tid + j * 512 < C::WCHUNK / 4) reinterpret_cast<float4*>(Ws)[tid + j * 512] = wreg[j];
  };
  if (PREFETCH) fetch(0);
  for (int ch = 0; ch < nchunk; ch++) {
    __syncthreads();                                           // previous chunk fully consumed
    if (PREFETCH) {
      commit();
    } else {
      for (int row = wave; row < NROW; row += C::NW) {
        bool ok;
        const float* src = row_src(row, ch, ok);
        float* dst = row_dst(row);
#pragma unroll
        for (int q = 0; q < ZL; q++) {
          const int zz = q * 64 + lane;
          if (zz < C::ZS) dst[zz] = (ok && zz >= H && zz < D + H) ? src[zz] : 0.f;
        }
      }
      const float4* wsrc = reinterpret_cast<const float4*>(W + (size_t)ch * C::WCHUNK);
      for (int i = tid; i < C::WCHUNK / 4; i += 512) reinterpret_cast<float4*>(Ws)[i] = wsrc[i];
    }
    __syncthreads();
    if (PREFETCH && ch + 1 < nchunk) fetch(ch + 1);            // in flight during this chunk's MFMAs
    // ---- all taps of the chunk.  One step = one (dx, dy) row = KS taps.  The fragments of step t+1 are
    // requested before the MFMAs of step t run (two register sets), so the matrix pipe never waits for
    // an LDS round trip; the fences keep the compiler from sinking the reads back next to their use.
    struct Frag { float a[KS][MT], b[KS][2][NZT]; };
    auto load_frag = [&](Frag& f, int t2) {
      const int dx = t2 / KS, dy = t2 % KS;
      const float* wr = Ws + ((dx * KS + dy) * KS) * 4 * COUT + kq * COUT + n;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int row = 2 * wave + r, rx = row / C::TY, ry = row % C::TY;
        const float* xr = Xs + kq * C::PLANE + ((rx + dx) * C::YS + (ry + dy)) * C::ZS + n;
#pragma unroll
        for (int dz = 0; dz < KS; dz++)
#pragma unroll
          for (int zt = 0; zt < NZT; zt++) f.b[dz][r][zt] = xr[zt * 16 + dz];
      }
#pragma unroll
      for (int dz = 0; dz < KS; dz++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) f.a[dz][mt] = wr[dz * 4 * COUT + mt * 16];
    };
    auto run_frag = [&](const Frag& f) {
#pragma unroll
      for (int dz = 0; dz < KS; dz++)
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (STRIDE == 2 && ((((2 * wave + r) / C::TY) | ((2 * wave + r) % C::TY)) & 1)) continue;   // odd x or y: no output
#pragma unroll
          for (int zt = 0; zt < NZT; zt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
              acc[r][zt][mt] = DLPD_MFMA_16x16x4(f.a[dz][mt], f.b[dz][r][zt], acc[r][zt][mt]);
        }
    };
    Frag f0, f1;
    load_frag(f0, 0);
#pragma unroll 1
    for (int t2 = 0; t2 < KS * KS; t2 += 2) {
      if (t2 + 1 < KS * KS) load_frag(f1, t2 + 1);
      DLPD_SCHED_FENCE();
      run_frag(f0);
      DLPD_SCHED_FENCE();
      if (t2 + 1 < KS * KS) {
        if (t2 + 2 < KS * KS) load_frag(f0, t2 + 2);
        DLPD_SCHED_FENCE();
        run_frag(f1);
        DLPD_SCHED_FENCE();
      }
    }
  }
  // ---- epilogue: lane holds rows (output channels) 4*kq + j, column (voxel) n of each tile
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int row = 2 * wave + r, gx = x0 + row / C::TY, gy = y0 + row % C::TY;
    if (gx >= D || gy >= D) continue;
    if (STRIDE == 2 && ((gx | gy) & 1)) continue;
    const int Do = (STRIDE == 2) ? (D - 1) / 2 + 1 : D;
    const size_t Do3 = (size_t)Do * Do * Do;
#pragma unroll
    for (int zt = 0; zt < NZT; zt++) {
      const int gz = zt * 16 + n;
      if (gz >= D || (STRIDE == 2 && (gz & 1))) continue;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float v = dlpd_acc4_get(acc[r][zt][mt], j);
          if (RELU) v = fmaxf(v, 0.f);
          Y[((size_t)b * cout_total + co_base + mt * 16 + 4 * kq + j) * Do3 +
            ((size_t)(gx / STRIDE) * Do + gy / STRIDE) * Do + gz / STRIDE] = v;
        }
    }
  }
}

template <int KS, int COUT, int NZT, int RELU, int STRIDE> static int launch_conv_rs(const float* X, const float* W, float* Y,
                                                                                  int B, int CIN, int D, int cout_total,
                                                                                  int co_base, hipStream_t st) {
  typedef ConvCfg<KS, COUT, NZT> C;
  dim3 grid((D + C::TX - 1) / C::TX, (D + C::TY - 1) / C::TY, B), block(512);
  int rc = dlpd_set_max_dyn_shared((const void*)k_conv3d_mfma<KS, COUT, NZT, RELU, STRIDE>, C::LDS_BYTES);
  if (rc) return rc;
  DLPD_LAUNCH((k_conv3d_mfma<KS, COUT, NZT, RELU, STRIDE>), grid, block, C::LDS_BYTES, st, X, W, Y, CIN, D, cout_total,
              co_base);
  return dlpd_check_launch();
}

template <int KS, int COUT, int NZT> static int launch_conv(const float* X, const float* W, float* Y, int B, int CIN,
                                                            int D, int relu, int stride, int cout_total, int co_base,
                                                            hipStream_t st) {
  if (stride == 2)      // (no fused ReLU on the strided form: the plugins have none behind their stride-2 layer's input side)
    return relu ? launch_conv_rs<KS, COUT, NZT, 1, 2>(X, W, Y, B, CIN, D, cout_total, co_base, st)
                : launch_conv_rs<KS, COUT, NZT, 0, 2>(X, W, Y, B, CIN, D, cout_total, co_base, st);
  return relu ? launch_conv_rs<KS, COUT, NZT, 1, 1>(X, W, Y, B, CIN, D, cout_total, co_base, st)
              : launch_conv_rs<KS, COUT, NZT, 0, 1>(X, W, Y, B, CIN, D, cout_total, co_base, st);
}

// Output channels are processed in groups of 32 (a last group of 16 if cout % 32 == 16), one launch each.
// wp[group][chunk][tap][k][co in group] = w[co][4*chunk + k][tap] (0 beyond cin): the order the kernel
// stages and reads.
__host__ __device__ inline int conv_group_width(int cout, int g) { return (cout - 32 * g) >= 32 ? 32 : 16; }
__global__ void __launch_bounds__(256) k_conv3d_pack(const float* __restrict__ w, float* __restrict__ wp, int cin,
                                                     int cout, int ntap) {
  const int nch = (cin + 3) / 4;
  const int total = nch * ntap * 4 * cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int per32 = nch * ntap * 4 * 32;                   // floats of a full group
    const int g = i / per32, gw = conv_group_width(cout, g), r = i - g * per32;
    const int col = r % gw, k = (r / gw) % 4, tap = (r / (4 * gw)) % ntap, ch = r / (4 * gw * ntap);
    const int ci = 4 * ch + k, co = 32 * g + col;
    wp[i] = ci < cin ? w[((size_t)co * cin + ci) * ntap + tap] : 0.f;
  }
}

// MaxPool3d(kernel 5, stride 2, padding 2) of the E3 plugin (ProteinRepresentationModels.py:101): out-of-range
// taps do not take part (torch pads with -inf).  One thread per output voxel, z fastest; the 125 taps of
// neighbouring outputs overlap, so the reads are served by L1/L2.
__global__ void __launch_bounds__(256) k_maxpool3d_5s2(const float* __restrict__ x, float* __restrict__ y, int nvol,
                                                       int D, int Do) {
  const size_t total = (size_t)nvol * Do * Do * Do;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int oz = (int)(i % Do), oy = (int)((i / Do) % Do), ox = (int)((i / ((size_t)Do * Do)) % Do);
    const size_t v = i / ((size_t)Do * Do * Do);
    const float* src = x + v * (size_t)D * D * D;
    float m = -INFINITY;
    for (int dx = -2; dx <= 2; dx++) {
      const int ix = 2 * ox + dx;
      if (ix < 0 || ix >= D) continue;
      for (int dy = -2; dy <= 2; dy++) {
        const int iy = 2 * oy + dy;
        if (iy < 0 || iy >= D) continue;
        const float* row = src + ((size_t)ix * D + iy) * D;
#pragma unroll
        for (int dz = -2; dz <= 2; dz++) {
          const int iz = 2 * oz + dz;
          if (iz >= 0 && iz < D) m = fmaxf(m, row[iz]);
        }
      }
    }
    y[i] = m;
  }
}

extern "C" {

int dlpd_maxpool3d_5s2(const float* x, float* y, int nvol, int D, void* stream) {
  if (!x || !y || nvol <= 0 || D < 1) return DLPD_ERR_ARG;
  const int Do = (D + 4 - 5) / 2 + 1;
  const size_t total = (size_t)nvol * Do * Do * Do;
  size_t nblk = (total + 255) / 256;
  if (nblk > 131072) nblk = 131072;
  DLPD_LAUNCH(k_maxpool3d_5s2, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, y, nvol, D, Do);
  return dlpd_check_launch();
}

size_t dlpd_conv3d_packed_floats(int cin, int cout, int ks) {
  return (size_t)((cin + 3) / 4) * ks * ks * ks * 4 * cout;
}

int dlpd_conv3d_pack(const float* w, float* wp, int cin, int cout, int ks, void* stream) {
  if (!w || !wp || cin <= 0 || cout <= 0 || ks <= 0) return DLPD_ERR_ARG;
  const int total = (int)dlpd_conv3d_packed_floats(cin, cout, ks);
  DLPD_LAUNCH(k_conv3d_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wp, cin, cout,
              ks * ks * ks);
  return dlpd_check_launch();
}

int dlpd_conv3d_supported(int cin, int cout, int ks, int D) {
  if (cin <= 0 || D <= 0 || D > 80) return 0;
  return ((ks == 3 || ks == 5) && cout >= 16 && cout % 16 == 0) ? 1 : 0;
}

int dlpd_conv3d_strided(const float* x, const float* wp, float* y, int B, int cin, int cout, int D, int ks, int relu,
                        int stride, void* stream);

int dlpd_conv3d(const float* x, const float* wp, float* y, int B, int cin, int cout, int D, int ks, int relu,
                void* stream) {
  return dlpd_conv3d_strided(x, wp, y, B, cin, cout, D, ks, relu, 1, stream);
}

int dlpd_conv3d_strided(const float* x, const float* wp, float* y, int B, int cin, int cout, int D, int ks, int relu,
                        int stride, void* stream) {
  if (!x || !wp || !y || B <= 0 || (stride != 1 && stride != 2)) return DLPD_ERR_ARG;
  if (!dlpd_conv3d_supported(cin, cout, ks, D)) return DLPD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nzt = (D + 15) / 16;                               // z tiles per row
  const size_t per32 = (size_t)((cin + 3) / 4) * ks * ks * ks * 4 * 32;
  for (int g = 0; 32 * g < cout; g++) {
    const int gw = conv_group_width(cout, g), base = 32 * g;
    const float* wg = wp + per32 * g;
    int rc = DLPD_ERR_UNSUPPORTED;
#define DLPD_CONV(KS, CO, NZ) if (rc == DLPD_ERR_UNSUPPORTED && ks == KS && gw == CO && nzt <= NZ) \
    rc = launch_conv<KS, CO, NZ>(x, wg, y, B, cin, D, relu, stride, cout, base, st)
    DLPD_CONV(3, 16, 3); DLPD_CONV(3, 16, 5); DLPD_CONV(5, 16, 3); DLPD_CONV(5, 16, 5);
    DLPD_CONV(3, 32, 3); DLPD_CONV(3, 32, 5); DLPD_CONV(5, 32, 3); DLPD_CONV(5, 32, 5);
#undef DLPD_CONV
    if (rc) return rc;
  }
  return DLPD_OK;
}

}  // extern "C"
