// Representation-plugin convolution: Conv3d(cin, cout, k, padding = k/2, stride 1,
// bias = False) [+ ReLU] as the reference's E3MultiResRepr4x4 stacks them
//   /root/reference/src/Models/ProteinRepresentationModels.py:85-114
// -- the per-batch cost of Docker.dockE3 (Docker.py:166-167) -- as an implicit GEMM on the f32-input
// matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered fmaf chain):
//     Y[co][voxel] = sum over (tap, ci)  W[co][ci][tap] * X[ci][voxel + tap - k/2]
//   M = 16 output channels, N = 16 z-consecutive voxels, K = 4 input channels of one tap per MFMA.
// One block = a 4 x 4 (x, y) patch of full-z rows for one volume; input channels are staged through
// LDS four at a time (halo tile + that chunk's weights), each wave owns two rows x NZT z-tiles x
// COUT/16 accumulator tiles.  A fragment (weights) is read once per tap and reused by all 2*NZT
// voxel tiles of the wave; the B fragment is one conflict-free ds_read_b32 per MFMA.
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

template <int KS, int COUT, int NZT, int RELU> __global__ void __launch_bounds__(512)
k_conv3d_mfma(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int CIN, int D) {
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
  for (int ch = 0; ch < nchunk; ch++) {
    __syncthreads();                                           // previous chunk fully consumed
    // ---- stage the halo tile of channels 4*ch .. 4*ch+3 (zeros outside the volume / beyond CIN)
    for (int i = tid; i < 4 * C::XS * C::YS * C::ZS; i += 512) {
      const int zz = i % C::ZS, yy = (i / C::ZS) % C::YS, xx = (i / (C::ZS * C::YS)) % C::XS, k = i / (C::ZS * C::YS * C::XS);
      const int gx = x0 + xx - H, gy = y0 + yy - H, gz = zz - H, ci = 4 * ch + k;
      float v = 0.f;
      if (ci < CIN && gx >= 0 && gx < D && gy >= 0 && gy < D && gz >= 0 && gz < D)
        v = Xb[(size_t)ci * D3 + ((size_t)gx * D + gy) * D + gz];
      Xs[k * C::PLANE + (xx * C::YS + yy) * C::ZS + zz] = v;
    }
    // ---- this chunk's weights: Ws[tap][k][co] = W[co][4*ch + k][tap]
    for (int i = tid; i < C::WCHUNK; i += 512) {
      const int co = i % COUT, k = (i / COUT) % 4, tap = i / (4 * COUT), ci = 4 * ch + k;
      Ws[i] = ci < CIN ? W[((size_t)co * CIN + ci) * C::NTAP + tap] : 0.f;
    }
    __syncthreads();
    // ---- all taps of the chunk
#pragma unroll 1
    for (int dx = 0; dx < KS; dx++)
#pragma unroll 1
      for (int dy = 0; dy < KS; dy++) {
        const float* xr[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int row = 2 * wave + r, rx = row / C::TY, ry = row % C::TY;
          xr[r] = Xs + kq * C::PLANE + ((rx + dx) * C::YS + (ry + dy)) * C::ZS + n;
        }
        const float* wr = Ws + ((dx * KS + dy) * KS) * 4 * COUT + kq * COUT + n;
#pragma unroll
        for (int dz = 0; dz < KS; dz++) {
          float a[MT];
#pragma unroll
          for (int mt = 0; mt < MT; mt++) a[mt] = wr[dz * 4 * COUT + mt * 16];
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int zt = 0; zt < NZT; zt++) {
              const float bv = xr[r][zt * 16 + dz];
#pragma unroll
              for (int mt = 0; mt < MT; mt++) acc[r][zt][mt] = DLPD_MFMA_16x16x4(a[mt], bv, acc[r][zt][mt]);
            }
        }
      }
  }
  // ---- epilogue: lane holds rows (output channels) 4*kq + j, column (voxel) n of each tile
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int row = 2 * wave + r, gx = x0 + row / C::TY, gy = y0 + row % C::TY;
    if (gx >= D || gy >= D) continue;
#pragma unroll
    for (int zt = 0; zt < NZT; zt++) {
      const int gz = zt * 16 + n;
      if (gz >= D) continue;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float v = dlpd_acc4_get(acc[r][zt][mt], j);
          if (RELU) v = fmaxf(v, 0.f);
          Y[((size_t)b * COUT + mt * 16 + 4 * kq + j) * D3 + ((size_t)gx * D + gy) * D + gz] = v;
        }
    }
  }
}

template <int KS, int COUT, int NZT> static int launch_conv(const float* X, const float* W, float* Y, int B, int CIN,
                                                            int D, int relu, hipStream_t st) {
  typedef ConvCfg<KS, COUT, NZT> C;
  dim3 grid((D + C::TX - 1) / C::TX, (D + C::TY - 1) / C::TY, B), block(512);
  int rc;
  if (relu) {
    rc = dlpd_set_max_dyn_shared((const void*)k_conv3d_mfma<KS, COUT, NZT, 1>, C::LDS_BYTES);
    if (rc) return rc;
    DLPD_LAUNCH((k_conv3d_mfma<KS, COUT, NZT, 1>), grid, block, C::LDS_BYTES, st, X, W, Y, CIN, D);
  } else {
    rc = dlpd_set_max_dyn_shared((const void*)k_conv3d_mfma<KS, COUT, NZT, 0>, C::LDS_BYTES);
    if (rc) return rc;
    DLPD_LAUNCH((k_conv3d_mfma<KS, COUT, NZT, 0>), grid, block, C::LDS_BYTES, st, X, W, Y, CIN, D);
  }
  return dlpd_check_launch();
}

extern "C" {

int dlpd_conv3d_supported(int cin, int cout, int ks, int D) {
  if (cin <= 0 || D <= 0 || D > 80) return 0;
  const bool k = (ks == 3 || ks == 5);
  return (k && (cout == 16 || cout == 32)) ? 1 : 0;
}

int dlpd_conv3d(const float* x, const float* w, float* y, int B, int cin, int cout, int D, int ks, int relu,
                void* stream) {
  if (!x || !w || !y || B <= 0) return DLPD_ERR_ARG;
  if (!dlpd_conv3d_supported(cin, cout, ks, D)) return DLPD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nzt = (D + 15) / 16;                               // z tiles per row
#define DLPD_CONV(KS, CO, NZ) if (ks == KS && cout == CO && nzt <= NZ) return launch_conv<KS, CO, NZ>(x, w, y, B, cin, D, relu, st)
  DLPD_CONV(3, 16, 3); DLPD_CONV(3, 16, 5); DLPD_CONV(5, 16, 3); DLPD_CONV(5, 16, 5);
  DLPD_CONV(3, 32, 3); DLPD_CONV(3, 32, 5); DLPD_CONV(5, 32, 3); DLPD_CONV(5, 32, 5);
#undef DLPD_CONV
  return DLPD_ERR_UNSUPPORTED;
}

}  // extern "C"
