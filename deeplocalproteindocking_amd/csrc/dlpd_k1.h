// Shared pieces of the channels-last K1 (trilinear rotation + forward z transform, Docker.py:218 + the z pass of the
// correlation of DockingModels.py:70-71): the block shapes and THE sample -- both formulations of the kernel
// (k_rotate_zfft_cl in dlpd_corr.hip: every wave gathers, transforms and stores in turn; k_rotate_zfft_cl_rs in
// dlpd_k1r.hip: gather waves and transform / store waves) call the same function, so their samples are the same bits.
#pragma once
#include <dlpd_platform.h>
#include "dlpd_fft.h"

#define DLPD_K1CL_CC 16                   // channel padding of the channels-last copy
// rows x channels per block (64 two-row pencils either way).  8 x 16: 64-byte gathers, 64-byte output pieces;
// 16 x 8: 32-byte gathers, full 128-byte output lines.  Measured: N = 128 (48 channels) 0.83 / 0.75 ms,
// N = 160 (16 channels) 0.61 / 0.66 ms.  32 x 4 (16-byte gathers, 256-byte pieces), round 3: 1.24 ms at N = 128.
// Also measured in round 3 (all bit-identical or equal to rounding, none kept):
//  * a lane fetching TWO adjacent channel quads of a corner itself (position, weights and offsets computed once per voxel
//    instead of once per lane: a third fewer vector instructions): 1.56 ms at N = 128, 0.66 against 0.60 at N = 160 (four
//    quads per lane 0.90) -- what the gather costs is the number of (lane, instruction) line requests, and two lanes
//    reading 32 adjacent bytes in ONE instruction are one request where one lane reading them in two instructions is two;
//  * the z transform as two half-length transforms (Z[2j] = FFT_L(z), Z[2j+1] = FFT_L(z w_N^n); samples kept in registers,
//    34 KB of LDS and 63 registers: four blocks per CU instead of two): 0.90-0.93 ms against 0.77 with 2, 3 or 4 resident
//    blocks alike -- the kernel is not waiting for a free block slot, and the second set of passes and barriers costs.
//  * 16 rows x 16 channels = 128 pencils per block (64-byte gathers AND 128-byte pieces; 145 KB, one block per CU, 1024
//    threads): 0.77-0.80 ms against 0.76-0.77.
//  * round 4, N = 160 (one 89 KB block per CU): 32 pencils per block -- 44 KB, three blocks per CU -- as 8 rows x 8 channels
//    0.686 ms (32-byte gathers), as 4 rows x 16 channels 1.67 ms (32-byte OUTPUT runs: 2.7 x), 16 rows x 8 channels
//    re-measured 0.675, against 0.61 for 8 x 16; 48 ch x 80^3: 1.93 / 3.65 / 2.12 against 1.71.  Occupancy is not what this
//    kernel lacks; the run lengths of its gathers and stores are what it pays for.
#ifndef DLPD_K1_YG160
#define DLPD_K1_YG160 8
#endif
#ifndef DLPD_K1_CC160
#define DLPD_K1_CC160 16
#endif
template <int N> struct K1ClCfg {
  static constexpr int YG = (N == 128) ? 16 : (N == 160 ? DLPD_K1_YG160 : 8), CC = (N == 160) ? DLPD_K1_CC160 : 128 / YG;
  static constexpr int NP = CC * (YG / 2);             // two-row pencils per block
};
struct K1ClRot { float r0, r1, r2, r3, r4, r5, r6, r7, r8; };
DLPD_D K1ClRot k1cl_load_rotation(const float* r) {
  K1ClRot o = {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]};
  return o;
}

// Rows yrow and yrow + 1 of output plane x at depth z, four channels (one float4 of the channels-last copy `src`, which
// already points at the lane's channel quad): same corner weights, products and summation order as trilinear_fetch.
DLPD_D void k1cl_sample_rows(const float4* __restrict__ src, int Cq, int L, int ext, float c0, const K1ClRot& r, int x, int yrow,
                             int z, float4 (&acc)[2]) {
  const float dx = x - c0, dz = z - c0;
  const int hi = L - 1;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        // outside the embedded box (ext < L) the sample is cropped: all eight weights zero (no branch: the loads of both
        // rows stay batched; a branch here cost 0.06 ms at N = 128)
        const bool live = max(x, max(yrow + u, z)) < ext;
        const float dy = (yrow + u) - c0;
        const float px = c0 + (r.r0 * dx + r.r3 * dy + r.r6 * dz);
        const float py = c0 + (r.r1 * dx + r.r4 * dy + r.r7 * dz);
        const float pz = c0 + (r.r2 * dx + r.r5 * dy + r.r8 * dz);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        const float ax = px - fx, ay = py - fy, az = pz - fz;
        const bool x0 = live & (ix >= 0) & (ix <= hi), x1 = live & (ix + 1 >= 0) & (ix + 1 <= hi);
        const bool y0 = (iy >= 0) & (iy <= hi), y1 = (iy + 1 >= 0) & (iy + 1 <= hi);
        const bool z0 = (iz >= 0) & (iz <= hi), z1 = (iz + 1 >= 0) & (iz + 1 <= hi);
        const float wx0 = x0 ? 1.f - ax : 0.f, wx1 = x1 ? ax : 0.f;
        const float wy0 = y0 ? 1.f - ay : 0.f, wy1 = y1 ? ay : 0.f;
        const float wz0 = z0 ? 1.f - az : 0.f, wz1 = z1 ? az : 0.f;
        const int cx0 = min(max(ix, 0), hi), cx1 = min(max(ix + 1, 0), hi);
        const int cy0 = min(max(iy, 0), hi), cy1 = min(max(iy + 1, 0), hi);
        const int cz0 = min(max(iz, 0), hi), cz1 = min(max(iz + 1, 0), hi);
        const float4 v000 = src[(size_t)((cx0 * L + cy0) * L + cz0) * Cq], v001 = src[(size_t)((cx0 * L + cy0) * L + cz1) * Cq];
        const float4 v010 = src[(size_t)((cx0 * L + cy1) * L + cz0) * Cq], v011 = src[(size_t)((cx0 * L + cy1) * L + cz1) * Cq];
        const float4 v100 = src[(size_t)((cx1 * L + cy0) * L + cz0) * Cq], v101 = src[(size_t)((cx1 * L + cy0) * L + cz1) * Cq];
        const float4 v110 = src[(size_t)((cx1 * L + cy1) * L + cz0) * Cq], v111 = src[(size_t)((cx1 * L + cy1) * L + cz1) * Cq];
        const float w000 = wx0 * wy0 * wz0, w001 = wx0 * wy0 * wz1, w010 = wx0 * wy1 * wz0, w011 = wx0 * wy1 * wz1;
        const float w100 = wx1 * wy0 * wz0, w101 = wx1 * wy0 * wz1, w110 = wx1 * wy1 * wz0, w111 = wx1 * wy1 * wz1;
#define DLPD_TRI(f)                                                                                               \
  {                                                                                                               \
    float a = v000.f * w000;                                                                                      \
    a += v001.f * w001;                                                                                           \
    a += v010.f * w010;                                                                                           \
    a += v011.f * w011;                                                                                           \
    a += v100.f * w100;                                                                                           \
    a += v101.f * w101;                                                                                           \
    a += v110.f * w110;                                                                                           \
    a += v111.f * w111;                                                                                           \
    acc[u].f = a;                                                                                                 \
  }
        DLPD_TRI(x) DLPD_TRI(y) DLPD_TRI(z) DLPD_TRI(w)
#undef DLPD_TRI
      }
}

// dlpd_k1r.hip: the role-split formulation behind dlpd_zfft_channels_last_form(form = 2); boxes 64 and 80
int dlpd_k1_role_split(const float4* cl, const float* R, cplx* A, int C, int nb, float c0, hipStream_t st, int CT_out, int c_base,
                       int ext, int L);
