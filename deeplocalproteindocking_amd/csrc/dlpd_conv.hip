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

// ------------------------------------------------------------------------------------------------------------------
// The same convolution on the BF16 matrix cores with f32-grade results (round 4): every f32 value is split into three
// bf16 terms, x = x_h + x_m + x_l (each the round-to-nearest bf16 of what the previous ones left: 3 x 8 significant
// bits), and a product x w is summed from six bf16 x bf16 products -- hh, hm, mh, mm, hl, lh; the three dropped ones are
// below 2^-24 |x w|, the size of f32's own rounding -- accumulated in f32 by v_mfma_f32_16x16x32_bf16, which runs at 16
// x the rate of the f32-input form: 16 / 6 = 2.7 x the matrix throughput for results that differ from the exact-f32
// kernel by a few 1e-7 relative (tests: <= 1e-5 against torch in float64).  Docker.dockE3's per-batch cost is two thirds
// this convolution (Docker.py:166-167).
//   M = 16 output channels, N = 16 z-consecutive voxels, K = 32 = 4 taps x 8 input channels per MFMA.
//   block = (4 x 4 (x, y) patch, 16 z) of one volume, 16 / RW waves x RW rows; input channels in chunks of 8, staged
//   through LDS CHANNELS-LAST and already split: Xs[split][voxel][8 ch] bf16 = 16 bytes per (split, voxel), so that a B
//   fragment -- the 8 channels of one tap of one voxel -- is one ds_read_b128; lane (n, kg) of a tap group reads tap
//   4 tg + kg at voxel n (per-lane tap offsets from a small LDS table).  A fragments (weights, split by
//   dlpd_conv3d_split_pack) come straight from global memory -- 1 KB contiguous per wave and fragment, shared by all
//   blocks, L2 / L1 resident -- one tap group ahead.  LDS: 61 KB (k = 5) / 31 KB (k = 3): two to four blocks per CU, whose
//   staging and matrix phases overlap each other.
// Measured (round 4, batch 16, the nine convolutions of E3MultiResRepr4x4(8) at box 80): 9.2 ms against 17.0-17.3 ms for
// the exact-f32 kernel (86.5 for torch / MIOpen); rows per wave 1 / 2 / 4 / 8: 13.3 / 9.7 / 9.2 / 11.2 ms (a wave's A
// fragments are re-read by every wave of the block: 4 rows halve that traffic, 8 rows leave two waves per block).  Also
// built and NOT kept: B fragments one tap group ahead as well plus the next chunk's staging loads held in registers
// across the matrix phase (166-210 registers, two waves per SIMD): 9.6-9.7 ms -- the k = 3 layers lose their four
// resident blocks and the k = 5 layers do not change (2.85 against 2.78 ms): per matrix instruction the kernel needs
// half an LDS fragment read and, at 16 output channels, an eighth of a global one -- matrix pipe, LDS and the vector
// memory path each sit near half of their rate, and what would raise the arithmetic per fragment (more output-channel
// tiles per wave) the 16-channel layers do not have.  A fragments TWO tap groups ahead: 9.5 against 9.2-9.3 ms.
#ifndef DLPD_CONV_DIAG
#define DLPD_CONV_DIAG 0                     // diagnostic builds only (EXPERIMENTS.md R5): 1 no matrix instructions, 2 no staging stores, 4 no B reads, 16 no A loads in the loop, 32 no staging loads
#endif
#ifndef DLPD_CONVS_RW
#define DLPD_CONVS_RW 4
#endif
template <int KS> struct ConvSCfg {
  static constexpr int TX = 4, TY = 4, ZT = 16, RW = DLPD_CONVS_RW, NW = TX * TY / RW, NT = 64 * NW, CK = 8;   // RW rows per wave
  static constexpr int H = KS / 2;
  static constexpr int XS = TX + KS - 1, YS = TY + KS - 1, ZS = ZT + KS - 1;
  static constexpr int NVOX = XS * YS * ZS;
  static constexpr int NTAP = KS * KS * KS, NTG = (NTAP + 3) / 4;
  static constexpr size_t LDS_BYTES = (size_t)3 * NVOX * 16 + (size_t)NTG * 4 * sizeof(int) + 32 + 64;   // + any_s[8] + nbr_s[54]
};

// x = h + m + l in bf16 (bit patterns)
DLPD_D void conv_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = dlpd_f2bf(x);
  const float r1 = x - dlpd_bf2f(h);
  m = dlpd_f2bf(r1);
  const float r2 = r1 - dlpd_bf2f(m);
  l = dlpd_f2bf(r2);
}

// TILE OCCUPANCY (round 5).  The plugins' convolutions have no bias, so an output tile whose receptive field holds only
// zeros IS zero -- and a protein fills a fraction of its box (the density splat is zero a few Angstrom away from the atoms,
// and every layer only widens the non-zero region by its kernel radius).  `occ_in[volume][tile x][tile y][z cell]` (one
// byte per 4 x 4 x 4 cell of the input -- the (x, y) tiling of this kernel, a quarter of its 16-voxel z tile --, non-zero = the
// cell holds a non-zero value in some channel) lets a block whose neighbourhood -- its own and the 8 adjacent tile columns,
// over its z range widened by one cell either way: 54 cells, a superset of its halo -- is all empty skip staging and matrix
// work and write its zeros straight away; `occ_out` receives the same for the four cells it wrote (stride 1: the next layer's
// occ_in, for free).  Bit-identical: the full computation of such a
// tile adds products of zeros to a +0.0 accumulator.  Either pointer may be null (dense behaviour / no map produced).
// SPARSE = false is the kernel without any of this (its own instantiation: the dense callers run the code they always ran).
// UNWRITTEN ACTIVATIONS (round 6, `unwritten` != 0, needs occ_in): the tensors travel together with their maps -- a block
// whose neighbourhood is empty writes NOTHING (its cells stay 0 in occ_out), and no voxel of a cell that occ_in marks
// empty is ever read: the staging takes it as the zero it stands for (the cell may hold zeros a computed tile wrote, or
// memory nobody wrote).  Same values in every cell a map marks; the zeros of the other cells exist only in the map.
template <int KS, int COUT, int RELU, int STRIDE, bool SPARSE> __global__ void __launch_bounds__(ConvSCfg<KS>::NT)
k_conv3d_bf16x3(const float* __restrict__ X, const float4* __restrict__ Wp, float* __restrict__ Y, int CIN, int D,
                int cout_total, int co_base, int nzb, const unsigned char* __restrict__ occ_in,
                unsigned char* __restrict__ occ_out, int unwritten) {
  typedef ConvSCfg<KS> C;
  constexpr int MT = COUT / 16, H = C::H, NTG = C::NTG, NVOX = C::NVOX, RW = C::RW, NT = C::NT;
  DLPD_DYN_SHARED(float4, Xs);                                 // [3][NVOX] 16-byte cells
  int* toff = reinterpret_cast<int*>(Xs + 3 * NVOX);           // [4 NTG] voxel offset of every tap (0 for the padding taps)
  int* any_s = toff + 4 * NTG;                                 // [8] block-wide flags of the tile-occupancy logic
  unsigned char* nbr_s = reinterpret_cast<unsigned char*>(any_s + 8);   // [54] occ_in of the neighbourhood (0 outside the volume)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * C::TX, y0 = blockIdx.y * C::TY, b = blockIdx.z / nzb, z0 = (blockIdx.z % nzb) * C::ZT;
  const int ntx = gridDim.x, nty = gridDim.y, tzb = blockIdx.z % nzb;
  bool empty = false;
  if (SPARSE && occ_in) {                                      // (block-uniform)
    if (tid == 0) any_s[0] = 0;
    __syncthreads();
    if (tid < 54) {                                            // 3 x 3 tile columns x the 6 z cells [4 tzb - 1, 4 tzb + 4]
      const int nzs = (D + 3) / 4;
      const int nx = (int)blockIdx.x + tid / 18 - 1, ny = (int)blockIdx.y + (tid / 6) % 3 - 1, nz = 4 * tzb + tid % 6 - 1;
      const bool full = nx >= 0 && nx < ntx && ny >= 0 && ny < nty && nz >= 0 && nz < nzs &&
                        occ_in[(((size_t)b * ntx + nx) * nty + ny) * nzs + nz];
      nbr_s[tid] = full ? 1 : 0;
      if (full) any_s[0] = 1;                                  // (plain store of the same value by whoever finds one)
    }
    __syncthreads();
    empty = any_s[0] == 0;
    if (empty && unwritten) return;                            // (block-uniform) nothing read, nothing written
  }
  const bool by_map = SPARSE && occ_in && unwritten;           // voxels of empty cells are zeros that nobody may have written
  const size_t D3 = (size_t)D * D * D;
  const float* Xb = X + (size_t)b * CIN * D3;
  const int kg = lane >> 4, n = lane & 15;
  for (int t = tid; t < 4 * NTG; t += NT) {
    const int dz = t % KS, dy = (t / KS) % KS, dx = t / (KS * KS);
    toff[t] = t < C::NTAP ? (dx * C::YS + dy) * C::ZS + dz : 0;
  }
  dlpd_acc4 acc[RW][MT];
#pragma unroll
  for (int r = 0; r < RW; r++)
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[r][mt] = dlpd_acc4_zero();
  int vbase[RW];
  bool skip[RW];
#pragma unroll
  for (int r = 0; r < RW; r++) {
    const int row = RW * wave + r;
    vbase[r] = ((row / C::TY) * C::YS + (row % C::TY)) * C::ZS + n;
    skip[r] = STRIDE == 2 && (((row / C::TY) | (row % C::TY)) & 1);      // stride 2: rows of odd x or y produce no output
  }
  const int nchunk = (SPARSE && empty) ? 0 : (CIN + C::CK - 1) / C::CK;    // an empty neighbourhood: nothing to stage, the zeros go out as they are
  for (int ch = 0; ch < nchunk; ch++) {
    __syncthreads();                                           // previous chunk consumed (first pass: toff written)
    // ---- staging: one voxel (8 channels) per thread and step; split into the three bf16 planes
    for (int v = tid; v < NVOX; v += NT) {
      const int zz = v % C::ZS, yy = (v / C::ZS) % C::YS, xx = v / (C::ZS * C::YS);
      const int gx = x0 + xx - H, gy = y0 + yy - H, gz = z0 + zz - H;
      bool ok = gx >= 0 && gx < D && gy >= 0 && gy < D && gz >= 0 && gz < D;
      if (by_map)      // (gx >> 2) - blockIdx.x + 1 in 0..2 (halo <= 2), (gz >> 2) - 4 tzb + 1 in 0..5: the 54 cells loaded above
        ok = ok && nbr_s[(((gx >> 2) - (int)blockIdx.x + 1) * 3 + ((gy >> 2) - (int)blockIdx.y + 1)) * 6 + ((gz >> 2) - 4 * tzb + 1)];
      const float* src = Xb + ((size_t)(ok ? gx : 0) * D + (ok ? gy : 0)) * D + (ok ? gz : 0);
      unsigned hh[8], mm[8], ll[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int ci = C::CK * ch + j;
#if !(DLPD_CONV_DIAG & 32)
        const float x = (ok && ci < CIN) ? src[(size_t)ci * D3] : 0.f;
#else
        const float x = (ok && ci < CIN) ? (float)(v + j) : 0.f;             // (diagnostic build: no staging loads)
#endif
        conv_split3(x, hh[j], mm[j], ll[j]);
      }
      float4 ph, pm, pl;
      ph.x = __uint_as_float(hh[0] | (hh[1] << 16)); ph.y = __uint_as_float(hh[2] | (hh[3] << 16));
      ph.z = __uint_as_float(hh[4] | (hh[5] << 16)); ph.w = __uint_as_float(hh[6] | (hh[7] << 16));
      pm.x = __uint_as_float(mm[0] | (mm[1] << 16)); pm.y = __uint_as_float(mm[2] | (mm[3] << 16));
      pm.z = __uint_as_float(mm[4] | (mm[5] << 16)); pm.w = __uint_as_float(mm[6] | (mm[7] << 16));
      pl.x = __uint_as_float(ll[0] | (ll[1] << 16)); pl.y = __uint_as_float(ll[2] | (ll[3] << 16));
      pl.z = __uint_as_float(ll[4] | (ll[5] << 16)); pl.w = __uint_as_float(ll[6] | (ll[7] << 16));
#if !(DLPD_CONV_DIAG & 2)
      Xs[v] = ph;
      Xs[NVOX + v] = pm;
      Xs[2 * NVOX + v] = pl;
#else
      if (ph.x == 1.2345f) Xs[v] = pm + pl;             // (diagnostic build: the staging stores are skipped)
#endif
    }
    __syncthreads();
    // ---- all tap groups of the chunk; A fragments one group ahead
    const float4* wch = Wp + (size_t)ch * 3 * NTG * 4 * COUT + kg * COUT + n;
    float4 a_cur[3][MT], a_nxt[3][MT];
#pragma unroll
    for (int sp = 0; sp < 3; sp++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a_cur[sp][mt] = wch[(size_t)(sp * NTG) * 4 * COUT + mt * 16];
#if (DLPD_CONV_DIAG & 16)
    const float4 a_fix = a_cur[0][0];                           // (diagnostic build: no A fragment loads inside the loop)
#endif
#pragma unroll 1
    for (int tg = 0; tg < NTG; tg++) {
      const int tgn = tg + 1 < NTG ? tg + 1 : tg;
#pragma unroll
      for (int sp = 0; sp < 3; sp++)
#pragma unroll
#if !(DLPD_CONV_DIAG & 16)
        for (int mt = 0; mt < MT; mt++) a_nxt[sp][mt] = wch[(size_t)(sp * NTG + tgn) * 4 * COUT + mt * 16];
#else
        for (int mt = 0; mt < MT; mt++) { a_nxt[sp][mt] = a_fix; a_nxt[sp][mt].x += (float)(tgn + sp + mt); }
#endif
      const int off = toff[4 * tg + kg];
      float4 bf[3][RW];
#pragma unroll
      for (int sp = 0; sp < 3; sp++)
#pragma unroll
        for (int r = 0; r < RW; r++)
#if !(DLPD_CONV_DIAG & 4)
          bf[sp][r] = Xs[sp * NVOX + vbase[r] + off];
#elif (DLPD_CONV_DIAG & 8)
        {                                                                 // (diagnostic build: no B reads, PSEUDO-RANDOM operand bits)
          unsigned h = (unsigned)(off * 2654435761u) ^ (unsigned)(lane * 40503u) ^ (unsigned)((sp * 7 + r) * 2246822519u) ^ (unsigned)tg * 3266489917u;
          h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
          const unsigned m = 0x3FFF3FFFu;                                 // finite bf16 pairs (exponent below all-ones)
          bf[sp][r] = make_float4(__uint_as_float(h & m), __uint_as_float((h * 3u) & m), __uint_as_float((h * 5u) & m), __uint_as_float((h * 7u) & m));
        }
#else
          bf[sp][r] = make_float4((float)off, (float)sp, (float)r, 1.f);   // (diagnostic build: no B fragment reads, constant operands)
#endif
      DLPD_SCHED_FENCE();
#pragma unroll
      for (int r = 0; r < RW; r++) {
        if (skip[r]) continue;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          dlpd_acc4 c = acc[r][mt];
#if !(DLPD_CONV_DIAG & 1)
          c = DLPD_MFMA_16x16x32_BF16(a_cur[2][mt], bf[0][r], c);      // w_l x_h   (small terms first)
          c = DLPD_MFMA_16x16x32_BF16(a_cur[0][mt], bf[2][r], c);      // w_h x_l
          c = DLPD_MFMA_16x16x32_BF16(a_cur[1][mt], bf[1][r], c);      // w_m x_m
          c = DLPD_MFMA_16x16x32_BF16(a_cur[1][mt], bf[0][r], c);      // w_m x_h
          c = DLPD_MFMA_16x16x32_BF16(a_cur[0][mt], bf[1][r], c);      // w_h x_m
          c = DLPD_MFMA_16x16x32_BF16(a_cur[0][mt], bf[0][r], c);      // w_h x_h
#else
          // (diagnostic build: the matrix instructions replaced by a few vector adds that keep every operand alive)
          c[0] += a_cur[0][mt].x + a_cur[1][mt].y + a_cur[2][mt].z + bf[0][r].x + bf[1][r].y + bf[2][r].z;
#endif
          acc[r][mt] = c;
        }
      }
      DLPD_SCHED_FENCE();
#pragma unroll
      for (int sp = 0; sp < 3; sp++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a_cur[sp][mt] = a_nxt[sp][mt];
    }
  }
  // ---- epilogue: lane holds output channels 4 kg + j, voxel n (the C/D layout of every 16x16 form)
  const int kq = kg;
  bool nonzero = false;
#pragma unroll
  for (int r = 0; r < RW; r++) {
    const int row = RW * wave + r, gx = x0 + row / C::TY, gy = y0 + row % C::TY, gz = z0 + n;
    if (gx >= D || gy >= D || gz >= D) continue;
    if (STRIDE == 2 && ((gx | gy | gz) & 1)) continue;
    const int Do = (STRIDE == 2) ? (D - 1) / 2 + 1 : D;
    const size_t Do3 = (size_t)Do * Do * Do;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float v = dlpd_acc4_get(acc[r][mt], j);
        if (RELU) v = fmaxf(v, 0.f);
        if (SPARSE) nonzero |= v != 0.f;
        Y[((size_t)b * cout_total + co_base + mt * 16 + 4 * kq + j) * Do3 +
          ((size_t)(gx / STRIDE) * Do + gy / STRIDE) * Do + gz / STRIDE] = v;
      }
  }
  if (SPARSE && occ_out && STRIDE == 1) {                      // the written tile's own occupancy, per 4-voxel z cell (zeroed by the host)
    if (tid < 4) any_s[4 + tid] = 0;
    __syncthreads();
    if (nonzero) any_s[4 + (n >> 2)] = 1;                      // (a lane's outputs are the channels of ONE voxel z0 + n)
    __syncthreads();
    const int nzs = (D + 3) / 4;
    if (tid < 4 && 4 * tzb + tid < nzs && any_s[4 + tid])
      occ_out[(((size_t)b * ntx + blockIdx.x) * nty + blockIdx.y) * nzs + 4 * tzb + tid] = 1;
  }
}

// occ[volume][tile x][tile y][z cell] = 1 where the 4 x 4 x 4 cell of x (B, CIN, D^3) holds a non-zero value in some channel
// ((x, y) tiles of k_conv3d_bf16x3, a quarter of its z tile).  One block per (tile x, tile y, volume): thread = (x, y, z mod 16),
// z tiles in turn.
__global__ void __launch_bounds__(256) k_conv3d_tile_occupancy(const float* __restrict__ X, unsigned char* __restrict__ occ,
                                                               int CIN, int D, int nzb) {
  __shared__ int any_s[4];
  const int tid = threadIdx.x, zl = tid & 15, xy = tid >> 4;
  const int gx = blockIdx.x * 4 + (xy >> 2), gy = blockIdx.y * 4 + (xy & 3), b = blockIdx.z;
  const size_t D3 = (size_t)D * D * D;
  const int nzs = (D + 3) / 4;
  const float* src = X + (size_t)b * CIN * D3 + ((size_t)gx * D + gy) * D;
  for (int zt = 0; zt < nzb; zt++) {
    const int gz = 16 * zt + zl;
    bool nz = false;
    if (gx < D && gy < D && gz < D)
      for (int c = 0; c < CIN; c++) nz |= src[(size_t)c * D3 + gz] != 0.f;
    __syncthreads();
    if (tid < 4) any_s[tid] = 0;
    __syncthreads();
    if (nz) any_s[zl >> 2] = 1;
    __syncthreads();
    if (tid < 4 && 4 * zt + tid < nzs)
      occ[(((size_t)b * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * nzs + 4 * zt + tid] = any_s[tid] ? 1 : 0;
  }
}

// wp[group][chunk][split][tap group][kg][co in group][8 ch] bf16: the A fragments in the order the kernel reads them
__global__ void __launch_bounds__(256) k_conv3d_split_pack(const float* __restrict__ w, unsigned short* __restrict__ wp, int cin,
                                                           int cout, int ks) {
  const int ntap = ks * ks * ks, ntg = (ntap + 3) / 4, nch = (cin + 7) / 8;
  const size_t per32 = (size_t)nch * 3 * ntg * 4 * 32 * 8;     // bf16 elements of a full group
  const size_t total = (size_t)nch * 3 * ntg * 4 * cout * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i / per32), gw = conv_group_width(cout, g);
    size_t r = i - (size_t)g * per32;
    const int j = (int)(r % 8); r /= 8;
    const int col = (int)(r % gw); r /= gw;
    const int kg = (int)(r % 4); r /= 4;
    const int tg = (int)(r % ntg); r /= ntg;
    const int sp = (int)(r % 3); r /= 3;
    const int ch = (int)r;
    const int tap = 4 * tg + kg, ci = 8 * ch + j, co = 32 * g + col;
    const float x = (tap < ntap && ci < cin) ? w[((size_t)co * cin + ci) * ntap + tap] : 0.f;
    unsigned h, m, l;
    conv_split3(x, h, m, l);
    wp[i] = (unsigned short)(sp == 0 ? h : (sp == 1 ? m : l));
  }
}

template <int KS, int COUT> static int launch_conv_split(const float* X, const float4* W, float* Y, int B, int CIN, int D,
                                                         int relu, int stride, int cout_total, int co_base, hipStream_t st,
                                                         const unsigned char* occ_in, unsigned char* occ_out, int unwritten) {
  typedef ConvSCfg<KS> C;
  const int nzb = (D + C::ZT - 1) / C::ZT;
  dim3 grid((D + C::TX - 1) / C::TX, (D + C::TY - 1) / C::TY, B * nzb), block(C::NT);
#define DLPD_CS1(R, S, SP) { int rc = dlpd_set_max_dyn_shared((const void*)k_conv3d_bf16x3<KS, COUT, R, S, SP>, C::LDS_BYTES); if (rc) return rc; \
    DLPD_LAUNCH((k_conv3d_bf16x3<KS, COUT, R, S, SP>), grid, block, C::LDS_BYTES, st, X, W, Y, CIN, D, cout_total, co_base, nzb, \
                occ_in, occ_out, unwritten); }
#define DLPD_CS(R, S) { if (occ_in || occ_out) DLPD_CS1(R, S, true) else DLPD_CS1(R, S, false) }
  if (stride == 2) { if (relu) DLPD_CS(1, 2) else DLPD_CS(0, 2) }
  else { if (relu) DLPD_CS(1, 1) else DLPD_CS(0, 1) }
#undef DLPD_CS
#undef DLPD_CS1
  return dlpd_check_launch();
}

// MaxPool3d(kernel 5, stride 2, padding 2) of the E3 plugin (ProteinRepresentationModels.py:101): out-of-range
// taps do not take part (torch pads with -inf).
// Tiled (round 5: a one-thread-per-output kernel took 1.3 ms of the E3 plugin's 4.5 per batch -- 125
// strided reads per output).  A block produces a 4 x 4 x 20 tile of one volume: its 11 x 11 x 43 inputs go to LDS once (43-float
// runs; -inf outside the volume, as torch pads), then the window maximum is taken one axis at a time -- z, y, x: 5 + 5 + 5
// comparisons instead of 125; a maximum does not depend on the order it is taken in.  occ_in (the input's occupancy cells,
// dlpd_conv3d_tile_occupancy; C channels per map volume): a tile whose inputs are all in empty cells is +0.0 and is written
// without reading them; occ_out: the cells of the OUTPUT grid that hold a non-zero value (zeroed by the host; any channel).
#define DLPD_MP_OX 4
#define DLPD_MP_OZ 20
// Round 6, second form: a block walks CPB channels of its (volume, tile) -- the channels of a batch entry share one map and
// one geometry, so the occupancy test, the cells of the region and every thread's staging offsets (21 of them: where in the
// volume, "outside" or "empty cell") are made ONCE and a channel costs 21 loads and LDS stores per thread; the first form
// (a block per channel) spent its time on that index arithmetic: 0.39-0.44 ms per batch of 16 at box 80.
template <int CPB> __global__ void __launch_bounds__(256) k_maxpool3d_5s2_tiled(const float* __restrict__ x, float* __restrict__ y, int D, int Do,
                                                             int nzt, int C, const unsigned char* __restrict__ occ_in,
                                                             unsigned char* __restrict__ occ_out, int unwritten) {
  constexpr int OX = DLPD_MP_OX, OZ = DLPD_MP_OZ, IX = 2 * OX + 3, IZ = 2 * OZ + 3;
  constexpr int NIN = IX * IX * IZ, NST = (NIN + 255) / 256;
  __shared__ float in[NIN];                      // 20.8 KB
  __shared__ float m1[IX * IX * OZ];             // max over z
  __shared__ float m2[IX * OX * OZ];             // ... and y
  __shared__ int flag[1], oflag[8];              // "some input cell is occupied"; the output cells that hold a non-zero value
  // the occupancy cells of the block's input region (at most 4 x 4 x 12)
  constexpr int CX = (IX + 2) / 4 + 1, CZ = (IZ + 2) / 4 + 1;
  __shared__ unsigned char cell[CX * CX * CZ];
  const int tid = threadIdx.x;
  const int ncg = (C + CPB - 1) / CPB;           // channel groups of a batch entry
  const int zt = blockIdx.z % nzt, cg = (blockIdx.z / nzt) % ncg, b = blockIdx.z / (nzt * ncg);
  const int c0 = cg * CPB, nch = (C - c0 < CPB) ? C - c0 : CPB;
  const int ox0 = blockIdx.x * OX, oy0 = blockIdx.y * OX, oz0 = zt * OZ;
  const int ix0 = 2 * ox0 - 2, iy0 = 2 * oy0 - 2, iz0 = 2 * oz0 - 2;
  const size_t D3 = (size_t)D * D * D, Do3 = (size_t)Do * Do * Do;
  const float* src = x + ((size_t)b * C + c0) * D3;
  float* dst = y + ((size_t)b * C + c0) * Do3;
  const int nc = (D + 3) / 4;
  const int cx0 = (ix0 < 0 ? 0 : ix0) >> 2, cy0 = (iy0 < 0 ? 0 : iy0) >> 2, cz0 = (iz0 < 0 ? 0 : iz0) >> 2;
  bool empty = false;
  if (occ_in) {                                  // (block-uniform) the occupancy cells the input region touches
    const int cx1 = (ix0 + IX - 1 >= D ? D - 1 : ix0 + IX - 1) >> 2, cy1 = (iy0 + IX - 1 >= D ? D - 1 : iy0 + IX - 1) >> 2;
    const int cz1 = (iz0 + IZ - 1 >= D ? D - 1 : iz0 + IZ - 1) >> 2;
    const int nx = cx1 - cx0 + 1, ny = cy1 - cy0 + 1, nz = cz1 - cz0 + 1;
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    for (int i = tid; i < nx * ny * nz; i += 256) {
      const int cz = i % nz, cy = (i / nz) % ny, cx = i / (nz * ny);
      const unsigned char o = occ_in[(((size_t)b * nc + cx0 + cx) * nc + cy0 + cy) * nc + cz0 + cz];
      cell[(cx * CX + cy) * CZ + cz] = o;
      if (o) flag[0] = 1;
    }
    __syncthreads();
    empty = flag[0] == 0;
  }
  if (empty && unwritten) return;                // unwritten activations (k_conv3d_bf16x3): the map says it all
  if (empty) {
    for (int c = 0; c < nch; c++)
      for (int i = tid; i < OX * OX * OZ; i += 256) {
        const int oz = oz0 + i % OZ, oy = oy0 + (i / OZ) % OX, ox = ox0 + i / (OZ * OX);
        if (ox < Do && oy < Do && oz < Do) dst[c * Do3 + ((size_t)ox * Do + oy) * Do + oz] = 0.f;
      }
    return;
  }
  // where each of this thread's staged voxels comes from: an offset into the volume, -1 = outside (torch pads with -inf),
  // -2 = a voxel of an empty cell under unwritten activations (the zero its map stands for; not read)
  int off[NST];
  {
    const bool by_map = unwritten && occ_in;
#pragma unroll
    for (int k = 0; k < NST; k++) {
      const int i = tid + 256 * k;
      const int zz = i % IZ, yy = (i / IZ) % IX, xx = i / (IZ * IX);
      const int gx = ix0 + xx, gy = iy0 + yy, gz = iz0 + zz;
      const bool ok = i < NIN && gx >= 0 && gx < D && gy >= 0 && gy < D && gz >= 0 && gz < D;
      int o = ok ? (gx * D + gy) * D + gz : -1;
      if (ok && by_map && !cell[(((gx >> 2) - cx0) * CX + (gy >> 2) - cy0) * CZ + (gz >> 2) - cz0]) o = -2;
      off[k] = o;
    }
  }
  if (tid < 8) oflag[tid] = 0;
  __syncthreads();
  for (int c = 0; c < nch; c++) {
    // (no barrier between channels: the next one's staging writes `in`, last read two barriers ago)
    const float* sc = src + c * D3;
#pragma unroll
    for (int k = 0; k < NST; k++) {
      const int i = tid + 256 * k;
      const float v = off[k] >= 0 ? sc[off[k]] : (off[k] == -1 ? -INFINITY : 0.f);
      if (i < NIN) in[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (IX * IX * OZ + 255) / 256; k++) {
      const int i = tid + 256 * k;
      if (i < IX * IX * OZ) {
        const int oz = i % OZ, r = i / OZ;
        const float* p = in + r * IZ + 2 * oz;
        m1[i] = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), p[4]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (IX * OX * OZ + 255) / 256; k++) {
      const int i = tid + 256 * k;
      if (i < IX * OX * OZ) {
        const int oz = i % OZ, oy = (i / OZ) % OX, xx = i / (OZ * OX);
        const float* p = m1 + (xx * IX + 2 * oy) * OZ + oz;
        m2[i] = fmaxf(fmaxf(fmaxf(p[0], p[OZ]), fmaxf(p[2 * OZ], p[3 * OZ])), p[4 * OZ]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (OX * OX * OZ + 255) / 256; k++) {
      const int i = tid + 256 * k;
      if (i < OX * OX * OZ) {
        const int oz = i % OZ, oy = (i / OZ) % OX, ox = i / (OZ * OX);
        const float* p = m2 + ((2 * ox) * OX + oy) * OZ + oz;
        const float v = fmaxf(fmaxf(fmaxf(p[0], p[OX * OZ]), fmaxf(p[2 * OX * OZ], p[3 * OX * OZ])), p[4 * OX * OZ]);
        if (ox0 + ox < Do && oy0 + oy < Do && oz0 + oz < Do) {
          dst[c * Do3 + ((size_t)(ox0 + ox) * Do + oy0 + oy) * Do + oz0 + oz] = v;
          if (occ_out && v != 0.f) oflag[oz >> 2] = 1;            // (OZ = 20: five 4-voxel cells, aligned: oz0 is a multiple of 20)
        }
      }
    }
  }
  if (occ_out) {
    __syncthreads();
    const int nco = (Do + 3) / 4;
    if (tid < OZ / 4 && (oz0 >> 2) + tid < nco && oflag[tid])
      occ_out[(((size_t)b * nco + blockIdx.x) * nco + blockIdx.y) * nco + (oz0 >> 2) + tid] = 1;
  }
}

extern "C" {

size_t dlpd_conv3d_tile_occupancy_bytes(int B, int D);

int dlpd_maxpool3d_5s2_sparse(const float* x, float* y, const unsigned char* occ_in, unsigned char* occ_out, int B, int C, int D,
                              int unwritten, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || D < 1 || (unwritten && !occ_in)) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Do = (D + 4 - 5) / 2 + 1;
  const int nzt = (Do + DLPD_MP_OZ - 1) / DLPD_MP_OZ, nxy = (Do + DLPD_MP_OX - 1) / DLPD_MP_OX;
  if (occ_out && hipMemsetAsync(occ_out, 0, dlpd_conv3d_tile_occupancy_bytes(B, Do), st) != hipSuccess) return DLPD_ERR_LAUNCH;
  // a block takes up to 8 channels of one batch entry (one map, one geometry: k_maxpool3d_5s2_tiled); a grid's z extent holds
  // 65,535 blocks: more (entry, channel group, z tile) triples than that go in several launches of whole batch entries
#ifndef DLPD_POOL_CPB
#define DLPD_POOL_CPB 8
#endif
  constexpr int CPB = DLPD_POOL_CPB;
  const int ncg = (C + CPB - 1) / CPB;
  const int per = 65535 / (ncg * nzt);                          // batch entries per launch
  if (per < 1) return DLPD_ERR_UNSUPPORTED;
  const size_t D3 = (size_t)D * D * D, Do3 = (size_t)Do * Do * Do;
  if (D3 * (size_t)(C < CPB ? C : CPB) > 0x7fffffffu) return DLPD_ERR_UNSUPPORTED;     // (32-bit offsets inside a volume)
  for (int b0 = 0; b0 < B; b0 += per) {
    const int nbl = (B - b0 < per) ? B - b0 : per;
    DLPD_LAUNCH((k_maxpool3d_5s2_tiled<CPB>), dim3(nxy, nxy, nbl * ncg * nzt), dim3(256), 0, st, x + (size_t)b0 * C * D3, y + (size_t)b0 * C * Do3,
                D, Do, nzt, C, occ_in ? occ_in + dlpd_conv3d_tile_occupancy_bytes(b0, D) : nullptr,
                occ_out ? occ_out + dlpd_conv3d_tile_occupancy_bytes(b0, Do) : nullptr, unwritten);
    int rc = dlpd_check_launch();
    if (rc) return rc;
  }
  return DLPD_OK;
}

int dlpd_maxpool3d_5s2(const float* x, float* y, int nvol, int D, void* stream) {
  if (!x || !y || nvol <= 0 || D < 1) return DLPD_ERR_ARG;
  const int Do = (D + 4 - 5) / 2 + 1;
  // (no maps: the volumes are channels of ONE entry to the kernel -- a block then amortises its index arithmetic over DLPD_POOL_CPB of them)
  if ((size_t)((nvol + DLPD_POOL_CPB - 1) / DLPD_POOL_CPB) * ((Do + DLPD_MP_OZ - 1) / DLPD_MP_OZ) <= 65535)
    return dlpd_maxpool3d_5s2_sparse(x, y, nullptr, nullptr, 1, nvol, D, 0, stream);
  return dlpd_maxpool3d_5s2_sparse(x, y, nullptr, nullptr, nvol, 1, D, 0, stream);      // (chunked launches for many volumes)
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

size_t dlpd_conv3d_split_packed_bytes(int cin, int cout, int ks) {
  return (size_t)((cin + 7) / 8) * 3 * ((ks * ks * ks + 3) / 4) * 4 * cout * 8 * sizeof(unsigned short);
}

int dlpd_conv3d_split_pack(const float* w, void* wp, int cin, int cout, int ks, void* stream) {
  if (!w || !wp || cin <= 0 || cout <= 0 || ks <= 0) return DLPD_ERR_ARG;
  const size_t total = dlpd_conv3d_split_packed_bytes(cin, cout, ks) / sizeof(unsigned short);
  size_t nblk = (total + 255) / 256;
  if (nblk > 65535) nblk = 65535;
  DLPD_LAUNCH(k_conv3d_split_pack, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wp, cin, cout, ks);
  return dlpd_check_launch();
}

size_t dlpd_conv3d_tile_occupancy_bytes(int B, int D) {
  return (size_t)B * ((D + 3) / 4) * ((D + 3) / 4) * ((D + 3) / 4);
}

int dlpd_conv3d_tile_occupancy(const float* x, unsigned char* occ, int B, int cin, int D, void* stream) {
  if (!x || !occ || B <= 0 || cin <= 0 || D <= 0) return DLPD_ERR_ARG;
  DLPD_LAUNCH(k_conv3d_tile_occupancy, dim3((D + 3) / 4, (D + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, x, occ, cin, D,
              (D + 15) / 16);
  return dlpd_check_launch();
}

int dlpd_conv3d_split_sparse(const float* x, const void* wp, float* y, const unsigned char* occ_in, unsigned char* occ_out,
                             int B, int cin, int cout, int D, int ks, int relu, int stride, int unwritten, void* stream);

int dlpd_conv3d_split(const float* x, const void* wp, float* y, int B, int cin, int cout, int D, int ks, int relu, int stride,
                      void* stream) {
  return dlpd_conv3d_split_sparse(x, wp, y, nullptr, nullptr, B, cin, cout, D, ks, relu, stride, 0, stream);
}

int dlpd_conv3d_split_sparse(const float* x, const void* wp, float* y, const unsigned char* occ_in, unsigned char* occ_out,
                             int B, int cin, int cout, int D, int ks, int relu, int stride, int unwritten, void* stream) {
  if (!x || !wp || !y || B <= 0 || (stride != 1 && stride != 2)) return DLPD_ERR_ARG;
  // unwritten activations: the output is only meaningful with its map, which the strided form does not produce
  if (unwritten && (!occ_in || !occ_out || stride != 1)) return DLPD_ERR_ARG;
  if (!dlpd_conv3d_supported(cin, cout, ks, D)) return DLPD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (stride != 1) occ_out = nullptr;                          // (the strided output has another tiling)
  if (occ_out && hipMemsetAsync(occ_out, 0, dlpd_conv3d_tile_occupancy_bytes(B, D), st) != hipSuccess) return DLPD_ERR_LAUNCH;
  const size_t per32 = (size_t)((cin + 7) / 8) * 3 * ((ks * ks * ks + 3) / 4) * 4 * 32 * 8 * sizeof(unsigned short) / 16;   // float4 cells
  for (int g = 0; 32 * g < cout; g++) {
    const int gw = conv_group_width(cout, g), base = 32 * g;
    const float4* wg = reinterpret_cast<const float4*>(wp) + per32 * g;
    int rc = DLPD_ERR_UNSUPPORTED;
    if (ks == 3 && gw == 16) rc = launch_conv_split<3, 16>(x, wg, y, B, cin, D, relu, stride, cout, base, st, occ_in, occ_out, unwritten);
    else if (ks == 3 && gw == 32) rc = launch_conv_split<3, 32>(x, wg, y, B, cin, D, relu, stride, cout, base, st, occ_in, occ_out, unwritten);
    else if (ks == 5 && gw == 16) rc = launch_conv_split<5, 16>(x, wg, y, B, cin, D, relu, stride, cout, base, st, occ_in, occ_out, unwritten);
    else if (ks == 5 && gw == 32) rc = launch_conv_split<5, 32>(x, wg, y, B, cin, D, relu, stride, cout, base, st, occ_in, occ_out, unwritten);
    if (rc) return rc;
  }
  return DLPD_OK;
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
