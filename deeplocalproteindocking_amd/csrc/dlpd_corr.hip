// Rotation x translation correlation search: the FFT pipeline kernels (gfx950 / CDNA4).
//
// Reference path being replaced (file:line in /root/reference):
//   src/Docker/Docker.py:218            VolumeRotation of the ligand representation volumes
//   src/Docker/Docker.py:225-226        clash correlation + threshold
//   src/Models/DockingModels.py:70-83   per-channel VolumeConvolution, concat, SimpleFilter MLP
//   src/Docker/Docker.py:232            mask multiply
//
// Pipeline per rotation (volumes (CT,L,L,L), N = 2L, NZ = N/2+1):
//   K1 k_rotate_zfft   trilinear rotation fused with the z-axis R2C FFT (two real rows per
//                      complex pencil)                           -> A  [b][c][kz][x][y]   (L x L)
//   K2 k_xy_corr       per (c,kz): zero-padded 2-D forward FFT over (x,y) in LDS, multiply by
//                      the receptor spectrum (conj), 2-D inverse  -> Bw [b][c][kz][x'][y'] (N x N)
//   K3 k_zifft_filter  per (x', y'-tile): z-axis C2R for all channels, clip, per-voxel MLP,
//                      clash mask                                 -> V  [b][x'][y'][z']
// The rotated volumes, the ligand spectrum and the 48 real correlation volumes never exist in
// HBM; only A (0.25x of a spectrum) and Bw (one spectrum) do.
#include <dlpd_platform.h>
#include "dlpd_fft.h"
#include "dlpd_internal.h"
#include "dlpd_k1.h"
#include "dlpd_k3.h"

#ifndef DLPD_K1CL_PAD
#define DLPD_K1CL_PAD 13                 // pencil row padding of the channels-last K1 (complex elements)
#endif
#ifndef DLPD_K1CL_PXOR
#define DLPD_K1CL_PXOR 0                 // diagnostic builds only (EXPERIMENTS.md R5)
#endif
#define DLPD_K1_UNROLL 2                 // samples per thread whose gathers are issued together (2..16 measured equal: not latency-bound)
template <int N> DLPD_D void init_twiddles(cplx* tw, int tid, int nthreads) {
  for (int k = tid; k < N; k += nthreads) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    tw[k] = c_make((float)c, (float)s);
  }
}

// ------------------------------------------------------------------------------------------
// trilinear sample of a (L,L,L) volume at position (px,py,pz), zeros outside
// ------------------------------------------------------------------------------------------
DLPD_D float trilinear_fetch(const float* __restrict__ v, int L, float px, float py, float pz) {
  // branch-free: out-of-box corners get weight 0 and a clamped (valid) address, so all loads are
  // unconditional and in flight together; the two z-neighbours come from ONE 8-byte load
  // (half the address-unit work of eight scalar gathers)
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  const float ax = px - fx, ay = py - fy, az = pz - fz;
  const int hi = L - 1;
  const bool x0 = (ix >= 0) & (ix <= hi), x1 = (ix + 1 >= 0) & (ix + 1 <= hi);
  const bool y0 = (iy >= 0) & (iy <= hi), y1 = (iy + 1 >= 0) & (iy + 1 <= hi);
  const bool z0 = (iz >= 0) & (iz <= hi), z1 = (iz + 1 >= 0) & (iz + 1 <= hi);
  const float wx0 = x0 ? 1.f - ax : 0.f, wx1 = x1 ? ax : 0.f;
  const float wy0 = y0 ? 1.f - ay : 0.f, wy1 = y1 ? ay : 0.f;
  const float wz0 = z0 ? 1.f - az : 0.f, wz1 = z1 ? az : 0.f;
  const int cx0 = min(max(ix, 0), hi), cx1 = min(max(ix + 1, 0), hi);
  const int cy0 = min(max(iy, 0), hi), cy1 = min(max(iy + 1, 0), hi);
  const int zb = min(max(iz, 0), hi - 1);          // pair (zb, zb+1) always inside the row
  const int d = iz - zb;                           // 0 inside; -1 / +1 at the two faces
  DLPD_PAIR p00 = dlpd_load_pair(v + (cx0 * L + cy0) * L + zb);
  DLPD_PAIR p01 = dlpd_load_pair(v + (cx0 * L + cy1) * L + zb);
  DLPD_PAIR p10 = dlpd_load_pair(v + (cx1 * L + cy0) * L + zb);
  DLPD_PAIR p11 = dlpd_load_pair(v + (cx1 * L + cy1) * L + zb);
  // value at z0 = iz is .x unless iz = zb+1 ; value at z1 = iz+1 is .y unless iz+1 = zb
  const float v000 = d > 0 ? p00.y : p00.x, v001 = d < 0 ? p00.x : p00.y;
  const float v010 = d > 0 ? p01.y : p01.x, v011 = d < 0 ? p01.x : p01.y;
  const float v100 = d > 0 ? p10.y : p10.x, v101 = d < 0 ? p10.x : p10.y;
  const float v110 = d > 0 ? p11.y : p11.x, v111 = d < 0 ? p11.x : p11.y;
  float acc = v000 * (wx0 * wy0 * wz0);
  acc += v001 * (wx0 * wy0 * wz1);
  acc += v010 * (wx0 * wy1 * wz0);
  acc += v011 * (wx0 * wy1 * wz1);
  acc += v100 * (wx1 * wy0 * wz0);
  acc += v101 * (wx1 * wy0 * wz1);
  acc += v110 * (wx1 * wy1 * wz0);
  acc += v111 * (wx1 * wy1 * wz1);
  return acc;
}

// Same sample from the QUAD layout of a volume: q[x][y][z] (y, z < L-1) = {v(x,y,z), v(x,y,z+1), v(x,y+1,z),
// v(x,y+1,z+1)} as one float4, so the eight corners are TWO 16-byte gathers instead of four 8-byte ones: the
// gather is bound by the number of cache lines its instructions touch, and this halves the instructions.
// Same weights, same products, same summation order as trilinear_fetch (bit-identical result).
DLPD_D float trilinear_fetch_quads(const float4* __restrict__ q, int L, float px, float py, float pz) {
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  const float ax = px - fx, ay = py - fy, az = pz - fz;
  const int hi = L - 1;
  const bool x0 = (ix >= 0) & (ix <= hi), x1 = (ix + 1 >= 0) & (ix + 1 <= hi);
  const bool y0 = (iy >= 0) & (iy <= hi), y1 = (iy + 1 >= 0) & (iy + 1 <= hi);
  const bool z0 = (iz >= 0) & (iz <= hi), z1 = (iz + 1 >= 0) & (iz + 1 <= hi);
  const float wx0 = x0 ? 1.f - ax : 0.f, wx1 = x1 ? ax : 0.f;
  const float wy0 = y0 ? 1.f - ay : 0.f, wy1 = y1 ? ay : 0.f;
  const float wz0 = z0 ? 1.f - az : 0.f, wz1 = z1 ? az : 0.f;
  const int cx0 = min(max(ix, 0), hi), cx1 = min(max(ix + 1, 0), hi);
  const int yb = min(max(iy, 0), hi - 1), zb = min(max(iz, 0), hi - 1);   // quad (yb..yb+1, zb..zb+1) inside the box
  const int dy = iy - yb, dz = iz - zb;              // 0 inside; -1 / +1 at the faces
  const int Q = L - 1;
  const float4 qa = q[((size_t)cx0 * Q + yb) * Q + zb];
  const float4 qb = q[((size_t)cx1 * Q + yb) * Q + zb];
  // rows y0 = iy and y1 = iy + 1 of the quad (a row outside the box has weight 0, any value will do)
  const float a_y0z0 = dy > 0 ? qa.z : qa.x, a_y0z1 = dy > 0 ? qa.w : qa.y;
  const float a_y1z0 = dy < 0 ? qa.x : qa.z, a_y1z1 = dy < 0 ? qa.y : qa.w;
  const float b_y0z0 = dy > 0 ? qb.z : qb.x, b_y0z1 = dy > 0 ? qb.w : qb.y;
  const float b_y1z0 = dy < 0 ? qb.x : qb.z, b_y1z1 = dy < 0 ? qb.y : qb.w;
  const float v000 = dz > 0 ? a_y0z1 : a_y0z0, v001 = dz < 0 ? a_y0z0 : a_y0z1;
  const float v010 = dz > 0 ? a_y1z1 : a_y1z0, v011 = dz < 0 ? a_y1z0 : a_y1z1;
  const float v100 = dz > 0 ? b_y0z1 : b_y0z0, v101 = dz < 0 ? b_y0z0 : b_y0z1;
  const float v110 = dz > 0 ? b_y1z1 : b_y1z0, v111 = dz < 0 ? b_y1z0 : b_y1z1;
  float acc = v000 * (wx0 * wy0 * wz0);
  acc += v001 * (wx0 * wy0 * wz1);
  acc += v010 * (wx0 * wy1 * wz0);
  acc += v011 * (wx0 * wy1 * wz1);
  acc += v100 * (wx1 * wy0 * wz0);
  acc += v101 * (wx1 * wy0 * wz1);
  acc += v110 * (wx1 * wy1 * wz0);
  acc += v111 * (wx1 * wy1 * wz1);
  return acc;
}

// (nvol, L, L, L) -> quad layout (nvol, L, L-1, L-1) float4
__global__ void __launch_bounds__(256) k_make_quads(const float* __restrict__ v, float4* __restrict__ q, int nvol, int L) {
  const int Q = L - 1;
  const size_t total = (size_t)nvol * L * Q * Q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int z = (int)(i % Q), y = (int)((i / Q) % Q);
    const size_t xv = i / ((size_t)Q * Q);               // vol * L + x
    const float* p = v + (xv * L + y) * L + z;
    q[i] = make_float4(p[0], p[1], p[L], p[L + 1]);
  }
}

// ------------------------------------------------------------------------------------------
// Standalone rotation (TPL VolumeRotation equivalent, Docker.py:218): out[b,c] = rot(vol[b,c])
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rotate(const float* __restrict__ vol, const float* __restrict__ R,
                                                float* __restrict__ out, int B, int C, int L,
                                                long long vol_bstride, float c0) {
  const size_t L3 = (size_t)L * L * L;
  const size_t total = (size_t)B * C * L3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int z = (int)(i % L), y = (int)((i / L) % L), x = (int)((i / ((size_t)L * L)) % L);
    const int c = (int)((i / L3) % C), b = (int)(i / (L3 * C));
    const float* r = R + (size_t)b * 9;
    const float dx = x - c0, dy = y - c0, dz = z - c0;
    const float px = c0 + (r[0] * dx + r[3] * dy + r[6] * dz);
    const float py = c0 + (r[1] * dx + r[4] * dy + r[7] * dz);
    const float pz = c0 + (r[2] * dx + r[5] * dy + r[8] * dz);
    out[i] = trilinear_fetch(vol + (size_t)b * vol_bstride + (size_t)c * L3, L, px, py, pz);
  }
}

// ------------------------------------------------------------------------------------------
// K1: (optional) rotation + z-axis R2C.  1-D grid of nb*CT*L blocks (XCD-aware decode), block (L/2)*T threads.
//   vol   (.., CT, L,L,L), batch stride vol_bstride (0: one ligand shared by every rotation)
//   R     (nb, 9) row-major rotation matrices (ignored when do_rotate == 0)
//   A     (nb, CT, NZ, L, L) complex, [kz][x][y]
// ------------------------------------------------------------------------------------------
#ifdef DLPD_STAMPS
__device__ unsigned long long dlpd_stamps_k1[16];
extern "C" int dlpd_debug_read_stamps_k1(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(dlpd_stamps_k1), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k1), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
template <int N> __global__ void __launch_bounds__((N / 4) * FftPlan<N>::T)
k_rotate_zfft(const float* __restrict__ vol, const float* __restrict__ R, cplx* __restrict__ A,
              int CT, int nb, long long vol_bstride, int do_rotate, float c0, int CT_out, int c_base,
              int transposed, const float4* __restrict__ quads, int ext, const unsigned char* __restrict__ occ, int skip_empty) {
  constexpr int L = N / 2, NZ = N / 2 + 1, RS = N + 1, NP = L / 2;
  constexpr int T = FftPlan<N>::T, R1 = FftPlan<N>::R1, R2 = FftPlan<N>::R2;
  constexpr int NT = NP * T;
  __shared__ cplx S[NP * RS + N];
  cplx* tw = S + NP * RS;
  const int tid = threadIdx.x;
  // XCD-aware decode: consecutive block ids are dealt round-robin over the 8 XCDs, so the L
  // x-planes of one (b,c) volume are given ids of equal (id % 8): they run on one XCD and the
  // 1 MiB source volume is fetched into ONE L2 instead of eight.  Speed only, never correctness.
  // (Walking 2-8 consecutive x-planes per block for L1 reuse was measured neutral to slower: fewer blocks in flight.)
  const int bid = blockIdx.x, jj = bid >> 3;
  const int grp = (bid & 7) + 8 * (jj / L);
  if (grp >= CT * nb) return;
  const int c = grp % CT, b = grp / CT;
  // (given volumes with maps: the twiddles are made behind the plane's occupancy test -- half of dockE3's planes leave there)
  if (do_rotate || !occ) init_twiddles<N>(tw, tid, NT);
  // Slab orientation.  When the source z axis lies closer to the output x axis than to the output y axis,
  // the gather of an x-plane is perpendicular to the contiguous direction of memory (every lane its own
  // cache line).  The caller groups such rotations into launches with transposed = 1: they are processed
  // with the roles of x and y exchanged -- blocks are y-planes, the in-plane axis is x -- and the slab is
  // stored transposed ([kz][y][x]); K2 undoes it while staging (dlpd_k2.hip).
  const int tr_flag = transposed;
  const int lane = tid & 63, wave = tid >> 6;
  (void)lane; (void)wave;
  DLPD_STAMP_DECL;
  const int x = jj % L;
  bool pencil_live = true;                     // (this thread's pencil pair p = tid % NP, see the transform passes)
  {
  const float* v = vol + (size_t)b * vol_bstride + (size_t)c * L * L * L;
  float* Sf = reinterpret_cast<float*>(S);
  if (do_rotate) {
    const float* r = R + (size_t)b * 9;
    // transposed processing = the same code with the x and y columns of the sample matrix exchanged
    const float r0 = r[tr_flag ? 3 : 0], r1 = r[tr_flag ? 4 : 1], r2 = r[tr_flag ? 5 : 2];
    const float r3 = r[tr_flag ? 0 : 3], r4 = r[tr_flag ? 1 : 4], r5 = r[tr_flag ? 2 : 5];
    const float r6 = r[6], r7 = r[7], r8 = r[8];
    const float dx = x - c0;
    // the kernel spends 85-90 % of its time here (stamps): latency-bound, so keep several samples' loads
    // in flight per thread
#pragma unroll DLPD_K1_UNROLL
    for (int s = tid; s < L * L; s += NT) {
      // 64 consecutive samples form an 8 x 8 (y, z) tile, not a z row: under an oblique rotation a row of
      // 64 samples crosses up to ~80 source cache lines per gather instruction, a tile ~20 (the TCP serves
      // about one line per clock, and that is what bounds this kernel); axis-aligned rotations go from
      // 4 lines to 8 -- measured: K1 over the whole 6-degree set 1.9 ms -> see DESIGN.md
      constexpr int TZ = 8, TYY = 64 / TZ;                     // tile = TYY (y) x TZ (z) samples (other shapes measured equal)
      const int chunk = s >> 6, q = s & 63;
      const int y = (chunk / (L / TZ)) * TYY + q / TZ, z = (chunk % (L / TZ)) * TZ + q % TZ;
      const float dy = y - c0, dz = z - c0;
      const float px = c0 + (r0 * dx + r3 * dy + r6 * dz);
      const float py = c0 + (r1 * dx + r4 * dy + r7 * dz);
      const float pz = c0 + (r2 * dx + r5 * dy + r8 * dz);
      // ext < L: the volume is an ext^3 box in the corner of the L^3 one (Docker._dock_volumes_embedded): the rotated
      // volume is cropped to that box, as the reference crops it to its own
      Sf[((y >> 1) * RS + z) * 2 + (y & 1)] =
          (max(x, max(y, z)) >= ext) ? 0.f
          : quads ? trilinear_fetch_quads(quads + (size_t)c * L * (L - 1) * (L - 1), L, px, py, pz)
                  : trilinear_fetch(v, L, px, py, pz);
    }
  } else {
    // occ (given volumes only, untransposed): the batch entry's occupancy map, one byte per 4 x 4 x 4 cell (the maps of
    // dlpd_conv3d_split_sparse): voxels of empty cells are the zeros the map stands for and are NOT read (with unwritten
    // activations nobody wrote them); a plane whose cells are all empty goes out as zeros without a transform
    constexpr int NC = (L + 3) / 4;
    const unsigned char* ob = occ ? occ + ((size_t)b * NC + (x >> 2)) * NC * NC : nullptr;
    // the plane's cells in LDS (round 6, second form: the staging loop looked every voxel's cell up in global memory, a byte
    // load in front of every float load), and per y cell whether ANY of its z cells is occupied: with skip_empty the pencil
    // pairs of the other y cells are neither transformed nor written -- the consumer's pencil map (dlpd_pencil_bits of the
    // same occupancy map) does not mark them
    __shared__ unsigned char ocell[NC * NC];
    __shared__ int occ_any, yany[NC];
    if (ob) {
      if (tid == 0) occ_any = 0;
      if (tid < NC) yany[tid] = 0;
      __syncthreads();
      for (int i = tid; i < NC * NC; i += NT) {
        const unsigned char o = ob[i];
        ocell[i] = o;
        if (o) { occ_any = 1; yany[i / NC] = 1; }              // (plain stores of the same value by whoever finds one)
      }
      __syncthreads();
      if (!occ_any) {                                          // (block-uniform)
        if (skip_empty) return;                                // the consumer goes by the pencil map (dlpd_xy_correlate_packed_occ)
        cplx* a = A + ((size_t)b * CT_out + c_base + c) * NZ * L * L + (size_t)x * L;
        for (int s = tid; s < NP * NZ; s += NT)
          DLPD_STORE_STREAM(reinterpret_cast<float4*>(a + (size_t)(s / NP) * L * L + 2 * (s % NP)), make_float4(0.f, 0.f, 0.f, 0.f));
        return;
      }
      init_twiddles<N>(tw, tid, NT);
    }
    for (int s = tid; s < L * L; s += NT) {
      const int y = s / L, z = s % L;
      // transposed: this block is the plane y_orig = x, its in-plane index runs over x_orig
      const bool zero = ob && !ocell[(y >> 2) * NC + (z >> 2)];
      Sf[((y >> 1) * RS + z) * 2 + (y & 1)] = zero ? 0.f : (tr_flag ? v[((size_t)y * L + x) * L + z] : v[((size_t)x * L + y) * L + z]);
    }
    pencil_live = !(ob && skip_empty) || yany[(tid % NP) >> 1];
  }
  DLPD_STAMP(0);
  __syncthreads();
  DLPD_STAMP(1);
  const int p = tid % NP, t = tid / NP;
  {
    FftPass<N, R1, 1, -1, T, L> ps;
    if (pencil_live) ps.load(S + p * RS, 1, t, tw);
    __syncthreads();
    if (pencil_live) ps.store(S + p * RS, 1, t);
    __syncthreads();
  }
  {
    FftPass<N, R2, R1, -1, T> ps;
    if (pencil_live) ps.load(S + p * RS, 1, t, tw);
    __syncthreads();
    if (pencil_live) ps.store(S + p * RS, 1, t);
    __syncthreads();
  }
  DLPD_STAMP(2);
  // untangle the two real rows packed in each complex pencil; write [kz][x][y]
  cplx* a = A + ((size_t)b * CT_out + c_base + c) * NZ * L * L + (size_t)x * L;
  for (int s = tid; s < NP * NZ; s += NT) {
    const int m = s % NP, k = s / NP;
    if (!pencil_live) continue;                // (NT is a multiple of NP: m = tid % NP = p in every round)
    const cplx zk = S[m * RS + k];
    const cplx zn = S[m * RS + ((N - k) % N)];
    // even row: (Z[k] + conj(Z[N-k]))/2 ; odd row: (Z[k] - conj(Z[N-k]))/(2i)
    float4 o;
    o.x = 0.5f * (zk.x + zn.x);
    o.y = 0.5f * (zk.y - zn.y);
    o.z = 0.5f * (zk.y + zn.y);
    o.w = 0.5f * (zn.x - zk.x);
    DLPD_STORE_STREAM(reinterpret_cast<float4*>(a + (size_t)k * L * L + 2 * m), o);
  }
  DLPD_STAMP(3);
  }
  DLPD_STAMP_FLUSH(dlpd_stamps_k1, DLPD_STAMPS);
}

// ------------------------------------------------------------------------------------------
// K1 from a CHANNELS-LAST ligand: cl[x][y][z][Cp] (Cp = C rounded up to 16, zero padded).  The rotation is the same
// for every channel, so one gather address serves all of them: a lane fetches FOUR channels of one corner with one
// 16-byte load and the four lanes of a voxel a 64-byte run -- the number of cache-line requests per sample no longer
// depends on how oblique the rotation is (the per-channel kernel above issues one 8-byte request per lane and corner
// pair and spends 85-90 % of its time in the texture-cache pipeline; slab orientation and the quad layout exist to
// soften exactly that and are not needed here).  Same corner weights, products and summation order as
// trilinear_fetch: bit-identical samples.
//   block = (rotation b, plane x, YG rows, CC channels), YG x CC = 128: 64 two-row pencils (K1ClCfg); 1-D grid,
//   every XCD takes a contiguous range of (b, x) so that neighbouring planes -- which share source lines -- meet in
//   one L2.  LDS: 64 pencils x (N + 13) complex; the pencils of channel quad q start SKEW q elements into their
//   rows, which puts the 8-byte stores of a voxel's lanes (channel quads 4 YG/2 pencils apart: same bank otherwise)
//   on disjoint banks.
// ------------------------------------------------------------------------------------------
// OCC (round 6, its own instantiation: the dense callers run the code they always ran): `occ` (nb, ceil(L/4)^3) bytes,
// [x cell][y cell][z cell] per rotation, 0 = every sample of that 4 x 4 x 4 cell of the ROTATED volume is zero
// (dlpd_rotated_occupancy: a conservative map made from the stored ligand's own map).  A real ligand's representation is
// zero away from the protein, so most pencils of a rotated volume are empty: a task in an empty cell skips its sixteen
// loads (the sample is the +0.0 the sum of zero products is), a block whose cells are all empty skips the transform too
// and writes its zeros.  Exact; same spectra as without the map.
template <int N, bool OCC> __global__ void __launch_bounds__(K1ClCfg<N>::NP * FftPlan<N>::T)
k_rotate_zfft_cl(const float4* __restrict__ cl, const float* __restrict__ R, cplx* __restrict__ A,
                 int C, int Cq, int nb, float c0, int CT_out, int c_base, int ext, const unsigned char* __restrict__ occ,
                 int skip_empty) {
  constexpr int L = N / 2, NZ = N / 2 + 1, NP = K1ClCfg<N>::NP, CC = K1ClCfg<N>::CC, YG = K1ClCfg<N>::YG, NPR = YG / 2;
  constexpr int LPV = CC / 4, SKEW = 16 / LPV;         // lanes per voxel; bank skew (complex) between channel quads
  static_assert(CC * NPR == NP && L % YG == 0 && (NP * FftPlan<N>::T) % 64 == 0, "whole waves of pencils per block");
  constexpr int RS = N + DLPD_K1CL_PAD;
  constexpr int T = FftPlan<N>::T, R1 = FftPlan<N>::R1, R2 = FftPlan<N>::R2;
  constexpr int NT = NP * T;
  DLPD_DYN_SHARED(cplx, S);
  cplx* tw = S + NP * RS;
  const int tid = threadIdx.x;
  const int nchunk = Cq / (CC / 4), per = (L / YG) * nchunk;
  const int groups = nb * L, gper = (groups + 7) / 8;
  const int bid = blockIdx.x, seq = bid >> 3;
  const int g = (bid & 7) * gper + seq / per;
  if (seq / per >= gper || g >= groups) return;
  const int inner = seq % per, yg = inner / nchunk, chunk = inner % nchunk;
  const int b = g / L, x = g % L;
  constexpr int NC = (L + 3) / 4, YC = YG / 4;                 // cells per axis; y cells of a block's rows
  static_assert(YG % 4 == 0, "a block's rows are whole cells");
  __shared__ unsigned char occ_s[OCC ? YC * NC : 1];
  __shared__ int occ_any;
  if (OCC) {
    const unsigned char* ob = occ + (((size_t)b * NC + (x >> 2)) * NC + (yg * YG >> 2)) * NC;
    if (tid == 0) occ_any = 0;
    __syncthreads();
    if (tid < YC * NC) {
      const unsigned char o = ob[tid];
      occ_s[tid] = o;
      if (o) occ_any = 1;                                      // (plain store of the same value by whoever finds one)
    }
    __syncthreads();
    if (!occ_any) {                                            // (block-uniform) nothing to gather or transform: the zeros go out
      if (skip_empty) return;                                  // ... unless the consumer goes by the pencil map (dlpd_xy_correlate_packed_occ)
      for (int s = tid; s < NP * NZ; s += NT) {
        const int pm = s % NP, k = s / NP;
        const int c = chunk * CC + pm / NPR, m = pm % NPR;
        if (c < C)
          DLPD_STORE_STREAM(reinterpret_cast<float4*>(A + (((size_t)b * CT_out + c_base + c) * NZ + k) * L * L + (size_t)x * L +
                                                      yg * YG + 2 * m), make_float4(0.f, 0.f, 0.f, 0.f));
      }
      return;
    }
  }
  init_twiddles<N>(tw, tid, NT);
  {
    const K1ClRot rot = k1cl_load_rotation(R + (size_t)b * 9);
    for (int task = tid; task < NPR * L * LPV; task += NT) {
      const int q = task % LPV, z = (task / LPV) % L, m = (task / LPV) / L;
      const float4* src = cl + chunk * (CC / 4) + q;
      float4 acc[2];
      if (OCC && !occ_s[((2 * m) >> 2) * NC + (z >> 2)])       // rows 2m, 2m + 1 lie in one cell
        acc[0] = acc[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      else
        k1cl_sample_rows(src, Cq, L, ext, c0, rot, x, yg * YG + 2 * m, z, acc);
      // rows 2m (real part) and 2m+1 (imaginary part) of the four channels' pencils
      cplx* P = S + ((4 * q) * NPR + m) * RS + SKEW * q + z;
      P[0] = c_make(acc[0].x, acc[1].x);
      P[NPR * RS] = c_make(acc[0].y, acc[1].y);
      P[2 * NPR * RS] = c_make(acc[0].z, acc[1].z);
      P[3 * NPR * RS] = c_make(acc[0].w, acc[1].w);
    }
  }
  __syncthreads();
  const int p = (tid % NP) ^ DLPD_K1CL_PXOR, t = tid / NP;   // (PXOR != 0: diagnostic builds, which lane transforms which pencil)
  cplx* Sp = S + p * RS + SKEW * (p / (4 * NPR));
  {
    FftPass<N, R1, 1, -1, T, L> ps;
    ps.load(Sp, 1, t, tw);
    __syncthreads();
    ps.store(Sp, 1, t);
    __syncthreads();
  }
  {
    FftPass<N, R2, R1, -1, T> ps;
    ps.load(Sp, 1, t, tw);
    __syncthreads();
    ps.store(Sp, 1, t);
    __syncthreads();
  }
  // untangle the two real rows packed in each complex pencil; write [kz][x][y]
  for (int s = tid; s < NP * NZ; s += NT) {
    const int pm = s % NP, k = s / NP;
    const int c = chunk * CC + pm / NPR, m = pm % NPR;
    if (c < C) {
      const cplx* Z = S + pm * RS + SKEW * (pm / (4 * NPR));
      const cplx zk = Z[k];
      const cplx zn = Z[(N - k) % N];
      float4 o;
      o.x = 0.5f * (zk.x + zn.x);
      o.y = 0.5f * (zk.y - zn.y);
      o.z = 0.5f * (zk.y + zn.y);
      o.w = 0.5f * (zn.x - zk.x);
      cplx* a = A + (((size_t)b * CT_out + c_base + c) * NZ + k) * L * L + (size_t)x * L + yg * YG + 2 * m;
      DLPD_STORE_STREAM(reinterpret_cast<float4*>(a), o);
    }
  }
}

// (C, L, L, L) -> channels-last (L, L, L, Cp), Cp = 4 Cq >= C, zero padded
__global__ void __launch_bounds__(256) k_make_channels_last(const float* __restrict__ v, float4* __restrict__ cl, int C, int Cq, int L) {
  const size_t L3 = (size_t)L * L * L, total = L3 * Cq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % Cq);
    const size_t vox = i / Cq;
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = (4 * c4 + j < C) ? v[(size_t)(4 * c4 + j) * L3 + vox] : 0.f;
    cl[i] = make_float4(e[0], e[1], e[2], e[3]);
  }
}

template <int N> static int launch_k1_cl(const float4* cl, const float* R, cplx* A, int C, int nb, float c0, hipStream_t st,
                                         int CT_out, int c_base, int ext = 0, const unsigned char* occ = nullptr, int skip_empty = 0) {
  constexpr int L = N / 2, RS = N + DLPD_K1CL_PAD;
  const int Cq = ((C + DLPD_K1CL_CC - 1) / DLPD_K1CL_CC) * (DLPD_K1CL_CC / 4);
  const size_t shmem = (size_t)(K1ClCfg<N>::NP * RS + N) * sizeof(cplx);
  const int per = (L / K1ClCfg<N>::YG) * (Cq / (K1ClCfg<N>::CC / 4));
  const int gper = (nb * L + 7) / 8;
  dim3 grid((unsigned)(8 * gper * per)), block(K1ClCfg<N>::NP * FftPlan<N>::T);
  const int e = (ext > 0 && ext < L) ? ext : L;
  if (occ) {
    int rc = dlpd_set_max_dyn_shared((const void*)k_rotate_zfft_cl<N, true>, shmem);
    if (rc) return rc;
    DLPD_LAUNCH((k_rotate_zfft_cl<N, true>), grid, block, shmem, st, cl, R, A, C, Cq, nb, c0, CT_out, c_base, e, occ, skip_empty);
  } else {
    int rc = dlpd_set_max_dyn_shared((const void*)k_rotate_zfft_cl<N, false>, shmem);
    if (rc) return rc;
    DLPD_LAUNCH((k_rotate_zfft_cl<N, false>), grid, block, shmem, st, cl, R, A, C, Cq, nb, c0, CT_out, c_base, e, occ, 0);
  }
  return dlpd_check_launch();
}

// occ_out[b][cell] = 1 unless every sample of output cell `cell` of the volume rotated by R[b] is certainly zero: the samples
// of the cell's 4^3 voxels lie within +-e_a of the mapped cell centre on source axis a (e_a = 1.5 * the absolute row sum
// of the sample matrix) and read the corners floor(p), floor(p) + 1 -- if every source cell that range touches is empty in
// occ_src (one map for all channels of the stored ligand, dlpd_conv3d_tile_occupancy), all their products are zero.
// Same sample map as k1cl_sample_rows (p = c0 + M (voxel - c0), M columns r0..r2 | r3..r5 | r6..r8).
__global__ void __launch_bounds__(256) k_rotated_occupancy(const unsigned char* __restrict__ occ_src, const float* __restrict__ R,
                                                           unsigned char* __restrict__ occ_out, int nb, int L, float c0) {
  const int nc = (L + 3) / 4, nc3 = nc * nc * nc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb * nc3; i += gridDim.x * blockDim.x) {
    const int b = i / nc3, cell = i % nc3, cz = cell % nc, cy = (cell / nc) % nc, cx = cell / (nc * nc);
    const float* r = R + (size_t)b * 9;
    // centre of the cell's voxels that exist (the last cell of a box that is no multiple of 4 is smaller)
    const float hx = 0.5f * (min(4 * cx + 3, L - 1) - 4 * cx), hy = 0.5f * (min(4 * cy + 3, L - 1) - 4 * cy),
                hz = 0.5f * (min(4 * cz + 3, L - 1) - 4 * cz);
    const float dx = 4 * cx + hx - c0, dy = 4 * cy + hy - c0, dz = 4 * cz + hz - c0;
    bool any = false;
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float s = c0 + (r[a] * dx + r[3 + a] * dy + r[6 + a] * dz);
      const float e = fabsf(r[a]) * hx + fabsf(r[3 + a]) * hy + fabsf(r[6 + a]) * hz + 1e-3f * (1.f + fabsf(s));   // (+ rounding slack)
      lo[a] = max((int)floorf(s - e), 0) >> 2;
      hi[a] = min((int)floorf(s + e) + 1, L - 1) >> 2;
    }
    for (int sx = lo[0]; sx <= hi[0]; sx++)
      for (int sy = lo[1]; sy <= hi[1]; sy++)
        for (int sz = lo[2]; sz <= hi[2]; sz++) any |= occ_src[(sx * nc + sy) * nc + sz] != 0;
    occ_out[i] = any ? 1 : 0;
  }
}
// pencil map of nb cell maps: word [b][x cell] has bit (y cell) set where some z cell of that column is occupied.
// One wave per word: lane = y cell (nc <= 32) ORs its column's z cells, a ballot makes the word.
__global__ void __launch_bounds__(256) k_pencil_bits(const unsigned char* __restrict__ occ, unsigned* __restrict__ bits, int nb, int nc) {
  const int lane = threadIdx.x & 63, i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;      // word index (wave-uniform)
  bool any = false;
  if (i < nb * nc && lane < nc) {
    const unsigned char* o = occ + ((size_t)i * nc + lane) * nc;
    for (int cz = 0; cz < nc; cz++) any |= o[cz] != 0;
  }
  const unsigned long long m = __ballot(any);
  if (i < nb * nc && lane == 0) bits[i] = (unsigned)m;
}

// ------------------------------------------------------------------------------------------
// K3: z-axis C2R + (MODE 1) filter MLP + clash mask, or (MODE 0) plain real output.
//   grid (N/16, N [x'], nb), block 512 threads (8 waves), dynamic LDS: 64 pencils + twiddles + raw staging.
//   Bw   (nb, CT, NZ, N, N) complex [kz][x'][y']
//   MODE 0: out (nb, CT, N,N,N) real, optionally clamped to +-clip
//   MODE 1: V   (nb, N,N,N) = mask * (W2 . relu(W1 . clamp(corr) + b1) + b2)
//           score channels [0,C), clash channel C if has_clash (mask = corr_C < thr)
//   W1t  (C, HP) transposed + zero padded, b1 (HP), W2 (HP)
// ------------------------------------------------------------------------------------------
// One wave owns one channel of the current group: it streams that channel's raw spectra
// HBM -> LDS with LDS-DMA (for the NEXT group, overlapped with everything else), packs them into
// 8 two-row pencils, and runs the two wave-local FFT passes -- no block barrier involved.  Two
// block barriers per group separate "all pencils transformed" from the accumulation phase, in
// which every thread folds all channels of the group into the hidden units of its 4 voxels.
// threads per block and number of channel-owning waves (LDS: WC * (8 pencils + raw staging))
// TY: y rows of the tile.  16 = one channel per wave (8 two-row pencils); 8 = two channels per wave (4 pencils
// each).  N = 160 fused (MODE 1) takes 8-row tiles: 5 waves x 4 voxels per thread keep the 96 accumulators at
// 2 waves per SIMD (16-row tiles need either 8 voxels per thread or 10 waves, and spill either way).
// (Measured and rejected at N = 128: 8-row tiles with two blocks per CU, 2.82 vs 2.51 ms; a block that walks
// several tiles with the next tile's first DMA behind the current tile's last group, 2.51-2.57 vs 2.40 ms --
// the outer loop costs ~20 VGPRs of hoisted addressing next to the 96 accumulators.)
template <int N, int MODE> struct K3Cfg { static constexpr int NT = 512, WC = 8, TY = 16; };
// N = 80: five channel-owning waves, ten accumulating (the coarse grid of the real shapes: 0.78 -> 0.60 ms)
#ifndef DLPD_K3_80_NT
#define DLPD_K3_80_NT 640
#endif
// (N = 128 with 16 waves, 8 owning channels: the 128-VGPR ceiling spills the transform, 7.3 vs 2.57 ms)
template <int MODE> struct K3Cfg<80, MODE> { static constexpr int NT = (MODE == 0 ? 320 : DLPD_K3_80_NT), WC = 5, TY = 16; };
template <> struct K3Cfg<160, 0> { static constexpr int NT = 320, WC = 5, TY = 16; };
// N = 160, fused: five waves own the channels' transforms, TEN accumulate (2 voxels per thread): with the two-pass
// z transform the accumulation is the longer phase and the extra waves pay (2.90 -> 2.63 ms at the real shapes;
// with the three-pass transform they did not: 3.96 vs 3.71 ms)
#ifndef DLPD_K3_160_NT
#define DLPD_K3_160_NT 640
#endif
template <> struct K3Cfg<160, 1> { static constexpr int NT = DLPD_K3_160_NT, WC = 5, TY = 8; };
template <> struct K3Cfg<160, 2> { static constexpr int NT = DLPD_K3_160_NT, WC = 5, TY = 8; };
#ifndef DLPD_K3_LAUNDER_160
#define DLPD_K3_LAUNDER_160 1            // N = 160: pencil / pack offsets recomputed per group instead of hoisted (no spills)
#endif
#ifdef DLPD_STAMPS   // diagnostic build only (scripts/stamps.py): where a K3 wave spends its cycles
__device__ unsigned long long dlpd_stamps[16];
#endif
template <int N, int HP, int MODE> __global__ void __launch_bounds__((K3Cfg<N, MODE>::NT))
k_zifft_filter(const cplx* __restrict__ Bw, float* __restrict__ out, int CT, int C, int has_clash, int G,
               const float* __restrict__ W1t, const float* __restrict__ b1, const float* __restrict__ W2,
               float b2, int has_clip, float clip, float thr, K3Aux aux, K3Cand cd) {
  typedef K3Cfg<N, MODE> Cfg;
  constexpr int NZ = N / 2 + 1, RS = N + 8, TY = Cfg::TY, NPAIR = TY / 2;
  constexpr int NT = Cfg::NT, WC = Cfg::WC;
  constexpr int CPW = 8 / NPAIR;               // channels per wave: its 8 pencils = CPW channels x NPAIR row pairs
  constexpr int LPK = 64 / NPAIR;              // kz rows per 64-lane DMA instruction
  static_assert((NPAIR == 8 || NPAIR == 4) && WC * 64 <= NT, "one wave = 8 pencils x 8 threads");
  constexpr int EPT = (NPAIR * N) / NT > 0 ? (NPAIR * N) / NT : 1;   // complex outputs per thread per channel
  constexpr int MSTEP = NT / N;                // pair stride between a thread's outputs
  static_assert((NPAIR * N) % NT == 0 || NPAIR * N < NT, "tile/thread mismatch");
  constexpr int RAWC = ((NZ * NPAIR + 63) / 64) * 64;    // float4 slots per channel (whole waves)
  DLPD_DYN_SHARED(cplx, S);
  cplx* tw = S + WC * 8 * RS;
  float4* raw = reinterpret_cast<float4*>(tw + N);        // [WC][CPW][RAWC] staging of raw spectra
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = blockIdx.x * TY, xo = blockIdx.y, b = blockIdx.z;

  // hidden pre-activations of the thread's 2*EPT voxels (the SLP vectoriser pairs adjacent
  // hidden units into v_pk_fma_f32 with the weight pair in SGPRs and the voxel value broadcast)
  constexpr int HPH = HP;
  float nrm[EPT * 2];
  float total[EPT * 2];
#pragma unroll
  for (int e = 0; e < EPT * 2; e++) { nrm[e] = 0.f; total[e] = b2; }
  // candidate filter of the top-K stage (0: none valid yet -> the rotation is flagged for the full select)
  const unsigned cand_tau = (MODE == 1 && cd.keys) ? *cd.tau : 0u;
  if (MODE == 1 && cd.keys && !cand_tau && tid == 0) cd.count[cd.nb + blockIdx.z] = 1u;
  const int zz = tid % N, m0 = tid / N;        // output ownership
  const bool owner = (NPAIR * N >= NT) || (m0 < NPAIR);
  const int tr = lane & 7, qr = lane >> 3;     // FFT: lane = 8*pencil + thread
  float4* rawg = raw + wave * CPW * RAWC;

  // this wave's channels of group `cb`: raw[k][m] <- Bw[b][cb + wave*CPW + j][k][xo][y0+2m .. +1]
  // (lane = NPAIR*(k % LPK) + m: LPK runs of NPAIR*16 bytes per DMA instruction; k = N/2 is the last, short one)
  constexpr int NFULL = (N / 2) / LPK;         // full 64-lane DMA instructions per channel (bins 0..N/2-1)
  static_assert(NZ * NPAIR == NFULL * 64 + NPAIR, "raw channel = NFULL full DMA instructions + one short one");
  auto issue_channel = [&](int cb) {
#pragma unroll
    for (int j = 0; j < CPW; j++) {
      const int g = wave * CPW + j;
      if (g < G && cb + g < CT) {
        const cplx* src = Bw + (((size_t)b * CT + cb + g) * NZ * N + xo) * N + y0;
        const cplx* lane_src = src + (size_t)(lane / NPAIR) * N * N + 2 * (lane % NPAIR);
        float4* rj = rawg + j * RAWC;
#pragma unroll
        for (int it = 0; it < NFULL; it++) DLPD_GLDS16(lane_src + (size_t)it * LPK * N * N, rj + it * 64);
        const int mt = lane % NPAIR;                        // tail lanes re-read valid elements
        DLPD_GLDS16(src + (size_t)(N / 2) * N * N + 2 * mt, rj + NFULL * 64);
      }
    }
  };
  DLPD_STAMP_DECL;
  issue_channel(0);                            // first DMA in flight behind the twiddle table and the bias loads
  init_twiddles<N>(tw, tid, NT);
  constexpr int j0 = 0;
  float h[EPT * 2][HPH > 0 ? HPH : 1];
  if (MODE >= 1) {
    if (MODE == 1 && aux.C > 0 && aux.is_preact) {
      // rows 2m, 2m+1 and columns z, z^1 of the fine grid share one coarse voxel
      const int Na = aux.N;
      const size_t cstride = (size_t)Na * Na * Na;
      const float* ab = aux.p + (size_t)b * HP * cstride + ((size_t)(xo >> 1) * Na + (y0 >> 1)) * Na + (zz >> 1);
      if (owner) {
#pragma unroll
        for (int e = 0; e < EPT; e++)
#pragma unroll
          for (int j = 0; j < HP; j++) {
            const float v = ab[(size_t)j * cstride + (size_t)(m0 + e * MSTEP) * Na];
            h[2 * e][j] = v;
            h[2 * e + 1][j] = v;
          }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT * 2; e++)
#pragma unroll
        for (int j = 0; j < HPH; j++) h[e][j] = b1[j0 + j];
    }
  }
  __syncthreads();                             // twiddle table visible

  for (int cbase = 0; cbase < CT; cbase += G) {
    const int gn = (CT - cbase) < G ? (CT - cbase) : G;
    if (wave * CPW < gn) {
      DLPD_WAIT_VMEM();                        // this wave's own DMA has landed
      DLPD_WAVE_SYNC();
      DLPD_STAMP(0);
      // pack two rows per complex pencil: Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k])
      // Lane -> (pencil m, kz row) of the element it packs.  The DMA layout of `raw` is [it][kq][m] (slot =
      // NPAIR*kq + m inside the 64-slot block of DMA instruction `it`, k = LPK*it + kq).  Packing in that same
      // lane order makes a 16-lane store group hit 8 pencils x 2 k: the pencil stride (RS = N + 8 complex = 16
      // dwords mod 32) leaves only 8 distinct banks, a 4-way conflict on every ds_write_b64 (a third of this
      // kernel's LDS cycles, SQ_LDS_BANK_CONFLICT).  For NPAIR = 8 the lanes are therefore re-dealt so that a
      // store group covers 4 pencils x 4 consecutive-block k and every lane additionally walks the DMA
      // instructions rotated by 2m: found by exhaustive search over lane-bit permutations x rotations
      // (model of the ds_write_b64 / ds_read_b128 lane groups, MI355X_MICROARCH.md LDS table): raw reads stay
      // conflict-free, stores cost 5.0 instead of 16.25 LDS cycles.
#pragma unroll
      for (int j = 0; j < CPW; j++) {
        if (wave * CPW + j >= gn) break;
        int m, kq, lq = lane;
        if (N == 160 && DLPD_K3_LAUNDER_160) DLPD_OPAQUE(lq);
        if (NPAIR == 8) {
          m = (lq & 3) | (((lq >> 4) & 1) << 2);
          kq = ((lq >> 2) & 1) | (((lq >> 5) & 1) << 1) | (((lq >> 3) & 1) << 2);
        } else {
          m = lq % NPAIR;
          kq = lq / NPAIR;
        }
        const int rot = (NPAIR == 8) ? 2 * m : 0;            // lane-dependent start of its walk over the DMA instructions
        const int slot = NPAIR * kq + m;
        cplx* P = S + (wave * 8 + j * NPAIR + m) * RS;
        const float4* rj = rawg + j * RAWC;
        constexpr int PCH = NFULL > 8 ? NFULL / 2 : NFULL;   // raw elements in flight per lane
        const float4 qh = rj[NFULL * 64 + (lane % NPAIR)];  // k = N/2 (lanes with lane / NPAIR == 0 store it)
#pragma unroll
        for (int it0 = 0; it0 < NFULL; it0 += PCH) {
          float4 q[PCH];
          int kk[PCH];
#pragma unroll
          for (int u = 0; u < PCH; u++) {
            const int it = (it0 + u + rot) % NFULL;
            kk[u] = it * LPK + kq;
            q[u] = rj[it * 64 + slot];
          }
#pragma unroll
          for (int u = 0; u < PCH; u++) {
            const int k = kk[u];
            // k = 0: the purely real bin of both rows (both stores then write the same value to the same place)
            const cplx lo = (k == 0) ? c_make(q[u].x, q[u].z) : c_make(q[u].x - q[u].w, q[u].y + q[u].z);
            const cplx hi = (k == 0) ? c_make(q[u].x, q[u].z) : c_make(q[u].x + q[u].w, q[u].z - q[u].y);
            P[pencil_in_pos<N>(k)] = lo;
            P[pencil_in_pos<N>((N - k) & (k == 0 ? 0 : ~0))] = hi;
          }
        }
        if (lane / NPAIR == 0) S[(wave * 8 + j * NPAIR + lane % NPAIR) * RS + pencil_in_pos<N>(N / 2)] = c_make(qh.x, qh.z);
      }
      DLPD_WAIT_LDS();                         // raw fully read before it is refilled
      DLPD_WAVE_SYNC();
    }
    DLPD_STAMP(1);
    issue_channel(cbase + G);                  // next group's channel streams in behind the math
    DLPD_STAMP(2);
    if (wave * CPW < gn) {
      // N = 160 (three passes, 8-row tiles): the lane-dependent pencil offsets are made opaque per group, otherwise
      // ~60 of them are hoisted out of the group loop and spill next to the 96 accumulators
      int tq = tr, qq = qr;
      if (N == 160 && DLPD_K3_LAUNDER_160) { DLPD_OPAQUE(tq); DLPD_OPAQUE(qq); }
      fft_wave_pencils<N, +1>(S, wave * 8, RS, qq, tq, tw);
    }
    DLPD_STAMP(3);
    DLPD_LDS_BARRIER();                        // all channels of the group transformed
    DLPD_STAMP(4);
    if (owner) {
      if (MODE == 0) {
        for (int g = 0; g < gn; g++) {
          const int c = cbase + g;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const int m = m0 + e * MSTEP;
            const cplx val = S[(g * NPAIR + m) * RS + pencil_out_pos<N>(zz)];
            float v0 = val.x, v1 = val.y;
            if (has_clip && c < C) { v0 = DLPD_CLAMP(v0, clip); v1 = DLPD_CLAMP(v1, clip); }
            float* o = out + ((((size_t)b * CT + c) * N + xo) * N + y0 + 2 * m) * N + zz;
            o[0] = v0;
            o[N] = v1;
          }
        }
      } else {
        // score channels of this group; the clash channel (index C, always last) is peeled off
        const int gs = (cbase + gn <= C) ? gn : (C - cbase > 0 ? C - cbase : 0);
        // first-layer weights are wave-uniform (scalar loads): channel g+1's row is requested
        // before channel g's FMAs so the scalar-load latency hides behind them
        float wcur[HPH > 0 ? HPH : 1], wnxt[HPH > 0 ? HPH : 1];
        cplx vcur[EPT], vnxt[EPT];
        if (gs > 0) {
#pragma unroll
          for (int j = 0; j < HPH; j++) wcur[j] = W1t[(size_t)cbase * HP + j0 + j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vcur[e] = S[(m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
        }
        for (int g = 0; g < gs; g++) {
          // channel g+1's weights (scalar loads) and values (LDS) are requested here, one
          // iteration ahead: both share lgkmcnt, so the only wait sits at the top of the next
          // iteration, behind this channel's 96 FMAs
          const int gn1 = (g + 1 < gs ? g + 1 : g);
#pragma unroll
          for (int j = 0; j < HPH; j++) wnxt[j] = W1t[(size_t)(cbase + gn1) * HP + j0 + j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vnxt[e] = S[(gn1 * NPAIR + m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
          DLPD_SCHED_FENCE();
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            float v0 = vcur[e].x, v1 = vcur[e].y;
            if (has_clip) { v0 = DLPD_CLAMP(v0, clip); v1 = DLPD_CLAMP(v1, clip); }
#pragma unroll
            for (int j = 0; j < HPH; j++) {
              h[2 * e][j] = fmaf(wcur[j], v0, h[2 * e][j]);
              h[2 * e + 1][j] = fmaf(wcur[j], v1, h[2 * e + 1][j]);
            }
          }
          DLPD_SCHED_FENCE();
#pragma unroll
          for (int j = 0; j < HPH; j++) wcur[j] = wnxt[j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vcur[e] = vnxt[e];
        }
        if (has_clash && cbase + gn > C) {
          const int g = C - cbase;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const cplx v = S[(g * NPAIR + m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
            nrm[2 * e] = v.x;
            nrm[2 * e + 1] = v.y;
          }
        }
      }
    }
    DLPD_STAMP(5);
    DLPD_LDS_BARRIER();                        // pencils free for the next group
    DLPD_STAMP(6);
  }
  DLPD_STAMP_FLUSH(dlpd_stamps, DLPD_STAMPS);
  if (MODE == 1 && owner && aux.C > 0 && !aux.is_preact) {
    // coarse-resolution channels: rows 2m and 2m+1 and columns z, z^1 share one coarse voxel
    const int Na = aux.N;
    const float* ab = aux.p + (size_t)b * aux.C * Na * Na * Na + ((size_t)(xo >> 1) * Na + (y0 >> 1)) * Na + (zz >> 1);
    constexpr int CH = EPT > 2 ? 4 : 8;       // channels per chunk: EPT*CH loads in flight per thread
    const size_t cstride = (size_t)Na * Na * Na;
    for (int cb = 0; cb < aux.C; cb += CH) {
      float av[CH][EPT];
#pragma unroll
      for (int k = 0; k < CH; k++)
#pragma unroll
        for (int e = 0; e < EPT; e++)
          av[k][e] = (cb + k < aux.C) ? ab[(size_t)(cb + k) * cstride + (size_t)(m0 + e * MSTEP) * Na] : 0.f;
#pragma unroll
      for (int k = 0; k < CH; k++) {
        if (cb + k < aux.C) {
          const float* w = W1t + (size_t)(C + cb + k) * HP + j0;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const float v = av[k][e];
#pragma unroll
            for (int j = 0; j < HPH; j++) {
              h[2 * e][j] = fmaf(w[j], v, h[2 * e][j]);
              h[2 * e + 1][j] = fmaf(w[j], v, h[2 * e + 1][j]);
            }
          }
        }
      }
    }
  }
  if (MODE == 1 && owner) {
#pragma unroll
    for (int e = 0; e < EPT * 2; e++)
#pragma unroll
      for (int j = 0; j < HPH; j++) total[e] = fmaf(W2[j0 + j], fmaxf(h[e][j], 0.f), total[e]);
  }
  if (MODE == 2 && owner) {
    // first-layer pre-activations of these channels as HP planes (bias included): the coarse resolution's half of
    // SimpleFilter's first layer, which the fine grid's kernel picks up by index (DockingModels.py:74-83)
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int m = m0 + e * MSTEP;
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < HP; j++)
          out[((((size_t)b * HP + j) * N + xo) * N + y0 + 2 * m + u) * N + zz] = h[2 * e + u][j];
    }
  }
  if (MODE == 1 && owner) {
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int m = m0 + e * MSTEP;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        float acc = total[2 * e + u];
        if (has_clash) acc = acc * ((nrm[2 * e + u] < thr) ? 1.0f : 0.0f);
        out[(((size_t)b * N + xo) * N + y0 + 2 * m + u) * N + zz] = acc;
        if (cd.keys && cand_tau) k3_emit(cd, cand_tau, b, (unsigned)((xo * N + y0 + 2 * m + u) * N + zz), acc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3 walking several tiles per block (MODE 1 only).  With few channels per tile (the reference's real model:
// 17 channels = 2 groups at N = 160) a third of a one-tile block is the exposed latency of its first DMA
// (stamps: dma_wait 33 %); here a block takes `tpb` consecutive tiles (all y-tiles of one x') and the first group
// of the NEXT tile streams in behind the last group of the current one.  ONE loop over (tile, group) steps, not a
// tile loop around a group loop, and lane-dependent offsets made opaque per step: otherwise the compiler hoists
// another ~20 VGPRs of addressing next to the 96 accumulators and spills.  Not used at N <= 128, where 7 groups per
// tile hide that latency already and the extra register pressure costs more (measured 2.51-2.57 vs 2.40 ms).
// ------------------------------------------------------------------------------------------
template <int N, int HP, int MODE> __global__ void __launch_bounds__((K3Cfg<N, MODE>::NT))
k_zifft_filter_tiles(const cplx* __restrict__ Bw, float* __restrict__ out, int CT, int C, int has_clash, int G,
               const float* __restrict__ W1t, const float* __restrict__ b1, const float* __restrict__ W2,
               float b2, int has_clip, float clip, float thr, K3Aux aux, int ntiles, int tpb, K3Cand cd) {
  typedef K3Cfg<N, MODE> Cfg;
  constexpr int NZ = N / 2 + 1, RS = N + 8, TY = Cfg::TY, NPAIR = TY / 2;
  constexpr int NT = Cfg::NT, WC = Cfg::WC, NYT = N / TY;
  constexpr int CPW = 8 / NPAIR;               // channels per wave: its 8 pencils = CPW channels x NPAIR row pairs
  constexpr int LPK = 64 / NPAIR;              // kz rows per 64-lane DMA instruction
  static_assert((NPAIR == 8 || NPAIR == 4) && WC * 64 <= NT, "one wave = 8 pencils x 8 threads");
  constexpr int EPT = (NPAIR * N) / NT > 0 ? (NPAIR * N) / NT : 1;   // complex outputs per thread per channel
  constexpr int MSTEP = NT / N;                // pair stride between a thread's outputs
  static_assert((NPAIR * N) % NT == 0 || NPAIR * N < NT, "tile/thread mismatch");
  constexpr int RAWC = ((NZ * NPAIR + 63) / 64) * 64;    // float4 slots per channel (whole waves)
  DLPD_DYN_SHARED(cplx, S);
  cplx* tw = S + WC * 8 * RS;
  float4* raw = reinterpret_cast<float4*>(tw + N);        // [WC][CPW][RAWC] staging of raw spectra
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t_beg = blockIdx.x * tpb, t_end = (t_beg + tpb < ntiles) ? t_beg + tpb : ntiles;
  if (t_beg >= t_end) return;
  init_twiddles<N>(tw, tid, NT);

  // candidate filter of the top-K stage (0: none valid yet -> the rotation is flagged for the full select)
  const unsigned cand_tau = (MODE == 1 && cd.keys) ? *cd.tau : 0u;
  const int zz = tid % N, m0 = tid / N;        // output ownership
  const bool owner = (NPAIR * N >= NT) || (m0 < NPAIR);
  const int tr = lane & 7, qr = lane >> 3;     // FFT: lane = 8*pencil + thread
  float4* rawg = raw + wave * CPW * RAWC;

  // this wave's channels of group `cb` of tile `t`: raw[k][m] <- Bw[b][cb + wave*CPW + j][k][xo][y0+2m .. +1]
  // (lane = NPAIR*(k % LPK) + m: LPK runs of NPAIR*16 bytes per DMA instruction; k = N/2 is the last, short one)
  constexpr int NFULL = (N / 2) / LPK;         // full 64-lane DMA instructions per channel (bins 0..N/2-1)
  static_assert(NZ * NPAIR == NFULL * 64 + NPAIR, "raw channel = NFULL full DMA instructions + one short one");
  auto issue_channel = [&](int t, int cb) {
    const int ty0 = (t % NYT) * TY, txo = (t / NYT) % N, tb = t / (NYT * N);
#pragma unroll
    for (int j = 0; j < CPW; j++) {
      const int g = wave * CPW + j;
      if (g < G && cb + g < CT) {
        const cplx* src = Bw + (((size_t)tb * CT + cb + g) * NZ * N + txo) * N + ty0;
        const cplx* lane_src = src + (size_t)(lane / NPAIR) * N * N + 2 * (lane % NPAIR);
        float4* rj = rawg + j * RAWC;
#pragma unroll
        for (int it = 0; it < NFULL; it++) DLPD_GLDS16(lane_src + (size_t)it * LPK * N * N, rj + it * 64);
        const int mt = lane % NPAIR;                        // tail lanes re-read valid elements
        DLPD_GLDS16(src + (size_t)(N / 2) * N * N + 2 * mt, rj + NFULL * 64);
      }
    }
  };
  DLPD_STAMP_DECL;
  issue_channel(t_beg, 0);
  __syncthreads();                             // twiddle table visible

  // ONE loop over (tile, channel group) steps -- not a tile loop around a group loop: with a single loop level the
  // compiler keeps the same values in registers as the one-tile kernel did (an outer tile loop made it hoist
  // another ~20 VGPRs of addressing and spill next to the 96 accumulators)
  float nrm[EPT * 2];
  float h[EPT * 2][HP > 0 ? HP : 1];
  const int ngroups = (CT + G - 1) / G;
  int t = t_beg, cbase = 0;
#pragma unroll 1
  for (int step = 0, nsteps = (t_end - t_beg) * ngroups; step < nsteps; step++) {
  const int y0 = (t % NYT) * TY, xo = (t / NYT) % N, b = t / (NYT * N);
  if (cbase == 0) {
  if (cd.keys && !cand_tau && tid == 0) cd.count[cd.nb + b] = 1u;
  // hidden pre-activations of the thread's 2*EPT voxels (the SLP vectoriser pairs adjacent
  // hidden units into v_pk_fma_f32 with the weight pair in SGPRs and the voxel value broadcast)
#pragma unroll
  for (int e = 0; e < EPT * 2; e++) nrm[e] = 0.f;
  if (MODE == 1) {
    if (aux.C > 0 && aux.is_preact) {
      // rows 2m, 2m+1 and columns z, z^1 of the fine grid share one coarse voxel
      const int Na = aux.N;
      const size_t cstride = (size_t)Na * Na * Na;
      const float* ab = aux.p + (size_t)b * HP * cstride + ((size_t)(xo >> 1) * Na + (y0 >> 1)) * Na + (zz >> 1);
      if (owner) {
#pragma unroll
        for (int e = 0; e < EPT; e++)
#pragma unroll
          for (int j = 0; j < HP; j++) {
            const float v = ab[(size_t)j * cstride + (size_t)(m0 + e * MSTEP) * Na];
            h[2 * e][j] = v;
            h[2 * e + 1][j] = v;
          }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT * 2; e++)
#pragma unroll
        for (int j = 0; j < HP; j++) h[e][j] = b1[j];
    }
  }
  }

  {
    const int gn = (CT - cbase) < G ? (CT - cbase) : G;
    if (wave * CPW < gn) {
      DLPD_WAIT_VMEM();                        // this wave's own DMA has landed
      DLPD_WAVE_SYNC();
      DLPD_STAMP(0);
      // pack two rows per complex pencil: Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k])
      // Lane -> (pencil m, kz row) of the element it packs.  The DMA layout of `raw` is [it][kq][m] (slot =
      // NPAIR*kq + m inside the 64-slot block of DMA instruction `it`, k = LPK*it + kq).  Packing in that same
      // lane order makes a 16-lane store group hit 8 pencils x 2 k: the pencil stride (RS = N + 8 complex = 16
      // dwords mod 32) leaves only 8 distinct banks, a 4-way conflict on every ds_write_b64 (a third of this
      // kernel's LDS cycles, SQ_LDS_BANK_CONFLICT).  For NPAIR = 8 the lanes are therefore re-dealt so that a
      // store group covers 4 pencils x 4 consecutive-block k and every lane additionally walks the DMA
      // instructions rotated by 2m: found by exhaustive search over lane-bit permutations x rotations
      // (model of the ds_write_b64 / ds_read_b128 lane groups, MI355X_MICROARCH.md LDS table): raw reads stay
      // conflict-free, stores cost 5.0 instead of 16.25 LDS cycles.
#pragma unroll
      for (int j = 0; j < CPW; j++) {
        if (wave * CPW + j >= gn) break;
        int m, kq, lq = lane;
        if (1) DLPD_OPAQUE(lq);            // (pack offsets recomputed per group, not kept in VGPRs)
        if (NPAIR == 8) {
          m = (lq & 3) | (((lq >> 4) & 1) << 2);
          kq = ((lq >> 2) & 1) | (((lq >> 5) & 1) << 1) | (((lq >> 3) & 1) << 2);
        } else {
          m = lq % NPAIR;
          kq = lq / NPAIR;
        }
        const int rot = (NPAIR == 8) ? 2 * m : 0;            // lane-dependent start of its walk over the DMA instructions
        const int slot = NPAIR * kq + m;
        cplx* P = S + (wave * 8 + j * NPAIR + m) * RS;
        const float4* rj = rawg + j * RAWC;
        constexpr int PCH = NFULL > 8 ? NFULL / 2 : NFULL;   // raw elements in flight per lane
        const float4 qh = rj[NFULL * 64 + (lane % NPAIR)];  // k = N/2 (lanes with lane / NPAIR == 0 store it)
#pragma unroll
        for (int it0 = 0; it0 < NFULL; it0 += PCH) {
          float4 q[PCH];
#pragma unroll
          for (int u = 0; u < PCH; u++) q[u] = rj[((it0 + u + rot) % NFULL) * 64 + slot];
#pragma unroll
          for (int u = 0; u < PCH; u++) {
            const int k = ((it0 + u + rot) % NFULL) * LPK + kq;
            // k = 0: the purely real bin of both rows (both stores then write the same value to the same place)
            const cplx lo = (k == 0) ? c_make(q[u].x, q[u].z) : c_make(q[u].x - q[u].w, q[u].y + q[u].z);
            const cplx hi = (k == 0) ? c_make(q[u].x, q[u].z) : c_make(q[u].x + q[u].w, q[u].z - q[u].y);
            P[pencil_in_pos<N>(k)] = lo;
            P[pencil_in_pos<N>((N - k) & (k == 0 ? 0 : ~0))] = hi;
          }
        }
        if (lane / NPAIR == 0) S[(wave * 8 + j * NPAIR + lane % NPAIR) * RS + pencil_in_pos<N>(N / 2)] = c_make(qh.x, qh.z);
      }
      DLPD_WAIT_LDS();                         // raw fully read before it is refilled
      DLPD_WAVE_SYNC();
    }
    DLPD_STAMP(1);
    // the next DMA streams in behind the math: the next group of this tile, or the first group of the next tile
    if (cbase + G < CT) issue_channel(t, cbase + G);
    else if (t + 1 < t_end) issue_channel(t + 1, 0);
    DLPD_STAMP(2);
    if (wave * CPW < gn) {
      // lane-dependent offsets made opaque per group: otherwise the ~60 swizzled pencil offsets of the two
      // passes are hoisted out of the group / tile loops and live in VGPRs next to the 96 accumulators (spills)
      int tq = tr, qq = qr;
      if (1) { DLPD_OPAQUE(tq); DLPD_OPAQUE(qq); }
      fft_wave_pencils<N, +1>(S, wave * 8, RS, qq, tq, tw);
    }
    DLPD_STAMP(3);
    DLPD_LDS_BARRIER();                        // all channels of the group transformed
    DLPD_STAMP(4);
    if (owner) {
      if (MODE == 0) {
        for (int g = 0; g < gn; g++) {
          const int c = cbase + g;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const int m = m0 + e * MSTEP;
            const cplx val = S[(g * NPAIR + m) * RS + pencil_out_pos<N>(zz)];
            float v0 = val.x, v1 = val.y;
            if (has_clip && c < C) { v0 = DLPD_CLAMP(v0, clip); v1 = DLPD_CLAMP(v1, clip); }
            float* o = out + ((((size_t)b * CT + c) * N + xo) * N + y0 + 2 * m) * N + zz;
            o[0] = v0;
            o[N] = v1;
          }
        }
      } else {
        // score channels of this group; the clash channel (index C, always last) is peeled off
        const int gs = (cbase + gn <= C) ? gn : (C - cbase > 0 ? C - cbase : 0);
        // first-layer weights are wave-uniform (scalar loads): channel g+1's row is requested
        // before channel g's FMAs so the scalar-load latency hides behind them
        float wcur[HP > 0 ? HP : 1], wnxt[HP > 0 ? HP : 1];
        cplx vcur[EPT], vnxt[EPT];
        if (gs > 0) {
#pragma unroll
          for (int j = 0; j < HP; j++) wcur[j] = W1t[(size_t)cbase * HP + j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vcur[e] = S[(m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
        }
        for (int g = 0; g < gs; g++) {
          // channel g+1's weights (scalar loads) and values (LDS) are requested here, one
          // iteration ahead: both share lgkmcnt, so the only wait sits at the top of the next
          // iteration, behind this channel's 96 FMAs
          const int gn1 = (g + 1 < gs ? g + 1 : g);
#pragma unroll
          for (int j = 0; j < HP; j++) wnxt[j] = W1t[(size_t)(cbase + gn1) * HP + j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vnxt[e] = S[(gn1 * NPAIR + m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
          DLPD_SCHED_FENCE();
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            float v0 = vcur[e].x, v1 = vcur[e].y;
            if (has_clip) { v0 = DLPD_CLAMP(v0, clip); v1 = DLPD_CLAMP(v1, clip); }
#pragma unroll
            for (int j = 0; j < HP; j++) {
              h[2 * e][j] = fmaf(wcur[j], v0, h[2 * e][j]);
              h[2 * e + 1][j] = fmaf(wcur[j], v1, h[2 * e + 1][j]);
            }
          }
          DLPD_SCHED_FENCE();
#pragma unroll
          for (int j = 0; j < HP; j++) wcur[j] = wnxt[j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vcur[e] = vnxt[e];
        }
        if (has_clash && cbase + gn > C) {
          const int g = C - cbase;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const cplx v = S[(g * NPAIR + m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
            nrm[2 * e] = v.x;
            nrm[2 * e + 1] = v.y;
          }
        }
      }
    }
    DLPD_STAMP(5);
    DLPD_LDS_BARRIER();                        // pencils free for the next group
    DLPD_STAMP(6);
  }
  if (cbase + G < CT) { cbase += G; continue; }          // more channel groups of this tile
  if (MODE == 1 && owner && aux.C > 0 && !aux.is_preact) {
    // coarse-resolution channels: rows 2m and 2m+1 and columns z, z^1 share one coarse voxel
    const int Na = aux.N;
    const float* ab = aux.p + (size_t)b * aux.C * Na * Na * Na + ((size_t)(xo >> 1) * Na + (y0 >> 1)) * Na + (zz >> 1);
    constexpr int CH = EPT > 2 ? 4 : 8;       // channels per chunk: EPT*CH loads in flight per thread
    const size_t cstride = (size_t)Na * Na * Na;
    for (int cb = 0; cb < aux.C; cb += CH) {
      float av[CH][EPT];
#pragma unroll
      for (int k = 0; k < CH; k++)
#pragma unroll
        for (int e = 0; e < EPT; e++)
          av[k][e] = (cb + k < aux.C) ? ab[(size_t)(cb + k) * cstride + (size_t)(m0 + e * MSTEP) * Na] : 0.f;
#pragma unroll
      for (int k = 0; k < CH; k++) {
        if (cb + k < aux.C) {
          const float* w = W1t + (size_t)(C + cb + k) * HP;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const float v = av[k][e];
#pragma unroll
            for (int j = 0; j < HP; j++) {
              h[2 * e][j] = fmaf(w[j], v, h[2 * e][j]);
              h[2 * e + 1][j] = fmaf(w[j], v, h[2 * e + 1][j]);
            }
          }
        }
      }
    }
  }
  if (MODE == 1 && owner) {
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int m = m0 + e * MSTEP;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        float acc = b2;
#pragma unroll
        for (int j = 0; j < HP; j++) acc = fmaf(W2[j], fmaxf(h[2 * e + u][j], 0.f), acc);
        if (has_clash) acc = acc * ((nrm[2 * e + u] < thr) ? 1.0f : 0.0f);
        out[(((size_t)b * N + xo) * N + y0 + 2 * m + u) * N + zz] = acc;
        if (cd.keys && cand_tau) k3_emit(cd, cand_tau, b, (unsigned)((xo * N + y0 + 2 * m + u) * N + zz), acc);
      }
    }
  }
  cbase = 0;
  t++;
  }   // (tile, group) steps
  DLPD_STAMP_FLUSH(dlpd_stamps, DLPD_STAMPS);
}

// ------------------------------------------------------------------------------------------
// Generic filter over already materialised correlation volumes (multi-resolution path:
// DockingModels.py:74-83).  conv_k (nb, C_k, N_k^3); nearest upsample index = i / (N/N_k).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_filter_generic(const float* __restrict__ conv0, int C0, int N0, const float* __restrict__ conv1, int C1, int N1,
                 const float* __restrict__ mask_norm, float thr, int has_clash,
                 const float* __restrict__ W1t, const float* __restrict__ b1, const float* __restrict__ W2, float b2,
                 int H, float* __restrict__ V, int nb) {
  const int N = N0;
  const size_t N3 = (size_t)N * N * N;
  const size_t total = (size_t)nb * N3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / N3);
    const size_t r = i % N3;
    const int z = (int)(r % N), y = (int)((r / N) % N), x = (int)(r / ((size_t)N * N));
    float hid[DLPD_MAX_HIDDEN];
    for (int j = 0; j < H; j++) hid[j] = b1[j];
    for (int c = 0; c < C0; c++) {
      const float v = conv0[((size_t)b * C0 + c) * N3 + r];
      for (int j = 0; j < H; j++) hid[j] = fmaf(W1t[(size_t)c * H + j], v, hid[j]);
    }
    if (C1 > 0) {
      const int s = N / N1;
      const size_t r1 = ((size_t)(x / s) * N1 + (y / s)) * N1 + (z / s);
      const size_t N13 = (size_t)N1 * N1 * N1;
      for (int c = 0; c < C1; c++) {
        const float v = conv1[((size_t)b * C1 + c) * N13 + r1];
        for (int j = 0; j < H; j++) hid[j] = fmaf(W1t[(size_t)(C0 + c) * H + j], v, hid[j]);
      }
    }
    float acc = b2;
    for (int j = 0; j < H; j++) acc = fmaf(W2[j], fmaxf(hid[j], 0.f), acc);
    if (has_clash) acc = acc * ((mask_norm[i] < thr) ? 1.0f : 0.0f);
    V[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Vectorised per-voxel filter over real correlation volumes: 4 z-consecutive voxels per thread
// (float4 loads, 4 x HP accumulators in registers, first-layer weights by scalar loads).  Used
// where the MLP is not fused into the inverse FFT (N = 160, and GlobalDockingModel.forward).
//   conv0 (nb, *, N^3) with batch stride c0_bstride, first C0 channels used
//   conv1 (nb, C1, N1^3), N1 = N or N/2 (nearest upsample by index, DockingModels.py:74-76)
//   mask  clash correlation (batch stride mask_bstride), V = (mask < thr) * mlp
// ------------------------------------------------------------------------------------------
template <int HP> __global__ void __launch_bounds__(256)
k_filter_vec(const float* __restrict__ conv0, int C0, long long c0_bstride, int N, const float* __restrict__ conv1,
             int C1, int N1, int pre1, const float* __restrict__ mask, long long mask_bstride, float thr, int has_clash,
             const float* __restrict__ W1t, const float* __restrict__ b1, const float* __restrict__ W2, float b2,
             float* __restrict__ V, int nb) {
  const size_t N3 = (size_t)N * N * N, q3 = N3 / 4;
  const size_t total = (size_t)nb * q3;
  const size_t N13 = (size_t)N1 * N1 * N1;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / q3);
    const size_t r = (g % q3) * 4;
    float h[4][HP];
    if (pre1) {
      // conv1 holds the HP first-layer pre-activations of the coarse channels (bias included),
      // computed once per coarse voxel by k_filter_preact: 8x fewer FMAs than per fine voxel
      const int z = (int)(r % N), y = (int)((r / N) % N), x = (int)(r / ((size_t)N * N));
      const int s = N / N1;
      const float* p = conv1 + (size_t)b * HP * N13 + ((size_t)(x / s) * N1 + (y / s)) * N1;
#pragma unroll
      for (int j = 0; j < HP; j++) {
        if (s == 2) {
          const float2 u = *reinterpret_cast<const float2*>(p + (size_t)j * N13 + (z >> 1));
          h[0][j] = h[1][j] = u.x; h[2][j] = h[3][j] = u.y;
        } else {
          const float4 u = *reinterpret_cast<const float4*>(p + (size_t)j * N13 + z);
          h[0][j] = u.x; h[1][j] = u.y; h[2][j] = u.z; h[3][j] = u.w;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < HP; j++) h[k][j] = b1[j];
    }
    const float* c0 = conv0 + (size_t)b * c0_bstride + r;
    // channels in groups of CH: all loads of a group are issued before its multiply-adds (the
    // kernel is latency-bound otherwise: one 16-byte load in flight per thread)
    constexpr int CH = 8;
    for (int cb = 0; cb < C0; cb += CH) {
      float4 v[CH];
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (cb + u < C0) v[u] = DLPD_LOAD_STREAM(reinterpret_cast<const float4*>(c0 + (size_t)(cb + u) * N3));
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (cb + u < C0) {
          const float* w = W1t + (size_t)(cb + u) * HP;
#pragma unroll
          for (int j = 0; j < HP; j++) {
            const float wj = w[j];
            h[0][j] = fmaf(wj, v[u].x, h[0][j]);
            h[1][j] = fmaf(wj, v[u].y, h[1][j]);
            h[2][j] = fmaf(wj, v[u].z, h[2][j]);
            h[3][j] = fmaf(wj, v[u].w, h[3][j]);
          }
        }
    }
    if (C1 > 0 && !pre1) {
      const int z = (int)(r % N), y = (int)((r / N) % N), x = (int)(r / ((size_t)N * N));
      const int s = N / N1;
      const float* c1 = conv1 + (size_t)b * C1 * N13 + ((size_t)(x / s) * N1 + (y / s)) * N1;
      for (int c = 0; c < C1; c++) {
        float v0, v1, v2, v3;
        const float* p = c1 + (size_t)c * N13;
        if (s == 2) {
          const float2 u = *reinterpret_cast<const float2*>(p + (z >> 1));
          v0 = v1 = u.x; v2 = v3 = u.y;
        } else if (s == 1) {
          const float4 u = *reinterpret_cast<const float4*>(p + z);
          v0 = u.x; v1 = u.y; v2 = u.z; v3 = u.w;
        } else {
          v0 = p[z / s]; v1 = p[(z + 1) / s]; v2 = p[(z + 2) / s]; v3 = p[(z + 3) / s];
        }
        const float* w = W1t + (size_t)(C0 + c) * HP;
#pragma unroll
        for (int j = 0; j < HP; j++) {
          const float wj = w[j];
          h[0][j] = fmaf(wj, v0, h[0][j]);
          h[1][j] = fmaf(wj, v1, h[1][j]);
          h[2][j] = fmaf(wj, v2, h[2][j]);
          h[3][j] = fmaf(wj, v3, h[3][j]);
        }
      }
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float acc = b2;
#pragma unroll
      for (int j = 0; j < HP; j++) acc = fmaf(W2[j], fmaxf(h[k][j], 0.f), acc);
      o[k] = acc;
    }
    if (has_clash) {
      const float4 nm = *reinterpret_cast<const float4*>(mask + (size_t)b * mask_bstride + r);
      o[0] *= (nm.x < thr) ? 1.0f : 0.0f;
      o[1] *= (nm.y < thr) ? 1.0f : 0.0f;
      o[2] *= (nm.z < thr) ? 1.0f : 0.0f;
      o[3] *= (nm.w < thr) ? 1.0f : 0.0f;
    }
    *reinterpret_cast<float4*>(V + (size_t)b * N3 + r) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// pre (nb, HP, n) = b1 + W1rows^T conv1 (nb, C1, n): the coarse half of the first layer on the coarse grid
template <int HP> __global__ void __launch_bounds__(256)
k_filter_preact(const float* __restrict__ conv1, int C1, size_t n, const float* __restrict__ W1rows,
                const float* __restrict__ b1, float* __restrict__ pre, int nb) {
  const size_t q = n / 4, total = (size_t)nb * q;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / q);
    const size_t r = (g % q) * 4;
    float h[4][HP];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < HP; j++) h[k][j] = b1[j];
    const float* c1 = conv1 + (size_t)b * C1 * n + r;
    for (int c = 0; c < C1; c++) {
      const float4 v = *reinterpret_cast<const float4*>(c1 + (size_t)c * n);
      const float* w = W1rows + (size_t)c * HP;
#pragma unroll
      for (int j = 0; j < HP; j++) {
        const float wj = w[j];
        h[0][j] = fmaf(wj, v.x, h[0][j]);
        h[1][j] = fmaf(wj, v.y, h[1][j]);
        h[2][j] = fmaf(wj, v.z, h[2][j]);
        h[3][j] = fmaf(wj, v.w, h[3][j]);
      }
    }
#pragma unroll
    for (int j = 0; j < HP; j++)
      *reinterpret_cast<float4*>(pre + ((size_t)b * HP + j) * n + r) = make_float4(h[0][j], h[1][j], h[2][j], h[3][j]);
  }
}

template <int HP> static int launch_filter_vec(const float* conv0, int C0, long long c0bs, int N, const float* conv1,
                                               int C1, int N1, int pre1, const float* mask, long long mbs, float thr,
                                               int has_clash, const float* W1t, const float* b1, const float* W2,
                                               float b2, float* V, int nb, hipStream_t st) {
  const size_t total = (size_t)nb * N * N * N / 4;
  size_t nblk = (total + 255) / 256;
  if (nblk > 65536) nblk = 65536;
  DLPD_LAUNCH((k_filter_vec<HP>), dim3((unsigned)nblk), dim3(256), 0, st, conv0, C0, c0bs, N, conv1, C1, N1, pre1, mask, mbs,
              thr, has_clash, W1t, b1, W2, b2, V, nb);
  return dlpd_check_launch();
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
template <int N> static int launch_k1(const float* vol, const float* R, cplx* A, int CT, int nb, long long vbs,
                                      int do_rotate, float c0, hipStream_t st, int CT_out = 0, int c_base = 0,
                                      int transposed = 0, const float4* quads = nullptr, int ext = 0,
                                      const unsigned char* occ = nullptr, int skip_empty = 0) {
  constexpr int L = N / 2;
  const int groups = ((CT * nb + 7) / 8) * 8;
  dim3 grid(groups * L), block((N / 4) * FftPlan<N>::T);
  DLPD_LAUNCH((k_rotate_zfft<N>), grid, block, 0, st, vol, R, A, CT, nb, vbs, do_rotate, c0,
              CT_out > 0 ? CT_out : CT, c_base, transposed ? 1 : 0, quads, (ext > 0 && ext < L) ? ext : L, occ, skip_empty);
  return dlpd_check_launch();
}

// K3 in its role-split formulation (dedicated transform / filter waves) lives in dlpd_k3r.hip
int dlpd_k3r_supported(int L, int HP, int mode);
int dlpd_k3r_filter(const cplx* Bw, float* V, int CT, int C, int has_clash, int nb, int L, const float* W1t, int HP,
                    const float* b1, const float* W2, float b2, int has_clip, float clip, float thr, K3Aux aux, K3Cand cd,
                    hipStream_t st);
int dlpd_k3r_preact(const cplx* Bw, float* pre, int C, int nb, int L, const float* W1rows, int HP, const float* b1,
                    int has_clip, float clip, hipStream_t st);
// which formulation the un-suffixed entry points take where both exist (variant builds: -DDLPD_K3_DEFAULT_FORM=1)
#ifndef DLPD_K3_DEFAULT_FORM
#define DLPD_K3_DEFAULT_FORM 2               // 1: channel-owning waves (this file), 2: role-split waves (dlpd_k3r.hip)
#endif

// K2 lives in dlpd_k2.hip (its own translation unit: it is built with -fno-slp-vectorize)
int dlpd_k2_forward(const cplx* A, cplx* out, int CT, int nb, int L, float scale, hipStream_t st);
int dlpd_k2_correlate(const cplx* A, const cplx* rec, cplx* out, int CT, int nb, int L, long long rbs, hipStream_t st,
                      int transposed = 0);
int dlpd_k2_orientation_supported(int L);
long long dlpd_k2_packed_receptor_floats(int CT, int L);
int dlpd_k2_pack_receptor(const cplx* rec, void* packed, int CT, int L, hipStream_t st);
int dlpd_k2_correlate_packed(const cplx* A, const cplx* packed, cplx* out, int CT, int nb, int L, hipStream_t st,
                             const unsigned char* pmap = nullptr, int nmasked = 0);

// channels per group.  One channel per wave (16-row tiles, N <= 128): as many as there are channel-owning waves -- 49
// channels on 8 waves are six full groups and one with the clash channel alone, 1 % faster than seven groups of seven,
// which leave a wave idle in every transform phase.  Two channels per wave (8-row tiles, N = 160): balanced groups
// (17 channels on 10 slots: 9 + 8 is 3 % faster than 10 + 7).
static int k3_group(int CT, int maxg, bool balanced) {
  if (!balanced) return CT < maxg ? CT : maxg;
  const int ng = (CT + maxg - 1) / maxg;
  return (CT + ng - 1) / ng;
}

template <int N, int HP, int MODE> static int launch_k3(const cplx* Bw, float* out, int CT, int C, int has_clash,
                                                        int nb, const float* W1t, const float* b1, const float* W2,
                                                        float b2, int has_clip, float clip, float thr,
                                                        hipStream_t st, K3Aux aux = K3Aux{nullptr, 0, 0, 0},
                                                        K3Cand cd = K3Cand{nullptr, nullptr, nullptr, 0, 0}) {
  typedef K3Cfg<N, MODE> Cfg;
  constexpr int RS = N + 8, NZ = N / 2 + 1, W = Cfg::WC, NPAIR = Cfg::TY / 2, CPW = 8 / NPAIR;
  constexpr int RAWC = ((NZ * NPAIR + 63) / 64) * 64;
  const size_t shmem = (size_t)(W * 8 * RS + N) * sizeof(cplx) + (size_t)W * CPW * RAWC * 16;
  int rc = dlpd_set_max_dyn_shared((const void*)k_zifft_filter<N, HP, MODE>, shmem);
  if (rc) return rc;
  const int G = k3_group(CT, W * CPW, CPW > 1);   // channels per group (CPW per wave), <= W * CPW
  dim3 grid(N / Cfg::TY, N, nb), block(Cfg::NT);
  DLPD_LAUNCH((k_zifft_filter<N, HP, MODE>), grid, block, shmem, st, Bw, out, CT, C, has_clash, G, W1t, b1, W2, b2,
              has_clip, clip, thr, aux, cd);
  return dlpd_check_launch();
}

// fused K3 over several tiles per block (see k_zifft_filter_tiles)
template <int N, int HP> static int launch_k3_tiles(const cplx* Bw, float* out, int CT, int C, int has_clash, int nb,
                                                    const float* W1t, const float* b1, const float* W2, float b2,
                                                    int has_clip, float clip, float thr, hipStream_t st, K3Aux aux,
                                                    K3Cand cd) {
  typedef K3Cfg<N, 1> Cfg;
  constexpr int RS = N + 8, NZ = N / 2 + 1, W = Cfg::WC, NPAIR = Cfg::TY / 2, CPW = 8 / NPAIR;
  constexpr int RAWC = ((NZ * NPAIR + 63) / 64) * 64;
  const size_t shmem = (size_t)(W * 8 * RS + N) * sizeof(cplx) + (size_t)W * CPW * RAWC * 16;
  int rc = dlpd_set_max_dyn_shared((const void*)k_zifft_filter_tiles<N, HP, 1>, shmem);
  if (rc) return rc;
  const int G = k3_group(CT, W * CPW, CPW > 1);
  const int ntiles = (N / Cfg::TY) * N * nb, tpb = N / Cfg::TY;       // one x' plane per block
  DLPD_LAUNCH((k_zifft_filter_tiles<N, HP, 1>), dim3((ntiles + tpb - 1) / tpb), dim3(Cfg::NT), shmem, st, Bw, out, CT, C,
              has_clash, G, W1t, b1, W2, b2, has_clip, clip, thr, aux, ntiles, tpb, cd);
  return dlpd_check_launch();
}

template <int N> static int k3_filter_dispatch(int HP, const cplx* Bw, float* V, int CT, int C, int has_clash, int nb,
                                               const float* W1t, const float* b1, const float* W2, float b2,
                                               int has_clip, float clip, float thr, hipStream_t st,
                                               K3Aux aux = K3Aux{nullptr, 0, 0, 0},
                                               K3Cand cd = K3Cand{nullptr, nullptr, nullptr, 0, 0}) {
  if constexpr (N == 160) {                    // few groups per tile: the tile-walking kernel
    switch (HP) {
      case 2: return launch_k3_tiles<N, 2>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
      case 4: return launch_k3_tiles<N, 4>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
      case 8: return launch_k3_tiles<N, 8>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
      case 16: return launch_k3_tiles<N, 16>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
      case 24: return launch_k3_tiles<N, 24>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
      case 32: return launch_k3_tiles<N, 32>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
      default: return DLPD_ERR_UNSUPPORTED;
    }
  }
  switch (HP) {
    case 2: return launch_k3<N, 2, 1>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 4: return launch_k3<N, 4, 1>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 8: return launch_k3<N, 8, 1>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 16: return launch_k3<N, 16, 1>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 24: return launch_k3<N, 24, 1>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 32: return launch_k3<N, 32, 1>(Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

extern "C" {

#ifdef DLPD_STAMPS
int dlpd_debug_read_stamps(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(dlpd_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

int dlpd_hidden_pad(int H) {
  const int opts[6] = {2, 4, 8, 16, 24, 32};
  for (int i = 0; i < 6; i++)
    if (H <= opts[i]) return opts[i];
  return -1;
}

int dlpd_grid_supported(int L) { return (L == 32 || L == 40 || L == 64 || L == 80) ? 1 : 0; }

// hidden width the FUSED pipeline pads H to on a fine grid of L^3 voxels per volume (two_res: plus a coarse grid of
// (L/2)^3, the reference's layout): dlpd_hidden_pad's widths, and 48 -- the reference class default's hidden width,
// two voxels per thread in the role-split K3 -- where that kernel exists; -1: no fused kernel (the ops path takes over)
int dlpd_fused_hidden_pad(int H, int L, int two_res) {
  const int hp = dlpd_hidden_pad(H);
  if (hp > 0) return hp;
  if (H <= 48 && dlpd_k3r_supported(L, 48, 1) && (!two_res || dlpd_k3r_supported(L / 2, 48, 2))) return 48;
  return -1;
}

int dlpd_rotate_trilinear(const float* vol, const float* R, float* out, int B, int C, int L, long long vol_bstride,
                          float center, void* stream) {
  if (!vol || !R || !out || B <= 0 || C <= 0 || L <= 0) return DLPD_ERR_ARG;
  const size_t total = (size_t)B * C * L * L * L;
  size_t nblk = (total + 255) / 256;
  if (nblk > 8192) nblk = 8192;
  DLPD_LAUNCH(k_rotate, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, vol, R, out, B, C, L, vol_bstride,
              center);
  return dlpd_check_launch();
}

// vol (nb*CT volumes as (nb, CT, L^3), or one (CT, L^3) set with vol_bstride = 0) -> channels
// [c_base, c_base + CT) of wsA (nb, CT_out, NZ, L, L)
int dlpd_zfft_oriented_ext(const float* vol, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                           long long vol_bstride, int do_rotate, float center, int transposed, int extent, void* stream) {
  if (!vol || !wsA || nb <= 0 || CT <= 0 || c_base < 0 || c_base + CT > CT_out || extent < 0 || extent > L) return DLPD_ERR_ARG;
  if (do_rotate && !R) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  cplx* A = (cplx*)wsA;
  switch (L) {
    case 32: return launch_k1<64>(vol, R, A, CT, nb, vol_bstride, do_rotate, center, st, CT_out, c_base, transposed, nullptr, extent);
    case 40: return launch_k1<80>(vol, R, A, CT, nb, vol_bstride, do_rotate, center, st, CT_out, c_base, transposed, nullptr, extent);
    case 64: return launch_k1<128>(vol, R, A, CT, nb, vol_bstride, do_rotate, center, st, CT_out, c_base, transposed, nullptr, extent);
    case 80: return launch_k1<160>(vol, R, A, CT, nb, vol_bstride, do_rotate, center, st, CT_out, c_base, transposed, nullptr, extent);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// given volumes (no rotation) with their occupancy maps: occ (nb, ceil(L/4)^3) bytes, one map per batch entry (all CT channels)
int dlpd_zfft_volumes_occ(const float* vol, const unsigned char* occ, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                          long long vol_bstride, int skip_empty, void* stream) {
  if (!vol || !occ || !wsA || nb <= 0 || CT <= 0 || c_base < 0 || c_base + CT > CT_out) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  cplx* A = (cplx*)wsA;
  switch (L) {
    case 32: return launch_k1<64>(vol, nullptr, A, CT, nb, vol_bstride, 0, 0.f, st, CT_out, c_base, 0, nullptr, 0, occ, skip_empty);
    case 40: return launch_k1<80>(vol, nullptr, A, CT, nb, vol_bstride, 0, 0.f, st, CT_out, c_base, 0, nullptr, 0, occ, skip_empty);
    case 64: return launch_k1<128>(vol, nullptr, A, CT, nb, vol_bstride, 0, 0.f, st, CT_out, c_base, 0, nullptr, 0, occ, skip_empty);
    case 80: return launch_k1<160>(vol, nullptr, A, CT, nb, vol_bstride, 0, 0.f, st, CT_out, c_base, 0, nullptr, 0, occ, skip_empty);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

int dlpd_zfft_oriented(const float* vol, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                       long long vol_bstride, int do_rotate, float center, int transposed, void* stream) {
  return dlpd_zfft_oriented_ext(vol, R, wsA, nb, CT, CT_out, c_base, L, vol_bstride, do_rotate, center, transposed, 0, stream);
}

size_t dlpd_quads_floats(int nvol, int L) { return (size_t)nvol * L * (L - 1) * (L - 1) * 4; }

int dlpd_make_quads(const float* vol, float* quads, int nvol, int L, void* stream) {
  if (!vol || !quads || nvol <= 0 || L < 2) return DLPD_ERR_ARG;
  const size_t total = (size_t)nvol * L * (L - 1) * (L - 1);
  size_t nblk = (total + 255) / 256;
  if (nblk > 65536) nblk = 65536;
  DLPD_LAUNCH(k_make_quads, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, vol, (float4*)quads, nvol, L);
  return dlpd_check_launch();
}

// rotation + z FFT of ONE volume set shared by all rotations, gathered from its quad layout
int dlpd_zfft_quads(const float* quads, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                    float center, int transposed, void* stream) {
  if (!quads || !R || !wsA || nb <= 0 || CT <= 0 || c_base < 0 || c_base + CT > CT_out) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  cplx* A = (cplx*)wsA;
  const float* dummy = quads;                          // the plain-volume pointer is unused on this path
  const float4* q4 = (const float4*)quads;
  switch (L) {
    case 32: return launch_k1<64>(dummy, R, A, CT, nb, 0, 1, center, st, CT_out, c_base, transposed, q4);
    case 40: return launch_k1<80>(dummy, R, A, CT, nb, 0, 1, center, st, CT_out, c_base, transposed, q4);
    case 64: return launch_k1<128>(dummy, R, A, CT, nb, 0, 1, center, st, CT_out, c_base, transposed, q4);
    case 80: return launch_k1<160>(dummy, R, A, CT, nb, 0, 1, center, st, CT_out, c_base, transposed, q4);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

size_t dlpd_channels_last_floats(int C, int L) {
  return (size_t)L * L * L * (size_t)(((C + DLPD_K1CL_CC - 1) / DLPD_K1CL_CC) * DLPD_K1CL_CC);
}

int dlpd_make_channels_last(const float* vol, float* cl, int C, int L, void* stream) {
  if (!vol || !cl || C <= 0 || L <= 0) return DLPD_ERR_ARG;
  const int Cq = ((C + DLPD_K1CL_CC - 1) / DLPD_K1CL_CC) * (DLPD_K1CL_CC / 4);
  const size_t total = (size_t)L * L * L * Cq;
  size_t nblk = (total + 255) / 256;
  if (nblk > 65536) nblk = 65536;
  DLPD_LAUNCH(k_make_channels_last, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, vol, (float4*)cl, C, Cq, L);
  return dlpd_check_launch();
}

// rotation + z FFT of the C score channels of ONE ligand shared by all rotations, gathered from its channels-last copy
#ifndef DLPD_K1_DEFAULT_FORM
#define DLPD_K1_DEFAULT_FORM 1               // 1: every wave gathers, transforms and stores (this file), 2: role-split (dlpd_k1r.hip)
#endif

int dlpd_zfft_channels_last_form(const float* cl, const float* R, void* wsA, int nb, int C, int CT_out, int c_base, int L,
                                 float center, int extent, int form, void* stream) {
  if (!cl || !R || !wsA || nb <= 0 || C <= 0 || c_base < 0 || c_base + C > CT_out || extent < 0 || extent > L) return DLPD_ERR_ARG;
  if (form < 0 || form > 2) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  cplx* A = (cplx*)wsA;
  const float4* c4 = (const float4*)cl;
  if (form == 0) form = (L == 64 || L == 80) ? DLPD_K1_DEFAULT_FORM : 1;
  if (form == 2) return dlpd_k1_role_split(c4, R, A, C, nb, center, st, CT_out, c_base, extent, L);
  switch (L) {
    case 32: return launch_k1_cl<64>(c4, R, A, C, nb, center, st, CT_out, c_base, extent);
    case 40: return launch_k1_cl<80>(c4, R, A, C, nb, center, st, CT_out, c_base, extent);
    case 64: return launch_k1_cl<128>(c4, R, A, C, nb, center, st, CT_out, c_base, extent);
    case 80: return launch_k1_cl<160>(c4, R, A, C, nb, center, st, CT_out, c_base, extent);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// occ_src (ceil(L/4)^3 bytes: the stored ligand's cells, all channels) -> occ_out (nb maps): the cells of each ROTATED volume
// that can hold a non-zero sample (conservative); R as K1 takes it
int dlpd_rotated_occupancy(const unsigned char* occ_src, const float* R, unsigned char* occ_out, unsigned* pencil_out, int nb,
                           int L, float center, void* stream) {
  if (!occ_src || !R || !occ_out || nb <= 0 || L <= 0 || L > 128) return DLPD_ERR_ARG;
  const int nc = (L + 3) / 4, total = nb * nc * nc * nc;
  DLPD_LAUNCH(k_rotated_occupancy, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, occ_src, R, occ_out,
              nb, L, center);
  if (pencil_out)
    DLPD_LAUNCH(k_pencil_bits, dim3((unsigned)((nb * nc + 3) / 4)), dim3(256), 0, (hipStream_t)stream, occ_out, pencil_out, nb, nc);
  return dlpd_check_launch();
}
// the pencil words of nb GIVEN cell maps (the volumes path: the plugin's own maps)
int dlpd_pencil_bits(const unsigned char* occ, unsigned* pencil_out, int nb, int L, void* stream) {
  if (!occ || !pencil_out || nb <= 0 || L <= 0 || L > 128) return DLPD_ERR_ARG;
  const int nc = (L + 3) / 4;
  DLPD_LAUNCH(k_pencil_bits, dim3((unsigned)((nb * nc + 3) / 4)), dim3(256), 0, (hipStream_t)stream, occ, pencil_out, nb, nc);
  return dlpd_check_launch();
}
// 1 where dlpd_xy_correlate_packed_occ exists: the packed-receptor boxes (80, 40)
int dlpd_pencil_map_supported(int L) { return dlpd_k2_packed_receptor_floats(1, L) > 0 ? 1 : 0; }

// dlpd_zfft_channels_last_ext with the rotated volumes' occupancy maps (dlpd_rotated_occupancy): empty cells are not gathered
int dlpd_zfft_channels_last_occ(const float* cl, const float* R, const unsigned char* occ, void* wsA, int nb, int C, int CT_out,
                                int c_base, int L, float center, int extent, int skip_empty, void* stream) {
  if (!cl || !R || !occ || !wsA || nb <= 0 || C <= 0 || c_base < 0 || c_base + C > CT_out || extent < 0 || extent > L) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  cplx* A = (cplx*)wsA;
  const float4* c4 = (const float4*)cl;
  switch (L) {
    case 32: return launch_k1_cl<64>(c4, R, A, C, nb, center, st, CT_out, c_base, extent, occ, skip_empty);
    case 40: return launch_k1_cl<80>(c4, R, A, C, nb, center, st, CT_out, c_base, extent, occ, skip_empty);
    case 64: return launch_k1_cl<128>(c4, R, A, C, nb, center, st, CT_out, c_base, extent, occ, skip_empty);
    case 80: return launch_k1_cl<160>(c4, R, A, C, nb, center, st, CT_out, c_base, extent, occ, skip_empty);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

int dlpd_zfft_channels_last_ext(const float* cl, const float* R, void* wsA, int nb, int C, int CT_out, int c_base, int L,
                                float center, int extent, void* stream) {
  return dlpd_zfft_channels_last_form(cl, R, wsA, nb, C, CT_out, c_base, L, center, extent, 0, stream);
}

int dlpd_zfft_channels_last(const float* cl, const float* R, void* wsA, int nb, int C, int CT_out, int c_base, int L,
                            float center, void* stream) {
  return dlpd_zfft_channels_last_ext(cl, R, wsA, nb, C, CT_out, c_base, L, center, 0, stream);
}

int dlpd_zfft_into(const float* vol, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                   long long vol_bstride, int do_rotate, float center, void* stream) {
  return dlpd_zfft_oriented(vol, R, wsA, nb, CT, CT_out, c_base, L, vol_bstride, do_rotate, center, 0, stream);
}

int dlpd_zfft(const float* vol, const float* R, void* wsA, int nb, int CT, int L, long long vol_bstride,
              int do_rotate, float center, void* stream) {
  return dlpd_zfft_into(vol, R, wsA, nb, CT, CT, 0, L, vol_bstride, do_rotate, center, stream);
}

// Padded 3-D R2C spectrum of nvol real (L^3) volumes: spec (nvol, NZ, N, N) [kz][kx][ky], times scale.
int dlpd_rfft3d_padded(const float* vol, void* spec, void* wsA, int nvol, int L, float scale, void* stream) {
  if (!vol || !spec || !wsA || nvol <= 0) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int rc = dlpd_zfft(vol, nullptr, wsA, 1, nvol, L, 0, 0, 0.f, stream);
  if (rc) return rc;
  return dlpd_k2_forward((const cplx*)wsA, (cplx*)spec, nvol, 1, L, scale, st);
}

int dlpd_orientation_supported(int L) { return dlpd_k2_orientation_supported(L); }

// wsA (nb, CT, NZ, L, L) x rec (CT or nb*CT spectra) -> wsB (nb, CT, NZ, N, N)
int dlpd_xy_correlate_oriented(const void* wsA, const void* rec, void* wsB, int nb, int CT, int L,
                               long long rec_bstride, int transposed, void* stream) {
  if (!wsA || !rec || !wsB || nb <= 0 || CT <= 0) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  return dlpd_k2_correlate((const cplx*)wsA, (const cplx*)rec, (cplx*)wsB, CT, nb, L, rec_bstride, st, transposed);
}

int dlpd_xy_correlate(const void* wsA, const void* rec, void* wsB, int nb, int CT, int L, long long rec_bstride,
                      void* stream) {
  return dlpd_xy_correlate_oriented(wsA, rec, wsB, nb, CT, L, rec_bstride, 0, stream);
}

// One receptor for the whole batch, re-ordered once (dlpd_receptor_pack) into the order K2 reads it: boxes 80 and 40
long long dlpd_receptor_packed_floats(int CT, int L) { return dlpd_k2_packed_receptor_floats(CT, L); }
int dlpd_receptor_pack(const void* rec, void* packed, int CT, int L, void* stream) {
  if (!rec || !packed || CT <= 0) return DLPD_ERR_ARG;
  return dlpd_k2_pack_receptor((const cplx*)rec, packed, CT, L, (hipStream_t)stream);
}
int dlpd_xy_correlate_packed(const void* wsA, const void* rec_packed, void* wsB, int nb, int CT, int L, void* stream) {
  if (!wsA || !rec_packed || !wsB || nb <= 0 || CT <= 0) return DLPD_ERR_ARG;
  return dlpd_k2_correlate_packed((const cplx*)wsA, (const cplx*)rec_packed, (cplx*)wsB, CT, nb, L, (hipStream_t)stream);
}
// ... going by a per-rotation pencil map (dlpd_rotated_occupancy's second output) for channels [0, nmasked): pencils the map
// marks empty are not read from wsA (K1 with skip_empty did not write them); the packed-receptor boxes only
int dlpd_xy_correlate_packed_occ(const void* wsA, const void* rec_packed, void* wsB, int nb, int CT, int L,
                                 const unsigned* pencil_map, int nmasked, void* stream) {
  if (!wsA || !rec_packed || !wsB || !pencil_map || nb <= 0 || CT <= 0 || nmasked < 0 || nmasked > CT) return DLPD_ERR_ARG;
  return dlpd_k2_correlate_packed((const cplx*)wsA, (const cplx*)rec_packed, (cplx*)wsB, CT, nb, L, (hipStream_t)stream,
                                  (const unsigned char*)pencil_map, nmasked);
}

// wsB -> real correlation volumes out (nb, CT, N^3), optional clamp
// channels [0, nclip) are clamped to +-clip, the rest (e.g. the clash correlation) are left alone
int dlpd_zifft_real_part(const void* wsB, float* out, int nb, int CT, int nclip, int L, int has_clip, float clip,
                         void* stream) {
  if (!wsB || !out || nb <= 0 || CT <= 0) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch (L) {
    case 32: return launch_k3<64, 0, 0>((const cplx*)wsB, out, CT, nclip, 0, nb, nullptr, nullptr, nullptr, 0.f, has_clip, clip, 0.f, st);
    case 40: return launch_k3<80, 0, 0>((const cplx*)wsB, out, CT, nclip, 0, nb, nullptr, nullptr, nullptr, 0.f, has_clip, clip, 0.f, st);
    case 64: return launch_k3<128, 0, 0>((const cplx*)wsB, out, CT, nclip, 0, nb, nullptr, nullptr, nullptr, 0.f, has_clip, clip, 0.f, st);
    case 80: return launch_k3<160, 0, 0>((const cplx*)wsB, out, CT, nclip, 0, nb, nullptr, nullptr, nullptr, 0.f, has_clip, clip, 0.f, st);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// wsB (nb, C, NZ, N, N) -> pre (nb, HP, N^3) = b1 + W1rows^T clamp(correlations): z C2R fused with the (linear) first
// layer over these C channels, the coarse-resolution half of the filter on its own grid
int dlpd_zifft_preact_form(const void* wsB, float* pre, int nb, int C, int L, const float* W1rows, const float* b1, int HP,
                           int has_clip, float clip, int form, void* stream);
int dlpd_zifft_preact(const void* wsB, float* pre, int nb, int C, int L, const float* W1rows, const float* b1, int HP,
                      int has_clip, float clip, void* stream) {
  return dlpd_zifft_preact_form(wsB, pre, nb, C, L, W1rows, b1, HP, has_clip, clip, 0, stream);
}
int dlpd_zifft_preact_form(const void* wsB, float* pre, int nb, int C, int L, const float* W1rows, const float* b1, int HP,
                           int has_clip, float clip, int form, void* stream) {
  if (!wsB || !pre || !W1rows || !b1 || nb <= 0 || C <= 0) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const cplx* B = (const cplx*)wsB;
  if (form == 0) form = DLPD_K3_DEFAULT_FORM;
  if (form == 2 && dlpd_k3r_supported(L, HP, 2)) return dlpd_k3r_preact(B, pre, C, nb, L, W1rows, HP, b1, has_clip, clip, st);
#define DLPD_ZP(NN, H) case H: return launch_k3<NN, H, 2>(B, pre, C, C, 0, nb, W1rows, b1, b1, 0.f, has_clip, clip, 0.f, st)
#define DLPD_ZPN(NN) switch (HP) { DLPD_ZP(NN, 2); DLPD_ZP(NN, 4); DLPD_ZP(NN, 8); DLPD_ZP(NN, 16); DLPD_ZP(NN, 24); DLPD_ZP(NN, 32); \
                                   default: return DLPD_ERR_UNSUPPORTED; }
  switch (L) {
    case 32: DLPD_ZPN(64)
#ifdef DLPD_TEST_VARIANTS
    case 40: DLPD_ZPN(80)                      // (product: the role-split kernel above; coarse grids of 64 / 80 have no fine grid)
    case 64: DLPD_ZPN(128)
    case 80: DLPD_ZPN(160)
#endif
    default: return DLPD_ERR_UNSUPPORTED;
  }
#undef DLPD_ZPN
#undef DLPD_ZP
}

int dlpd_zifft_real(const void* wsB, float* out, int nb, int CT, int L, int has_clip, float clip, void* stream) {
  return dlpd_zifft_real_part(wsB, out, nb, CT, CT, L, has_clip, clip, stream);
}

// wsB -> V (nb, N^3): z C2R + clip + MLP + clash mask
// aux (nb, Caux, (N/2)^3): already-real first-layer inputs of a coarser resolution (may be null)
int dlpd_zifft_filter_cand(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                           const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                           const float* aux, int Caux, int aux_is_preact, const void* tau, void* cand_keys,
                           void* cand_count, int cap, void* stream);
int dlpd_zifft_filter_form(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                           const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                           const float* aux, int Caux, int aux_is_preact, const void* tau, void* cand_keys,
                           void* cand_count, int cap, int form, void* stream);

int dlpd_zifft_filter_aux(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                          const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                          const float* aux, int Caux, int aux_is_preact, void* stream) {
  return dlpd_zifft_filter_cand(wsB, V, nb, C, has_clash, L, W1t, b1, W2, b2, HP, has_clip, clip, thr, aux, Caux,
                                aux_is_preact, nullptr, nullptr, nullptr, 0, stream);
}

// dlpd_zifft_filter_aux that also feeds the top-K stage's candidate lists (dlpd_topk_select_cand): every score whose
// key is <= *tau (published by dlpd_topk_merge_tau) is appended to cand_keys (nb, cap); see dlpd_topk.hip.
int dlpd_zifft_filter_cand(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                           const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                           const float* aux, int Caux, int aux_is_preact, const void* tau, void* cand_keys,
                           void* cand_count, int cap, void* stream) {
  return dlpd_zifft_filter_form(wsB, V, nb, C, has_clash, L, W1t, b1, W2, b2, HP, has_clip, clip, thr, aux, Caux,
                                aux_is_preact, tau, cand_keys, cand_count, cap, 0, stream);
}

// The same with the kernel formulation named: form 0 = the library's default, 1 = channel-owning waves with
// barrier-separated transform / filter phases (k_zifft_filter[_tiles]), 2 = role-split transform / filter waves
// (dlpd_k3r.hip; falls back to 1 where it is not compiled).  Same arithmetic, same results bit for bit.
int dlpd_zifft_filter_form(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                           const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                           const float* aux, int Caux, int aux_is_preact, const void* tau, void* cand_keys,
                           void* cand_count, int cap, int form, void* stream) {
  if (!wsB || !V || !W1t || !b1 || !W2 || nb <= 0 || C <= 0 || Caux < 0 || (Caux > 0 && !aux)) return DLPD_ERR_ARG;
  if (L % 2) return DLPD_ERR_ARG;
  if (cand_keys && (!tau || !cand_count || cap <= 0)) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int CT = C + (has_clash ? 1 : 0);
  const K3Aux ax = {aux, Caux, L, (Caux > 0 && aux_is_preact) ? 1 : 0};              // coarse grid N/2 = L
  const K3Cand cd = {(const unsigned*)tau, (unsigned long long*)cand_keys, (unsigned*)cand_count, cap, nb};
  if (form == 0) form = DLPD_K3_DEFAULT_FORM;
  if (form == 2 && dlpd_k3r_supported(L, HP, 1) && (Caux == 0 || aux_is_preact))
    return dlpd_k3r_filter((const cplx*)wsB, V, CT, C, has_clash, nb, L, W1t, HP, b1, W2, b2, has_clip, clip, thr, ax, cd, st);
  switch (L) {
    case 32: return k3_filter_dispatch<64>(HP, (const cplx*)wsB, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, ax, cd);
    case 40: return k3_filter_dispatch<80>(HP, (const cplx*)wsB, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, ax, cd);
#ifdef DLPD_TEST_VARIANTS
    // the channel-owning formulation at N = 128 / 160: no default path reaches it (the role-split kernel covers every
    // hidden width it has), it is the bit-exactness reference of the tests -- compiled into tests/variants only
    case 64: return k3_filter_dispatch<128>(HP, (const cplx*)wsB, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, ax, cd);
    case 80: return k3_filter_dispatch<160>(HP, (const cplx*)wsB, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, ax, cd);
#endif
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

int dlpd_zifft_filter(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                      const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                      void* stream) {
  return dlpd_zifft_filter_aux(wsB, V, nb, C, has_clash, L, W1t, b1, W2, b2, HP, has_clip, clip, thr, nullptr, 0,
                               0, stream);
}

// Fused driver for one batch of rotations (single-resolution model):
//   lig (CT, L^3) [score channels then, if has_clash, the ligand forbidden volume]
//   recF (CT, NZ, N, N) from dlpd_rfft3d_padded(receptor, scale = 1/N^3)
//   R (nb, 9) -> V (nb, N^3).  wsA: nb*CT*NZ*L*L cplx, wsB: nb*CT*NZ*N*N cplx.
int dlpd_score_rotations_oriented(const float* lig, const void* recF, const float* R, int nb, int C, int has_clash,
                                  int L, float center, const float* W1t, const float* b1, const float* W2, float b2,
                                  int HP, int has_clip, float clip, float thr, void* wsA, void* wsB, float* V,
                                  int transposed, void* stream);

int dlpd_score_rotations(const float* lig, const void* recF, const float* R, int nb, int C, int has_clash, int L,
                         float center, const float* W1t, const float* b1, const float* W2, float b2, int HP,
                         int has_clip, float clip, float thr, void* wsA, void* wsB, float* V, void* stream) {
  return dlpd_score_rotations_oriented(lig, recF, R, nb, C, has_clash, L, center, W1t, b1, W2, b2, HP, has_clip, clip,
                                       thr, wsA, wsB, V, 0, stream);
}

int dlpd_score_rotations_oriented(const float* lig, const void* recF, const float* R, int nb, int C, int has_clash,
                                  int L, float center, const float* W1t, const float* b1, const float* W2, float b2,
                                  int HP, int has_clip, float clip, float thr, void* wsA, void* wsB, float* V,
                                  int transposed, void* stream) {
  const int CT = C + (has_clash ? 1 : 0);
  int rc = dlpd_zfft_oriented(lig, R, wsA, nb, CT, CT, 0, L, 0, 1, center, transposed, stream);
  if (rc) return rc;
  rc = dlpd_xy_correlate_oriented(wsA, recF, wsB, nb, CT, L, 0, transposed, stream);
  if (rc) return rc;
  return dlpd_zifft_filter(wsB, V, nb, C, has_clash, L, W1t, b1, W2, b2, HP, has_clip, clip, thr, stream);
}

// Vectorised filter: weights padded to HP (dlpd_hidden_pad), strided channel/batch layouts.
int dlpd_filter_preact(const float* conv1, int C1, int N1, const float* W1rows, const float* b1, int HP,
                       float* pre, int nb, void* stream) {
  if (!conv1 || !W1rows || !b1 || !pre || nb <= 0 || C1 <= 0 || N1 <= 0 || N1 % 4) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)N1 * N1 * N1;
  size_t nblk = ((size_t)nb * n / 4 + 255) / 256;
  if (nblk > 65536) nblk = 65536;
#define DLPD_FP(H) case H: DLPD_LAUNCH((k_filter_preact<H>), dim3((unsigned)nblk), dim3(256), 0, st, conv1, C1, n, \
                                       W1rows, b1, pre, nb); return dlpd_check_launch()
  switch (HP) {
    DLPD_FP(2); DLPD_FP(4); DLPD_FP(8); DLPD_FP(16); DLPD_FP(24); DLPD_FP(32);
    default: return DLPD_ERR_UNSUPPORTED;
  }
#undef DLPD_FP
}

int dlpd_filter_volumes(const float* conv0, int C0, long long conv0_bstride, int N0, const float* conv1, int C1,
                        int N1, int conv1_is_preact, const float* mask_norm, long long mask_bstride, float thr, int has_clash,
                        const float* W1t, const float* b1, const float* W2, float b2, int HP, float* V, int nb,
                        void* stream) {
  if (!conv0 || !V || !W1t || !b1 || !W2 || nb <= 0 || C0 <= 0 || N0 <= 0 || N0 % 4) return DLPD_ERR_ARG;
  if (C1 > 0 && (!conv1 || N1 <= 0 || N0 % N1 != 0 || (N0 / N1 == 2 && N1 % 2))) return DLPD_ERR_ARG;
  if (conv1_is_preact && (C1 <= 0 || N0 / N1 > 2)) return DLPD_ERR_ARG;
  if (has_clash && !mask_norm) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define DLPD_FV(H) case H: return launch_filter_vec<H>(conv0, C0, conv0_bstride, N0, conv1, C1, N1, conv1_is_preact, mask_norm, \
                                                      mask_bstride, thr, has_clash, W1t, b1, W2, b2, V, nb, st)
  switch (HP) {
    DLPD_FV(2); DLPD_FV(4); DLPD_FV(8); DLPD_FV(16); DLPD_FV(24); DLPD_FV(32);
    default: return DLPD_ERR_UNSUPPORTED;
  }
#undef DLPD_FV
}

// Generic per-voxel filter over real correlation volumes (multi-resolution model).
int dlpd_filter_mask(const float* conv0, int C0, int N0, const float* conv1, int C1, int N1, const float* mask_norm,
                     float thr, int has_clash, const float* W1t, const float* b1, const float* W2, float b2, int H,
                     float* V, int nb, void* stream) {
  if (!conv0 || !V || !W1t || !b1 || !W2 || nb <= 0 || H <= 0 || H > DLPD_MAX_HIDDEN) return DLPD_ERR_ARG;
  if (C1 > 0 && (!conv1 || N1 <= 0 || N0 % N1 != 0)) return DLPD_ERR_ARG;
  if (has_clash && !mask_norm) return DLPD_ERR_ARG;
  if (dlpd_hidden_pad(H) == H && N0 % 4 == 0 && (C1 == 0 || N0 / N1 != 2 || N1 % 2 == 0))   // no padding needed
    return dlpd_filter_volumes(conv0, C0, (long long)C0 * N0 * N0 * N0, N0, conv1, C1, N1, 0, mask_norm,
                               (long long)N0 * N0 * N0, thr, has_clash, W1t, b1, W2, b2, H, V, nb, stream);
  const size_t total = (size_t)nb * N0 * N0 * N0;
  size_t nblk = (total + 255) / 256;
  if (nblk > 16384) nblk = 16384;
  DLPD_LAUNCH(k_filter_generic, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, conv0, C0, N0, conv1, C1, N1,
              mask_norm, thr, has_clash, W1t, b1, W2, b2, H, V, nb);
  return dlpd_check_launch();
}

}  // extern "C"
