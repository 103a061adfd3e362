// Atom-level front end of the search: rigid transform of typed coordinates
// and their projection onto the density grid, replacing
//   /root/reference/src/Docker/Docker.py:221-224  CPU CoordsRotate + CoordsTranslate, H2D copy,
//                                                  TorchProteinLibrary TypedCoords2Volume, channel sum
//   /root/reference/src/Docker/Docker.py:204,208  TypedCoords2Volume of receptor / ligand
// by ONE kernel that rotates on the fly (no per-batch host round trip).
// TorchProteinLibrary's source is absent: the density shape is BUILD-DEFINED (parity unpinned) and therefore a
// PARAMETER (dlpd_project_atoms_ext; deeplocalproteindocking_amd/Utils/Conventions.py; scripts/calibrate_tpl.py finds the
// values on a machine that has the library): every atom adds exp(-|r - x|^2 / (2 sigma^2)) (Angstrom^2) to the
// (2d+1)^3 voxels around it (times a prefactor `norm`), voxel (i,j,k) sitting at (i,j,k + voxel_offset) * resolution.
// Defaults: sigma 1, d 2, voxel_offset 0, norm 1.
//
// DETERMINISTIC: the contributions are accumulated as 2^-22 fixed-point UNSIGNED integers (integer atomic adds
// commute, float ones do not; every contribution is positive), then converted to float in place.  Two runs -- and
// two ranks projecting the same receptor -- produce bit-identical volumes, so the clash mask `corr < threshold`
// (Docker.py:226) and with it the ranked list never depends on the order in which atoms happen to be added.
// Resolution and range: a contribution is rounded to a multiple of 2^-22, i.e. by at most 2^-23 = 1.2e-7 -- one ulp of
// 1.0f, what a float sum of such contributions (a unit-sigma Gaussian per atom at protein packing density, ~0.1 heavy
// atoms / A^3, sums to 1-2 per voxel) loses per addition anyway; the 32-bit accumulator holds 1023 per voxel --
// a thousand atoms stacked on one site (duplicated records, multi-model files summed into one channel) before it could
// wrap.  (Round 3 used 2^-20: four times the range for four times the rounding step; round 2 2^-24 signed: 127.)
#include <dlpd_platform.h>
#include "dlpd_internal.h"

#define DLPD_SPLAT_SCALE 4194304.0f           // 2^22
#define DLPD_SPLAT_MAX_WINDOW 6

// coords (B, 3*stride_atoms) f32 [x0 y0 z0 x1 ...] ordered by atom type; ntype (B, T) counts,
// offs (B, T) first atom of each type.  p' = R_b p + shift (R row-major, may be null).
// out (B, T, L^3), or (B, 1, L^3) when sum_types != 0.  One thread per (b, atom, x-plane of its window): 2d + 1 threads share
// an atom (round 6: one thread per atom left 56 blocks of 125 serial atomics each -- 80-110 us per call, and Docker.dockSE3
// makes one per batch, Docker.dockE3 two); every voxel gets the same contribution, integer adds commute: the same bits.
__global__ void __launch_bounds__(256)
k_project_atoms(const float* __restrict__ coords, const int* __restrict__ ntype, const int* __restrict__ offs,
                const float* __restrict__ R, float sx, float sy, float sz, unsigned* __restrict__ out, int B,
                int stride_atoms, int T, int L, float res, int sum_types, float inv2s2, int d, float voff, float norm) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nw = 2 * d + 1, wi = gid % nw, ga = gid / nw;
  const int b = ga / stride_atoms, a = ga % stride_atoms;
  if (b >= B) return;
  int ty = -1;
  for (int t = 0; t < T; t++) {
    const int o = offs[b * T + t];
    if (a >= o && a < o + ntype[b * T + t]) ty = t;
  }
  if (ty < 0) return;                                   // padding slot
  const float* p = coords + ((size_t)b * stride_atoms + a) * 3;
  float x = p[0], y = p[1], z = p[2];
  if (R) {
    const float* r = R + (size_t)b * 9;
    const float rx = r[0] * x + r[1] * y + r[2] * z;
    const float ry = r[3] * x + r[4] * y + r[5] * z;
    const float rz = r[6] * x + r[7] * y + r[8] * z;
    x = rx; y = ry; z = rz;
  }
  x += sx; y += sy; z += sz;
  // (voff = 0 leaves every expression below at the value it had without the parameter)
  const int ci = (int)floorf(x / res - voff), cj = (int)floorf(y / res - voff), ck = (int)floorf(z / res - voff);
  const int ch = sum_types ? 0 : ty, nch = sum_types ? 1 : T;
  unsigned* vol = out + ((size_t)b * nch + ch) * L * L * L;
  {
    const int i = ci - d + wi;                          // this thread's plane of the window
    if (i < 0 || i >= L) return;
    const float dx = x - (i + voff) * res;
    for (int j = cj - d; j <= cj + d; j++) {
      if (j < 0 || j >= L) continue;
      const float dy = y - (j + voff) * res;
      for (int k = ck - d; k <= ck + d; k++) {
        if (k < 0 || k >= L) continue;
        const float dz = z - (k + voff) * res;
        const float w = norm * expf(-inv2s2 * (dx * dx + dy * dy + dz * dz));      // (norm = 1: the same value)
        atomicAdd(&vol[((size_t)i * L + j) * L + k], (unsigned)rintf(w * DLPD_SPLAT_SCALE));
      }
    }
  }
}

// fixed point -> float, in place (both are 4 bytes per voxel)
__global__ void __launch_bounds__(256) k_splat_to_float(unsigned* __restrict__ acc, size_t n) {
  float* f = reinterpret_cast<float*>(acc);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned v = acc[i];
    f[i] = (float)v * (1.0f / DLPD_SPLAT_SCALE);
  }
}

// ---- the same projection CELL-WISE (round 6): a protein fills a few per cent of its box, and Docker.dockE3 projects sixteen
// poses per batch (Docker.py:163-165) -- clearing, converting and then scanning (sum over types, tile occupancy) 360 MB that
// are almost all zero cost more than everything the atoms touch.  Here the 4 x 4 x 4 cells an atom's window can reach are
// marked first; only marked cells are cleared, accumulated into and converted; the rest of `out` is NOT WRITTEN and `occ` (the
// maps of dlpd_conv3d_tile_occupancy: one byte per cell, all types) says which is which -- for consumers that go by the map
// (dlpd_conv3d_split_sparse with unwritten != 0).  Same accumulation, same conversion: the same values where the map is set.
__global__ void __launch_bounds__(256)
k_mark_atom_cells(const float* __restrict__ coords, const int* __restrict__ ntype, const int* __restrict__ offs,
                  const float* __restrict__ R, float sx, float sy, float sz, unsigned char* __restrict__ occ, int B, int stride_atoms,
                  int T, int L, float res, int d, float voff) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid / stride_atoms, a = gid % stride_atoms;
  if (b >= B) return;
  bool typed = false;
  for (int t = 0; t < T; t++) {
    const int o = offs[b * T + t];
    typed |= (a >= o && a < o + ntype[b * T + t]);
  }
  if (!typed) return;
  const float* p = coords + ((size_t)b * stride_atoms + a) * 3;
  float x = p[0], y = p[1], z = p[2];
  if (R) {                                              // (the same expressions as k_project_atoms: the same voxel indices)
    const float* r = R + (size_t)b * 9;
    const float rx = r[0] * x + r[1] * y + r[2] * z;
    const float ry = r[3] * x + r[4] * y + r[5] * z;
    const float rz = r[6] * x + r[7] * y + r[8] * z;
    x = rx; y = ry; z = rz;
  }
  x += sx; y += sy; z += sz;
  const int ci = (int)floorf(x / res - voff), cj = (int)floorf(y / res - voff), ck = (int)floorf(z / res - voff);
  const int nc = (L + 3) / 4;
  const int i0 = max(ci - d, 0) >> 2, i1 = min(ci + d, L - 1) >> 2, j0 = max(cj - d, 0) >> 2, j1 = min(cj + d, L - 1) >> 2;
  const int k0 = max(ck - d, 0) >> 2, k1 = min(ck + d, L - 1) >> 2;
  if (ci + d < 0 || cj + d < 0 || ck + d < 0 || ci - d >= L || cj - d >= L || ck - d >= L) return;   // window outside the box
  for (int i = i0; i <= i1; i++)
    for (int j = j0; j <= j1; j++)
      for (int k = k0; k <= k1; k++) occ[(((size_t)b * nc + i) * nc + j) * nc + k] = 1;     // (plain stores of the same value)
}

// one thread per voxel of every cell: MODE 0 clears the marked cells' accumulators, MODE 1 converts them to float
template <int MODE> __global__ void __launch_bounds__(256)
k_marked_cells(const unsigned char* __restrict__ occ, unsigned* __restrict__ acc, int B, int T, int L) {
  const int nc = (L + 3) / 4;
  const size_t total = (size_t)B * nc * nc * nc * 64;
  const size_t L3 = (size_t)L * L * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t cell = i >> 6;
    if (!occ[cell]) continue;
    const int v = (int)(i & 63), cz = (int)(cell % nc), cy = (int)((cell / nc) % nc), cx = (int)((cell / ((size_t)nc * nc)) % nc);
    const int b = (int)(cell / ((size_t)nc * nc * nc));
    const int x = 4 * cx + (v >> 4), y = 4 * cy + ((v >> 2) & 3), z = 4 * cz + (v & 3);
    if (x >= L || y >= L || z >= L) continue;
    unsigned* a = acc + (size_t)b * T * L3 + ((size_t)x * L + y) * L + z;
    for (int t = 0; t < T; t++) {
      if (MODE == 0) a[(size_t)t * L3] = 0u;
      else reinterpret_cast<float*>(a)[(size_t)t * L3] = (float)a[(size_t)t * L3] * (1.0f / DLPD_SPLAT_SCALE);
    }
  }
}

extern "C" {

int dlpd_project_atoms_cells(const float* coords, const int* num_atoms_of_type, const int* offsets, const float* R,
                             float shift_x, float shift_y, float shift_z, float* out, unsigned char* occ, int B, int stride_atoms,
                             int ntypes, int L, float resolution, float sigma, int window, float voxel_offset, float norm,
                             void* stream) {
  if (!coords || !num_atoms_of_type || !offsets || !out || !occ || B <= 0 || stride_atoms <= 0 || ntypes <= 0 || L <= 0 ||
      resolution <= 0.f || !(sigma > 0.f) || window < 0 || window > DLPD_SPLAT_MAX_WINDOW || !(norm > 0.f) || norm > 64.f)
    return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nc = (L + 3) / 4;
  const size_t ncell = (size_t)B * nc * nc * nc;
  if (hipMemsetAsync(occ, 0, ncell, st) != hipSuccess) return DLPD_ERR_LAUNCH;
  const int total = B * stride_atoms;
  const int total_w = total * (2 * window + 1);          // k_project_atoms: one thread per (atom, window plane)
  DLPD_LAUNCH(k_mark_atom_cells, dim3((total + 255) / 256), dim3(256), 0, st, coords, num_atoms_of_type, offsets, R, shift_x, shift_y,
              shift_z, occ, B, stride_atoms, ntypes, L, resolution, window, voxel_offset);
  size_t nblk = (ncell * 64 + 255) / 256;
  if (nblk > 32768) nblk = 32768;
  DLPD_LAUNCH(k_marked_cells<0>, dim3((unsigned)nblk), dim3(256), 0, st, occ, reinterpret_cast<unsigned*>(out), B, ntypes, L);
  DLPD_LAUNCH(k_project_atoms, dim3((total_w + 255) / 256), dim3(256), 0, st, coords, num_atoms_of_type, offsets, R,
              shift_x, shift_y, shift_z, reinterpret_cast<unsigned*>(out), B, stride_atoms, ntypes, L, resolution, 0,
              0.5f / (sigma * sigma), window, voxel_offset, norm);
  DLPD_LAUNCH(k_marked_cells<1>, dim3((unsigned)nblk), dim3(256), 0, st, occ, reinterpret_cast<unsigned*>(out), B, ntypes, L);
  return dlpd_check_launch();
}

// Clears `out`, accumulates the densities in fixed point, converts to float: bit-reproducible.
int dlpd_project_atoms_ext(const float* coords, const int* num_atoms_of_type, const int* offsets, const float* R,
                           float shift_x, float shift_y, float shift_z, float* out, int B, int stride_atoms,
                           int ntypes, int L, float resolution, int sum_types, float sigma, int window,
                           float voxel_offset, float norm, void* stream) {
  if (!coords || !num_atoms_of_type || !offsets || !out || B <= 0 || stride_atoms <= 0 || ntypes <= 0 || L <= 0 ||
      resolution <= 0.f || !(sigma > 0.f) || window < 0 || window > DLPD_SPLAT_MAX_WINDOW || !(norm > 0.f) || norm > 64.f)
    return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)B * (sum_types ? 1 : ntypes) * L * L * L * sizeof(float);
  if (hipMemsetAsync(out, 0, bytes, st) != hipSuccess) return DLPD_ERR_LAUNCH;
  const int total = B * stride_atoms * (2 * window + 1);      // one thread per (atom, window plane)
  DLPD_LAUNCH(k_project_atoms, dim3((total + 255) / 256), dim3(256), 0, st, coords, num_atoms_of_type, offsets, R,
              shift_x, shift_y, shift_z, reinterpret_cast<unsigned*>(out), B, stride_atoms, ntypes, L, resolution, sum_types,
              0.5f / (sigma * sigma), window, voxel_offset, norm);
  const size_t n = bytes / sizeof(float);
  size_t nblk = (n + 255) / 256;
  if (nblk > 16384) nblk = 16384;
  DLPD_LAUNCH(k_splat_to_float, dim3((unsigned)nblk), dim3(256), 0, st, reinterpret_cast<unsigned*>(out), n);
  return dlpd_check_launch();
}

int dlpd_project_atoms(const float* coords, const int* num_atoms_of_type, const int* offsets, const float* R,
                       float shift_x, float shift_y, float shift_z, float* out, int B, int stride_atoms,
                       int ntypes, int L, float resolution, int sum_types, void* stream) {
  return dlpd_project_atoms_ext(coords, num_atoms_of_type, offsets, R, shift_x, shift_y, shift_z, out, B, stride_atoms,
                                ntypes, L, resolution, sum_types, 1.0f, 2, 0.0f, 1.0f, stream);
}

}  // extern "C"
