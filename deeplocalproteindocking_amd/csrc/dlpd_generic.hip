// Circular cross-correlation on the 2L zero-padded grid for ANY box size L (the reference's `box_size` is a free
// constructor argument, /root/reference/src/Docker/Docker.py:18,22-24,31): the stand-alone VolumeConvolution op
// (Docker.py:32,225; DockingModels.py:48,71) where the fused pipeline has no compiled plan (L other than 32 / 40 / 64 / 80).
//
//     out[v][t mod N] = sum_r v1[v][r + t] v2[v][r],  N = 2L        (semantics: src/Models/MultiplyVolumes.py:13-47)
//     = IDFT3( DFT3(pad v1) . conj(DFT3(pad v2)) ) / N^3
//
// A correctness path, not a tuned one: every 1-D transform is a direct O(n N) sum from an LDS-resident tile (any length,
// no radix plan), pruned to the L non-zero inputs on the way in.  3 N^4 complex multiply-adds per volume for the inverse
// (~5e9 at box 100): tens of milliseconds per rotation of a 48-channel pair, against ~0.4 ms on the compiled plans.
#include <dlpd_platform.h>
#include "dlpd_fft.h"
#include "dlpd_internal.h"

#define GEN_MAXN 256                      // box sizes up to 128 (tile: n_in x 64 complex <= 128 KB)

// transform along a MIDDLE axis: in (outer, n_in, inner) -> out (outer, N, inner), inner contiguous;
// out[o][k][i] = sum_j in[o][j][i] exp(dir 2 pi i j k / N).  grid (ceil(inner / 64), outer), block 256 = 64 columns x 4
// groups of output rows; each thread takes its column for four consecutive k at a time.
__global__ void __launch_bounds__(256) k_gdft_mid(const cplx* __restrict__ in, cplx* __restrict__ out, int n_in, int N,
                                                  long long inner, int dir) {
  DLPD_DYN_SHARED(cplx, S);
  cplx* tile = S;                         // [n_in][64]
  cplx* tw = S + (size_t)n_in * 64;       // exp(dir 2 pi i k / N)
  const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
  const long long i0 = (long long)blockIdx.x * 64, o = blockIdx.y;
  for (int k = tid; k < N; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)N, &s, &co);
    tw[k] = c_make((float)co, (float)(dir * s));
  }
  const cplx* src = in + (size_t)o * n_in * inner;
  for (int e = tid; e < n_in * 64; e += 256) {
    const int j = e >> 6, cc = e & 63;
    tile[e] = (i0 + cc < inner) ? src[(size_t)j * inner + i0 + cc] : c_make(0.f, 0.f);
  }
  __syncthreads();
  if (i0 + c >= inner) return;
  cplx* dst = out + (size_t)o * N * inner + i0 + c;
  for (int k0 = 4 * g; k0 < N; k0 += 16) {
    float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f};
    int idx[4] = {0, 0, 0, 0};
    for (int j = 0; j < n_in; j++) {
      const cplx v = tile[j * 64 + c];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const cplx w = tw[idx[u]];
        ar[u] = fmaf(v.x, w.x, fmaf(-v.y, w.y, ar[u]));
        ai[u] = fmaf(v.x, w.y, fmaf(v.y, w.x, ai[u]));
        // (j k) mod N, incrementally.  The step is reduced mod N first: for an odd box N = 2L is not a multiple of 4
        // and the lanes k0 + u >= N of the last round (their results are discarded below) would otherwise walk idx past
        // the table -- a read beyond the LDS allocation (harmless zeros on the GPU, a heap over-read in the emulator)
        idx[u] += (k0 + u < N) ? (k0 + u) : (k0 + u - N);
        if (idx[u] >= N) idx[u] -= N;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k0 + u < N) dst[(size_t)(k0 + u) * inner] = c_make(ar[u], ai[u]);
  }
}

// transform along the LAST (contiguous) axis of P pencils: in (P, n_in) real or complex -> out (P, N) complex, or real
// (the real part times `scale`, clamped to +-clip if has_clip).  grid ceil(P / 16), block 256 = 16 pencils x 16 lanes.
__global__ void __launch_bounds__(256) k_gdft_last(const void* __restrict__ in, void* __restrict__ out, long long P,
                                                   int n_in, int N, int dir, int in_real, int out_real, float scale,
                                                   int has_clip, float clip) {
  DLPD_DYN_SHARED(cplx, S);
  const int RSL = n_in + 1;
  cplx* tile = S;                         // [16][n_in + 1]
  cplx* tw = S + 16 * RSL;
  const int tid = threadIdx.x, p = tid >> 4, kl = tid & 15;
  const long long p0 = (long long)blockIdx.x * 16;
  for (int k = tid; k < N; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)N, &s, &co);
    tw[k] = c_make((float)co, (float)(dir * s));
  }
  for (int e = tid; e < 16 * n_in; e += 256) {
    const int pp = e / n_in, j = e % n_in;
    cplx v = c_make(0.f, 0.f);
    if (p0 + pp < P) {
      if (in_real) v.x = reinterpret_cast<const float*>(in)[(size_t)(p0 + pp) * n_in + j];
      else v = reinterpret_cast<const cplx*>(in)[(size_t)(p0 + pp) * n_in + j];
    }
    tile[pp * RSL + j] = v;
  }
  __syncthreads();
  if (p0 + p >= P) return;
  for (int k = kl; k < N; k += 16) {
    float ar = 0.f, ai = 0.f;
    int idx = 0;
    for (int j = 0; j < n_in; j++) {
      const cplx v = tile[p * RSL + j], w = tw[idx];
      ar = fmaf(v.x, w.x, fmaf(-v.y, w.y, ar));
      ai = fmaf(v.x, w.y, fmaf(v.y, w.x, ai));
      idx += k;
      if (idx >= N) idx -= N;
    }
    if (out_real) {
      float r = ar * scale;
      if (has_clip) r = DLPD_CLAMP(r, clip);
      reinterpret_cast<float*>(out)[(size_t)(p0 + p) * N + k] = r;
    } else {
      reinterpret_cast<cplx*>(out)[(size_t)(p0 + p) * N + k] = c_make(ar, ai);
    }
  }
}

// a <- a * conj(b)
__global__ void __launch_bounds__(256) k_gmul_conj(cplx* __restrict__ a, const cplx* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = c_mulc(a[i], b[i]);
}

static int gen_mid(const cplx* in, cplx* out, long long outer, int n_in, int N, long long inner, int dir, hipStream_t st) {
  const size_t shmem = ((size_t)n_in * 64 + N) * sizeof(cplx);
  int rc = dlpd_set_max_dyn_shared((const void*)k_gdft_mid, shmem);
  if (rc) return rc;
  DLPD_LAUNCH(k_gdft_mid, dim3((unsigned)((inner + 63) / 64), (unsigned)outer), dim3(256), shmem, st, in, out, n_in, N, inner, dir);
  return DLPD_OK;
}
static int gen_last(const void* in, void* out, long long P, int n_in, int N, int dir, int in_real, int out_real, float scale,
                    int has_clip, float clip, hipStream_t st) {
  const size_t shmem = ((size_t)16 * (n_in + 1) + N) * sizeof(cplx);
  int rc = dlpd_set_max_dyn_shared((const void*)k_gdft_last, shmem);
  if (rc) return rc;
  DLPD_LAUNCH(k_gdft_last, dim3((unsigned)((P + 15) / 16)), dim3(256), shmem, st, in, out, P, n_in, N, dir, in_real, out_real, scale,
              has_clip, clip);
  return DLPD_OK;
}

extern "C" {

// bytes of workspace dlpd_correlate_generic needs for nvol volume pairs of box L
size_t dlpd_correlate_generic_ws_bytes(int nvol, int L) {
  if (nvol <= 0 || L <= 0) return 0;
  const size_t N = 2 * (size_t)L;
  return (size_t)nvol * (2 * N * N * N + (size_t)L * L * N + (size_t)L * N * N) * sizeof(cplx);
}

int dlpd_generic_box_supported(int L) { return (L >= 1 && 2 * L <= GEN_MAXN) ? 1 : 0; }

// v1, v2 (nvol, L^3) f32 -> out (nvol, N^3) f32, N = 2L, optional clamp to +-clip (VolumeConvolution(clip),
// DockingModels.py:48); ws: dlpd_correlate_generic_ws_bytes(nvol, L) bytes of device scratch.
int dlpd_correlate_generic(const float* v1, const float* v2, float* out, int nvol, int L, int has_clip, float clip, void* ws,
                           void* stream) {
  if (!v1 || !v2 || !out || !ws || nvol <= 0) return DLPD_ERR_ARG;
  if (!dlpd_generic_box_supported(L)) return DLPD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int N = 2 * L;
  const size_t N3 = (size_t)N * N * N;
  cplx* P = (cplx*)ws;                    // (nvol, N, N, N)
  cplx* Q = P + (size_t)nvol * N3;
  cplx* T1 = Q + (size_t)nvol * N3;       // (nvol, L, L, N)
  cplx* T2 = T1 + (size_t)nvol * L * L * N;   // (nvol, L, N, N)
  int rc = 0;
  for (int which = 0; which < 2 && !rc; which++) {
    const float* v = which ? v2 : v1;
    cplx* F = which ? Q : P;
    // forward, pruned to the L non-zero inputs: z (real -> complex), y, x
    rc = gen_last(v, T1, (long long)nvol * L * L, L, N, -1, 1, 0, 1.f, 0, 0.f, st);
    if (!rc) rc = gen_mid(T1, T2, (long long)nvol * L, L, N, N, -1, st);
    if (!rc) rc = gen_mid(T2, F, nvol, L, N, (long long)N * N, -1, st);
  }
  if (rc) return rc;
  {
    size_t nblk = ((size_t)nvol * N3 + 255) / 256;
    if (nblk > 65536) nblk = 65536;
    DLPD_LAUNCH(k_gmul_conj, dim3((unsigned)nblk), dim3(256), 0, st, P, (const cplx*)Q, (size_t)nvol * N3);
  }
  // inverse: x, y, z (real part, 1 / N^3, clamp)
  rc = gen_mid(P, Q, nvol, N, N, (long long)N * N, +1, st);
  if (!rc) rc = gen_mid(Q, P, (long long)nvol * N, N, N, N, +1, st);
  if (!rc) rc = gen_last(P, out, (long long)nvol * N * N, N, N, +1, 0, 1, 1.0f / ((float)N * (float)N * (float)N), has_clip, clip, st);
  if (rc) return rc;
  return dlpd_check_launch();
}

}  // extern "C"
