/* A C99 client of the C ABI (include/dlpd.h) -- no Python, no torch, no C++: what a maintainer binding libdlpd.so from
 * another language sees.  Test infrastructure (tests/test_c_abi_client.py builds and runs it).
 *
 *   client <library.so> host      the CPU-emulated library of the test-suite (tests/emu): buffers are malloc'ed
 *   client <library.so> hip       libdlpd.so on an MI355X: buffers come from hipMalloc (libamdhip64 is dlopen'ed here,
 *                                 so this file needs no HIP header)
 *
 * It runs three entry points whose results can be checked without an oracle:
 *   1. dlpd_rotate_trilinear with the identity rotation returns the volume itself, bit for bit   (Docker.py:218)
 *   2. dlpd_topk_select on a volume with known minima returns them in the reference's pick order (Docker.py:89-98)
 *   3. dlpd_rfft3d_padded + dlpd_zfft + dlpd_xy_correlate + dlpd_zifft_real of a volume with a one-voxel volume at
 *      (1,2,3): the circular cross-correlation on the 2L grid is the volume shifted by (-1,-2,-3)  (DockingModels.py:71,
 *      MultiplyVolumes.py:13-47)
 * and prints "c-abi client ok" (exit code 0) or the first mismatch (exit code 1). */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dlpd.h"

typedef int (*hip_malloc_t)(void**, size_t);
typedef int (*hip_memcpy_t)(void*, const void*, size_t, int);
typedef int (*hip_sync_t)(void);
typedef int (*hip_free_t)(void*);
static hip_malloc_t p_malloc;
static hip_memcpy_t p_memcpy;
static hip_sync_t p_sync;
static hip_free_t p_free;
static int on_gpu;

static void* dev_alloc(size_t n) {
  void* p = NULL;
  if (!on_gpu) return calloc(1, n);
  if (p_malloc(&p, n) != 0) { fprintf(stderr, "hipMalloc(%zu) failed\n", n); exit(2); }
  return p;
}
static void to_dev(void* d, const void* h, size_t n) { if (on_gpu) p_memcpy(d, h, n, 1); else memcpy(d, h, n); }
static void to_host(void* h, const void* d, size_t n) { if (on_gpu) { p_sync(); p_memcpy(h, d, n, 2); } else memcpy(h, d, n); }
static void dev_free(void* p) { if (on_gpu) p_free(p); else free(p); }

#define SYM(name) name##_fn = (name##_t)dlsym(lib, #name); if (!name##_fn) { fprintf(stderr, "missing symbol %s\n", #name); return 1; }
typedef int (*dlpd_version_t)(void);
typedef int (*dlpd_grid_supported_t)(int);
typedef int (*dlpd_rotate_trilinear_t)(const float*, const float*, float*, int, int, int, long long, float, void*);
typedef size_t (*dlpd_topk_workspace_bytes_t)(int, int);
typedef int (*dlpd_topk_select_t)(const float*, int, long long, int, float*, int*, void*, void*);
typedef int (*dlpd_rfft3d_padded_t)(const float*, void*, void*, int, int, float, void*);
typedef int (*dlpd_zfft_t)(const float*, const float*, void*, int, int, int, long long, int, float, void*);
typedef int (*dlpd_xy_correlate_t)(const void*, const void*, void*, int, int, int, long long, void*);
typedef int (*dlpd_zifft_real_t)(const void*, float*, int, int, int, int, float, void*);

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: client <library.so> host|hip\n"); return 2; }
  on_gpu = strcmp(argv[2], "hip") == 0;
  if (on_gpu) {
    void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hip) { fprintf(stderr, "dlopen libamdhip64.so: %s\n", dlerror()); return 2; }
    p_malloc = (hip_malloc_t)dlsym(hip, "hipMalloc");
    p_memcpy = (hip_memcpy_t)dlsym(hip, "hipMemcpy");
    p_sync = (hip_sync_t)dlsym(hip, "hipDeviceSynchronize");
    p_free = (hip_free_t)dlsym(hip, "hipFree");
    if (!p_malloc || !p_memcpy || !p_sync || !p_free) { fprintf(stderr, "HIP runtime symbols missing\n"); return 2; }
  }
  void* lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror()); return 2; }
  dlpd_version_t dlpd_version_fn; dlpd_grid_supported_t dlpd_grid_supported_fn; dlpd_rotate_trilinear_t dlpd_rotate_trilinear_fn;
  dlpd_topk_workspace_bytes_t dlpd_topk_workspace_bytes_fn; dlpd_topk_select_t dlpd_topk_select_fn;
  dlpd_rfft3d_padded_t dlpd_rfft3d_padded_fn; dlpd_zfft_t dlpd_zfft_fn; dlpd_xy_correlate_t dlpd_xy_correlate_fn;
  dlpd_zifft_real_t dlpd_zifft_real_fn;
  SYM(dlpd_version) SYM(dlpd_grid_supported) SYM(dlpd_rotate_trilinear) SYM(dlpd_topk_workspace_bytes) SYM(dlpd_topk_select)
  SYM(dlpd_rfft3d_padded) SYM(dlpd_zfft) SYM(dlpd_xy_correlate) SYM(dlpd_zifft_real)
  if (dlpd_version_fn() < 100 || dlpd_grid_supported_fn(32) != 1 || dlpd_grid_supported_fn(33) != 0) { fprintf(stderr, "version / grid query\n"); return 1; }

  enum { L = 32, N = 64, NZ = 33 };
  const size_t L3 = (size_t)L * L * L, N3 = (size_t)N * N * N;
  float* h = (float*)malloc(L3 * sizeof(float));
  unsigned s = 12345u;
  for (size_t i = 0; i < L3; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.0f - 0.5f; }

  /* 1. identity rotation */
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float *vol = (float*)dev_alloc(L3 * 4), *out = (float*)dev_alloc(L3 * 4), *R = (float*)dev_alloc(9 * 4);
  to_dev(vol, h, L3 * 4); to_dev(R, I, 36);
  if (dlpd_rotate_trilinear_fn(vol, R, out, 1, 1, L, (long long)L3, L / 2.0f, NULL) != DLPD_OK) { fprintf(stderr, "rotate failed\n"); return 1; }
  float* g = (float*)malloc(L3 * 4);
  to_host(g, out, L3 * 4);
  if (memcmp(g, h, L3 * 4) != 0) { fprintf(stderr, "identity rotation changed the volume\n"); return 1; }
  if (dlpd_rotate_trilinear_fn(NULL, R, out, 1, 1, L, (long long)L3, L / 2.0f, NULL) != DLPD_ERR_ARG) { fprintf(stderr, "null pointer not refused\n"); return 1; }

  /* 2. top-K picks: three planted minima, then the reference's zero-fill quirk: a pick is set to 0.0 in place
   *    (Docker.py:98), so once only zeros are left the FIRST zero in flat order is picked again and again */
  enum { K = 5 };
  float* hv = (float*)calloc(N3, sizeof(float));
  for (size_t i = 0; i < N3; i++) hv[i] = 1.0f + (float)(i % 7);
  hv[1000] = -3.0f; hv[77] = -2.0f; hv[200000] = -1.0f; hv[5] = 0.0f; hv[9] = 0.0f;
  float* V = (float*)dev_alloc(N3 * 4); to_dev(V, hv, N3 * 4);
  float* sc = (float*)dev_alloc(K * 4); int* ix = (int*)dev_alloc(K * 4);
  void* ws = dev_alloc(dlpd_topk_workspace_bytes_fn(1, K));
  if (dlpd_topk_select_fn(V, 1, (long long)N3, K, sc, ix, ws, NULL) != DLPD_OK) { fprintf(stderr, "topk failed\n"); return 1; }
  float hs[K]; int hi[K];
  to_host(hs, sc, sizeof(hs)); to_host(hi, ix, sizeof(hi));
  const int want_i[K] = {1000, 77, 200000, 5, 5};
  const float want_s[K] = {-3.f, -2.f, -1.f, 0.f, 0.f};
  for (int k = 0; k < K; k++)
    if (hi[k] != want_i[k] || hs[k] != want_s[k]) { fprintf(stderr, "pick %d: (%d, %g), expected (%d, %g)\n", k, hi[k], hs[k], want_i[k], want_s[k]); return 1; }

  /* 3. correlation with a one-voxel volume: out[t] = sum_r v1[r + t] v2[r] = v1[t + (1,2,3)] */
  float* hd = (float*)calloc(L3, sizeof(float));
  hd[(1 * L + 2) * L + 3] = 1.0f;
  float* v2 = (float*)dev_alloc(L3 * 4); to_dev(v2, hd, L3 * 4);
  void *wsA = dev_alloc((size_t)NZ * L * L * 8), *spec = dev_alloc((size_t)NZ * N * N * 8), *wsB = dev_alloc((size_t)NZ * N * N * 8);
  float* corr = (float*)dev_alloc(N3 * 4);
  int rc = dlpd_rfft3d_padded_fn(vol, spec, wsA, 1, L, 1.0f / (float)N3, NULL);
  rc |= dlpd_zfft_fn(v2, NULL, wsA, 1, 1, L, 0, 0, 0.0f, NULL);
  rc |= dlpd_xy_correlate_fn(wsA, spec, wsB, 1, 1, L, 0, NULL);
  rc |= dlpd_zifft_real_fn(wsB, corr, 1, 1, L, 0, 0.0f, NULL);
  if (rc != DLPD_OK) { fprintf(stderr, "correlation stages failed (%d)\n", rc); return 1; }
  float* hc = (float*)malloc(N3 * 4);
  to_host(hc, corr, N3 * 4);
  double worst = 0.0;
  for (int tx = -3; tx <= 3; tx++) for (int ty = -3; ty <= 3; ty++) for (int tz = -3; tz <= 3; tz++) {
    const int x = tx + 1, y = ty + 2, z = tz + 3;
    const float want = (x >= 0 && x < L && y >= 0 && y < L && z >= 0 && z < L) ? h[((size_t)x * L + y) * L + z] : 0.0f;
    const float got = hc[(((size_t)((tx + N) % N)) * N + (size_t)((ty + N) % N)) * N + (size_t)((tz + N) % N)];
    const double e = fabs((double)got - (double)want);
    if (e > worst) worst = e;
  }
  if (worst > 1e-5) { fprintf(stderr, "correlation with a one-voxel volume is off by %g\n", worst); return 1; }
  dev_free(vol); dev_free(out); dev_free(R); dev_free(V); dev_free(sc); dev_free(ix); dev_free(ws); dev_free(v2); dev_free(wsA);
  dev_free(spec); dev_free(wsB); dev_free(corr);
  printf("c-abi client ok (%s, version %d, correlation error %.2g)\n", argv[2], dlpd_version_fn(), worst);
  return 0;
}
