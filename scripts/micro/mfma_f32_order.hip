// In what order, and with how many roundings, does v_mfma_f32_16x16x4_f32 sum its four products?
// (round 6: the question behind moving K3's first filter layer -- today a chain of fmaf over the channels, channel order --
// to the f32-input matrix core: the lists stay bit-identical only if D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))).)
// One wave, random operands of mixed magnitude (so that different orders / roundings differ), the D the hardware returns
// compared bit for bit with candidate formulas evaluated on the host in the same f32 / f64 arithmetic:
//   chain_up    fmaf chain k = 0, 1, 2, 3 starting from C
//   chain_down  fmaf chain k = 3, 2, 1, 0 starting from C
//   pairs       (a0 b0 + a1 b1) + (a2 b2 + a3 b3) + C with f32 roundings
//   exact       C + sum of the four exact products, rounded once (evaluated in long double)
// and the same for a chain of TWO instructions (k = 0..7: is the accumulator rounded between them as a chain would?).
// Also: are f32 denormal inputs / results kept (the vector FMA keeps them in this build)?
//   build: hipcc --offload-arch=gfx950 -O2 mfma_f32_order.hip -o mfma_f32_order        exit code 0 always; read the table
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f4 __attribute__((ext_vector_type(4)));

// A (16 x 8), B (8 x 16), C (16 x 16) row-major in global memory; D1 after k = 0..3, D2 after k = 0..7
__global__ void __launch_bounds__(64) k_mfma(const float* A, const float* B, const float* C, float* D1, float* D2) {
  const int l = threadIdx.x, m = l & 15, q = l >> 4;
  f4 acc;
  for (int j = 0; j < 4; j++) acc[j] = C[(4 * q + j) * 16 + m];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 8 + q], B[q * 16 + m], acc, 0, 0, 0);
  for (int j = 0; j < 4; j++) D1[(4 * q + j) * 16 + m] = acc[j];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 8 + 4 + q], B[(4 + q) * 16 + m], acc, 0, 0, 0);
  for (int j = 0; j < 4; j++) D2[(4 * q + j) * 16 + m] = acc[j];
}

static float rnd(unsigned& s, int spread) {
  s = s * 1664525u + 1013904223u;
  const float u = (float)((s >> 8) & 0xFFFFFF) / 16777216.0f - 0.5f;
  s = s * 1664525u + 1013904223u;
  const int e = (int)((s >> 10) % (unsigned)(2 * spread + 1)) - spread;
  return ldexpf(u, e);
}

int main() {
  float *dA, *dB, *dC, *dD1, *dD2;
  hipMalloc(&dA, 128 * 4); hipMalloc(&dB, 128 * 4); hipMalloc(&dC, 256 * 4); hipMalloc(&dD1, 256 * 4); hipMalloc(&dD2, 256 * 4);
  const char* names[4] = {"chain_up", "chain_down", "pairs", "exact"};
  for (int spread = 0; spread <= 12; spread += 6) {
    long match1[4] = {0, 0, 0, 0}, match2[4] = {0, 0, 0, 0}, total = 0;
    unsigned seed = 12345u + spread;
    for (int trial = 0; trial < 200; trial++) {
      float A[128], B[128], C[256], D1[256], D2[256];
      for (int i = 0; i < 128; i++) { A[i] = rnd(seed, spread); B[i] = rnd(seed, spread); }
      for (int i = 0; i < 256; i++) C[i] = rnd(seed, spread);
      hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
      hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
      k_mfma<<<1, 64>>>(dA, dB, dC, dD1, dD2);
      hipMemcpy(D1, dD1, sizeof(D1), hipMemcpyDeviceToHost); hipMemcpy(D2, dD2, sizeof(D2), hipMemcpyDeviceToHost);
      for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++) {
          float cand1[4], cand2[4];
          for (int half = 0; half < 2; half++) {
            float* cand = half ? cand2 : cand1;
            const int k0 = 4 * half;
            const float c0 = C[r * 16 + c];
            const float start = half ? D1[r * 16 + c] : c0;      // (second instruction: from what the first one returned)
            float up = start, dn = start;
            for (int k = 0; k < 4; k++) up = fmaf(A[r * 8 + k0 + k], B[(k0 + k) * 16 + c], up);
            for (int k = 3; k >= 0; k--) dn = fmaf(A[r * 8 + k0 + k], B[(k0 + k) * 16 + c], dn);
            const float p01 = fmaf(A[r * 8 + k0 + 1], B[(k0 + 1) * 16 + c], A[r * 8 + k0] * B[k0 * 16 + c]);
            const float p23 = fmaf(A[r * 8 + k0 + 3], B[(k0 + 3) * 16 + c], A[r * 8 + k0 + 2] * B[(k0 + 2) * 16 + c]);
            const float prs = (p01 + p23) + start;
            long double ex = (long double)start;
            for (int k = 0; k < 4; k++) ex += (long double)A[r * 8 + k0 + k] * (long double)B[(k0 + k) * 16 + c];
            cand[0] = up; cand[1] = dn; cand[2] = prs; cand[3] = (float)ex;
          }
          total++;
          for (int f = 0; f < 4; f++) {
            match1[f] += memcmp(&cand1[f], &D1[r * 16 + c], 4) == 0;
            match2[f] += memcmp(&cand2[f], &D2[r * 16 + c], 4) == 0;
          }
        }
    }
    printf("exponent spread +-%d, %ld outputs:\n", spread, total);
    for (int f = 0; f < 4; f++)
      printf("  %-10s one instruction %6.2f %%   two chained instructions %6.2f %%\n", names[f], 100.0 * match1[f] / total, 100.0 * match2[f] / total);
  }
  // denormals: a product that is denormal, an input that is denormal, an accumulator that is denormal
  {
    float A[128] = {0}, B[128] = {0}, C[256] = {0}, D1[256], D2[256];
    A[0 * 8 + 0] = 1.0e-20f; B[0 * 16 + 0] = 1.0e-20f;                  // D[0][0]: product 1e-40 (denormal result)
    A[1 * 8 + 0] = 1.0e-40f; B[0 * 16 + 1] = 1.0f;                      // D[1][1]: denormal input times 1   (B[0][1])
    C[2 * 16 + 2] = 1.0e-40f;                                           // D[2][2]: denormal accumulator passed through
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dC, dD1, dD2);
    hipMemcpy(D1, dD1, sizeof(D1), hipMemcpyDeviceToHost);
    printf("denormals: product 1e-20 * 1e-20 -> %g (fmaf: %g); input 1e-40 * 1 -> %g; accumulator 1e-40 + 0 -> %g\n", D1[0], fmaf(1.0e-20f, 1.0e-20f, 0.f),
           D1[1 * 16 + 1], D1[2 * 16 + 2]);
  }
  return 0;
}
