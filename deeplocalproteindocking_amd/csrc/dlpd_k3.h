// Declarations shared by the K3 kernels (z-axis C2R + filter MLP + clash mask): dlpd_corr.hip (channel-owning waves,
// barrier-separated phases) and dlpd_k3r.hip (role-split waves).
#pragma once
#include <dlpd_platform.h>
#include "dlpd_fft.h"

// Extra first-layer inputs that are already real volumes on the coarser (N/2) grid, nearest-upsampled by
// index (DockingModels.py:74-76): either the Caux clipped correlations of that resolution (W1t rows
// C..C+Caux-1 are applied here), or -- is_preact -- the HP first-layer pre-activations k_filter_preact
// computed from them once per COARSE voxel (bias included; the first layer is linear): 8x fewer multiply-adds.
struct K3Aux {
  const float* p;   // (nb, Caux or HP, Naux^3), Naux = N/2
  int C, N, is_preact;
};
// Candidate emission for the top-K stage (dlpd_topk.hip, candidate path): every score whose order-preserving key is
// <= *tau is appended to the rotation's list; *tau == 0 means "no valid filter yet" and flags the rotation for the
// full radix select.  tau is written by the merge kernel of earlier batches on another stream: a stale (larger) value
// only lengthens the list.  count: [0, nb) counters, [nb, 2 nb) need-full flags.
struct K3Cand {
  const unsigned* tau;
  unsigned long long* keys;   // (nb, cap)
  unsigned* count;
  int cap, nb;
};
DLPD_D unsigned k3_score_key(float v) {           // == f2key of dlpd_topk.hip
  v = v + 0.0f;
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
DLPD_D void k3_emit(const K3Cand& cd, unsigned tau, int b, unsigned flat, float score) {
  const unsigned key = k3_score_key(score);
  if (key <= tau) {
    const unsigned slot = atomicAdd(&cd.count[b], 1u);
    if (slot < (unsigned)cd.cap) cd.keys[(size_t)b * cd.cap + slot] = ((unsigned long long)key << 32) | flat;
  }
}
