// Top-K (minimum) selection over the translation grid and the running global merge.
//
// Reference being replaced: Docker.update_top, /root/reference/src/Docker/Docker.py:86-105 --
// max_conf x { min over z, y, x ; record (x,y,z,V) ; V[x,y,z] = 0 }, append to top_list,
// stable sort by score, truncate.  Equivalent closed form implemented here:
//   * per rotation the picks are the strictly negative voxels among the K smallest, ordered by
//     (value, flat index) [-0.0 == +0.0]; every picked voxel is zeroed, so once the negatives
//     run out the minimum is the first zero in flat order among {original zeros} U {picked
//     voxels} and it is picked again and again with score 0.0 (with no zero at all, the
//     smallest positive once, then that voxel with 0.0) -- the reference's zero-fill quirk,
//     pinned by tests/golden/g3_update_top.npz;
//   * the global list is the K smallest entries ordered by (score, rotation, pick order).
//
// Selection = exact radix select on unique 64-bit keys (order-preserving float key << 32 | idx):
// 3 value digits + 3 index digits with early exit, multi-block histograms, then a compaction
// and a one-block bitonic sort of the K survivors.  No host synchronisation anywhere.
#include <dlpd_platform.h>
#include "dlpd_internal.h"
#ifndef DLPD_TOPK_DIAG
#define DLPD_TOPK_DIAG 0                     // diagnostic builds only (EXPERIMENTS.md R5): kernels of the radix select left out (1 hist, 2 scan, 4 collect, 8 sort)
#endif

typedef unsigned long long u64;

#define TOPK_BINS 2048
#define TOPK_HIST_BLOCKS 64
#define TOPK_HIST_THREADS 256
#define TOPK_LDSK 4096          // up to this many conformations the sorts run in one block's LDS ...
#define TOPK_MAXK 65536         // ... above it in global scratch (same kernels, same results, slower: Docker.py:18 accepts any max_conf)

struct TopkState {           // one per rotation in the batch
  u64 kth;                   // decided high bits of the K-th key, finally the K-th key itself
  unsigned krem;             // rank still to resolve inside the current prefix
  int done;
  unsigned ncand;
  unsigned pad;
  unsigned hist[TOPK_BINS];
};

DLPD_HD unsigned f2key(float v) {
  v = v + 0.0f;                                   // -0.0 -> +0.0
  unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
DLPD_HD float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// digits of the 64-bit key, most significant first: three of the score (32 bits), three of the flat index (32 bits:
// grids up to 2^31 voxels -- any box size the generic correlation path accepts)
#define TOPK_NPASS 6
__device__ const int kShift[TOPK_NPASS] = {53, 42, 32, 22, 11, 0};
__device__ const int kBits[TOPK_NPASS] = {11, 11, 10, 10, 11, 11};

__global__ void __launch_bounds__(256) k_topk_init(TopkState* st, int nb, unsigned K) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < TOPK_BINS; i += blockDim.x) st[b].hist[i] = 0;
  if (threadIdx.x == 0) { st[b].kth = 0; st[b].krem = K; st[b].done = 0; st[b].ncand = 0; }
}

// Histogram of one digit of the keys of rotation b (all voxels in pass 0, afterwards those that share the selected prefix).
// NO LDS ATOMICS: every wave counts into a table of its own, and inside a wave the lanes that hit the same bin are
// grouped with ballots (scalar work per different bin), then the first lane of every group adds the group's size with one plain
// read-modify-write (dlpd_lds_count; few groups per wave: the scores of a rotation share their leading bits).  Round 5 found that the `ds_add_u32` version of
// this kernel, running beside a workgroup that feeds `v_mfma_f32_16x16x32_bf16` from `ds_read_b128` (the plugin's bf16 x 3
// convolution), changed the results of the N = 80 FFT kernels that shared the CU with both (EXPERIMENTS.md R5: 259 of 300
// scorings; 0 of 150 with plain LDS updates) -- lanes 48-63 of their transform waves, low mantissa bits.  The counts are the
// same, so the select is bit-identical; the table costs 32 KB of LDS instead of 8 (this kernel runs only while the running
// list fills, and for callers without candidate lists).
__global__ void __launch_bounds__(TOPK_HIST_THREADS)
k_topk_hist(const float* __restrict__ V, long long nvox, TopkState* st, int pass) {
  const int b = blockIdx.y;
  if (st[b].done) return;
  constexpr int NW = TOPK_HIST_THREADS / 64;
  __shared__ unsigned lh[NW][TOPK_BINS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < NW * TOPK_BINS; i += blockDim.x) (&lh[0][0])[i] = 0;
  __syncthreads();
  const int shift = kShift[pass], bits = kBits[pass];
  const unsigned mask = (1u << bits) - 1u;
  const u64 kth = st[b].kth;
  const int hs = shift + bits;
  const float* v = V + (size_t)b * nvox;
  constexpr int STEP = 4 * TOPK_HIST_THREADS;                       // four consecutive scores per lane and round (one 16-byte load)
  const long long per = ((nvox + gridDim.x - 1) / gridDim.x + STEP - 1) / STEP * STEP;
  const long long beg = (long long)blockIdx.x * per;
  const long long end = beg + per < nvox ? beg + per : nvox;
  const bool vec = (nvox & 3) == 0 && ((size_t)v & 15) == 0;
  unsigned* mine = lh[wave];
  for (long long i0 = beg; i0 < end; i0 += STEP) {                 // every lane of the block runs the same number of rounds
    const long long i = i0 + 4 * (long long)threadIdx.x;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec && i + 3 < end) {
      const float4 q = *reinterpret_cast<const float4*>(v + i);
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) if (i + j < end) x[j] = v[i + j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 key = ((u64)f2key(x[j]) << 32) | (u64)(i + j);
      const bool hit = i + j < end && (pass == 0 || (key >> hs) == (kth >> hs));
      dlpd_lds_count(mine, (unsigned)(key >> shift) & mask, hit, lane);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TOPK_BINS; i += blockDim.x) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) sum += lh[w][i];
    if (sum) atomicAdd(&st[b].hist[i], sum);
  }
}

__global__ void __launch_bounds__(256) k_topk_scan(TopkState* st, int pass) {
  const int b = blockIdx.x;
  if (st[b].done) return;
  __shared__ unsigned part[256];
  __shared__ unsigned sel[3];
  const int tid = threadIdx.x;
  constexpr int PER = TOPK_BINS / 256;
  unsigned s = 0;
  for (int i = 0; i < PER; i++) s += st[b].hist[tid * PER + i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    const unsigned krem = st[b].krem;
    unsigned cum = 0;
    int chunk = 0;
    for (; chunk < 256; chunk++) {
      if (cum + part[chunk] >= krem) break;
      cum += part[chunk];
    }
    if (chunk == 256) chunk = 255;   // cannot happen when K <= nvox
    int bin = chunk * PER;
    unsigned cnt = 0;
    for (; bin < chunk * PER + PER; bin++) {
      cnt = st[b].hist[bin];
      if (cum + cnt >= krem) break;
      cum += cnt;
    }
    if (bin == chunk * PER + PER) bin--;
    const int shift = kShift[pass];
    u64 kth = st[b].kth | ((u64)bin << shift);
    if (cum + cnt == krem || pass == TOPK_NPASS - 1) {
      // everything in this bin (and below) is selected: close the key with all-ones below
      if (pass != TOPK_NPASS - 1) kth |= ((u64)1 << shift) - 1;
      st[b].done = 1;
    } else {
      st[b].krem = krem - cum;
    }
    st[b].kth = kth;
  }
  __syncthreads();
  for (int i = tid; i < TOPK_BINS; i += 256) st[b].hist[i] = 0;
  (void)sel;
}

__global__ void __launch_bounds__(TOPK_HIST_THREADS)
k_topk_collect(const float* __restrict__ V, long long nvox, TopkState* st, u64* __restrict__ cand, int K) {
  const int b = blockIdx.y;
  if (st[b].done == 2) return;                       // served from K3's candidate list
  const u64 kth = st[b].kth;
  const float* v = V + (size_t)b * nvox;
  const long long per = (nvox + gridDim.x - 1) / gridDim.x;
  const long long beg = (long long)blockIdx.x * per;
  const long long end = beg + per < nvox ? beg + per : nvox;
  for (long long i = beg + threadIdx.x; i < end; i += blockDim.x) {
    const u64 key = ((u64)f2key(v[i]) << 32) | (u64)i;
    if (key <= kth) {
      const unsigned slot = atomicAdd(&st[b].ncand, 1u);
      if (slot < (unsigned)K) cand[(size_t)b * K + slot] = key;
    }
  }
}

// in-LDS bitonic sort of n (power of two) 64-bit keys, ascending; all threads of the block call
DLPD_D void bitonic_sort_u64(u64* a, int n, int tid, int nt) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < n; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 x = a[i], y = a[ixj];
          const bool up = ((i & k) == 0);
          if ((x > y) == up) { a[i] = y; a[ixj] = x; }
        }
      }
    }
  }
  __syncthreads();
}

// Block-wide sum / minimum, the same value in every thread, and a block-wide exclusive count -- all WITHOUT LDS atomics
// (round 5: see k_topk_hist): butterflies inside a wave, one plain LDS word per wave, plain reads.  Every thread of
// the block must call them; `scr` = 32 words of LDS.
DLPD_D int block_sum(int v, int* scr, int tid, int nt) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  __syncthreads();
  if ((tid & 63) == 0) scr[tid >> 6] = v;
  __syncthreads();
  int s = 0;
  for (int w = 0; w < (nt + 63) / 64; w++) s += scr[w];
  return s;
}
DLPD_D unsigned block_min(unsigned v, int* scr, int tid, int nt) {
  for (int m = 32; m > 0; m >>= 1) { const unsigned o = __shfl_xor(v, m); v = o < v ? o : v; }
  __syncthreads();
  if ((tid & 63) == 0) scr[tid >> 6] = (int)v;
  __syncthreads();
  unsigned s = 0xffffffffu;
  for (int w = 0; w < (nt + 63) / 64; w++) { const unsigned o = (unsigned)scr[w]; s = o < s ? o : s; }
  return s;
}
// position of this thread among the threads of the block with `yes` (in thread order), and their number
DLPD_D int block_rank(bool yes, int* scr, int tid, int nt, int& total) {
  const unsigned long long m = __ballot(yes);
  const int lane = tid & 63, wave = tid >> 6;
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) scr[wave] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < (nt + 63) / 64; w++) { const int c = scr[w]; base += w < wave ? c : 0; tot += c; }
  total = tot;
  return base + before;
}

// one block per rotation: sort the K survivors, apply the zero-fill quirk, write (score, idx)
template <bool LARGE> __global__ void __launch_bounds__(1024)
k_topk_sort(const float* __restrict__ V, long long nvox, const TopkState* st, const u64* __restrict__ cand, int K,
            float* __restrict__ out_score, int* __restrict__ out_idx, u64* __restrict__ sortbuf, int KPs) {
  __shared__ u64 keys_lds[LARGE ? 1 : TOPK_LDSK];
  u64* keys = LARGE ? sortbuf + (size_t)blockIdx.x * KPs : keys_lds;      // LARGE: K > TOPK_LDSK, sorted in global scratch
  __shared__ int scr[32];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  if (st[b].done == 2) return;                       // served from K3's candidate list
  int KP = 1;
  while (KP < K) KP <<= 1;
  for (int i = tid; i < KP; i += nt) keys[i] = (i < K) ? cand[(size_t)b * K + i] : ~(u64)0;
  bitonic_sort_u64(keys, KP, tid, nt);
  // number of strictly negative entries (canonical key < key(+0.0)) and their smallest index
  int local = 0;
  unsigned lmin = 0xffffffffu;
  for (int i = tid; i < K; i += nt)
    if ((unsigned)(keys[i] >> 32) < 0x80000000u) {
      local++;
      const unsigned id = (unsigned)(keys[i] & 0xffffffffu);
      lmin = id < lmin ? id : lmin;
    }
  const int q = block_sum(local, scr, tid, nt);
  const unsigned pmin_s = block_min(lmin, scr, tid, nt);
  // zero-fill: picked voxels were set to 0.0, so after the negatives the minimum is the first
  // zero in flat order among {original zeros} U {picked voxels}; with no zero at all the
  // smallest positive is picked once and then repeats with 0.0.
  int fill_idx = 0;
  bool first_stored = false;      // position q reports the stored value of an unpicked voxel
  if (q < K) {
    const unsigned kq = (unsigned)(keys[q] >> 32), iq = (unsigned)(keys[q] & 0xffffffffu);
    const bool orig_zero = (kq == 0x80000000u);
    if (orig_zero && (q == 0 || iq < pmin_s)) { fill_idx = (int)iq; first_stored = true; }
    else if (q > 0) { fill_idx = (int)pmin_s; }
    else { fill_idx = (int)iq; first_stored = true; }
  }
  for (int i = tid; i < K; i += nt) {
    float sc;
    int idx;
    if (i < q) {
      sc = key2f((unsigned)(keys[i] >> 32));
      idx = (int)(keys[i] & 0xffffffffu);
    } else {
      idx = fill_idx;
      sc = (i == q && first_stored) ? V[(size_t)b * nvox + idx] : 0.0f;   // stored (maybe -0.0), then +0.0
    }
    out_score[(size_t)b * K + i] = sc;
    out_idx[(size_t)b * K + i] = idx;
  }
}

// ------------------------------------------------------------------------------------------
// Candidate path.  Once the running list is full and its K-th score tau is negative, a pick of any later rotation
// can enter the list only if its score is <= tau (the list is the K smallest (score, rotation, pick) triples; zero-fill
// entries score 0).  K3 therefore appends every voxel with score <= the tau it saw (a possibly older, i.e. larger
// one: tau only decreases) to a per-rotation list, and this kernel turns that list straight into the rotation's picks:
// all voxels of the rotation that score below a candidate are candidates themselves, so a candidate's rank in the
// sorted list IS its pick order.  Rotations whose list is incomplete -- K3 saw no valid tau (need_full), or more than
// `cap` candidates -- go through the radix select as before.  ccount: [0, nb) counters, [nb, 2 nb) need_full flags.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_topk_from_cand(TopkState* st, const u64* __restrict__ cbuf, unsigned* __restrict__ ccount, int nb, int cap, int K,
                 float* __restrict__ out_score, int* __restrict__ out_idx) {
  DLPD_DYN_SHARED(u64, keys);
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const unsigned n = ccount[b], full = ccount[nb + b];
  __syncthreads();
  if (tid == 0) { ccount[b] = 0; ccount[nb + b] = 0; }      // ready for the batch after next
  if (full || n > (unsigned)cap) return;                    // radix select takes this rotation
  int NP = 64;
  while (NP < (int)n) NP <<= 1;
  for (int i = tid; i < NP; i += nt) keys[i] = (i < (int)n) ? cbuf[(size_t)b * cap + i] : ~(u64)0;
  bitonic_sort_u64(keys, NP, tid, nt);
  const float inf = __uint_as_float(0x7f800000u);           // filler: never below tau
  for (int i = tid; i < K; i += nt) {
    const bool real = i < (int)n;
    out_score[(size_t)b * K + i] = real ? key2f((unsigned)(keys[i] >> 32)) : inf;
    out_idx[(size_t)b * K + i] = real ? (int)(keys[i] & 0xffffffffu) : 0;
  }
  if (tid == 0) st[b].done = 2;
}

// ------------------------------------------------------------------------------------------
// running global list.  glist: u64 header[2] {count, unused}, u64 hi[K], u64 lo[K]
//   hi = canonical score key << 32 | rotation ; lo = pick << 32 | negzero << 31 | flat idx
// ------------------------------------------------------------------------------------------
DLPD_D bool pair_gt(u64 ah, u64 al, u64 bh, u64 bl) { return ah > bh || (ah == bh && al > bl); }

DLPD_D void bitonic_sort_pairs(u64* hi, u64* lo, int n, int tid, int nt) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < n; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 xh = hi[i], xl = lo[i], yh = hi[ixj], yl = lo[ixj];
          const bool up = ((i & k) == 0);
          if (pair_gt(xh, xl, yh, yl) == up) { hi[i] = yh; lo[i] = yl; hi[ixj] = xh; lo[ixj] = xl; }
        }
      }
    }
  }
  __syncthreads();
}

// Large lists (pair arrays in global memory): the running list [0, count) is sorted and the pending entries
// [KP, KP + nnew) are few in the steady state -- sort those in LDS (<= TOPK_LDSK of them), give every old and every new
// entry its rank in the union by a binary search in the other run, write the K smallest behind the old list and copy
// them back.  Keys are unique (score, rotation | pick, index), so ranks never collide.  sm: 2 x TOPK_LDSK u64 of LDS.
DLPD_D void rank_merge_pairs(u64* hi, u64* lo, int KP, int count, int nnew, int K, u64* sm, int tid, int nt) {
  u64* nh = sm;
  u64* nl = sm + TOPK_LDSK;
  int NP = 64;
  while (NP < nnew) NP <<= 1;
  for (int i = tid; i < NP; i += nt) {
    nh[i] = (i < nnew) ? hi[KP + i] : ~(u64)0;
    nl[i] = (i < nnew) ? lo[KP + i] : ~(u64)0;
  }
  bitonic_sort_pairs(nh, nl, NP, tid, nt);              // (ends with a block barrier: the pending slots may be overwritten)
  for (int i = tid; i < count; i += nt) {               // old entry i: how many new ones sort before it
    const u64 h = hi[i], l = lo[i];
    int a = 0, b = nnew;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (pair_gt(h, l, nh[m], nl[m])) a = m + 1; else b = m;
    }
    const int pos = i + a;
    if (pos < K) { hi[KP + pos] = h; lo[KP + pos] = l; }
  }
  for (int j = tid; j < nnew; j += nt) {                // new entry j: how many old ones sort before it
    const u64 h = nh[j], l = nl[j];
    int a = 0, b = count;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (pair_gt(h, l, hi[m], lo[m])) a = m + 1; else b = m;
    }
    const int pos = j + a;
    if (pos < K) { hi[KP + pos] = h; lo[KP + pos] = l; }
  }
  __syncthreads();
  const int total = (count + nnew < K) ? count + nnew : K;
  for (int i = tid; i < total; i += nt) { hi[i] = hi[KP + i]; lo[i] = lo[KP + i]; }
  __syncthreads();
}

template <bool LARGE> __global__ void __launch_bounds__(1024)
k_topk_merge(const float* __restrict__ cs, const int* __restrict__ ci, const int* __restrict__ rot_ids, int nb, int K,
             u64* __restrict__ glist, unsigned* __restrict__ tau_out) {
  DLPD_DYN_SHARED(u64, sm);
  const int tid = threadIdx.x, nt = blockDim.x;
  int KP = 1;
  while (KP < K) KP <<= 1;
  const int CAP = 2 * KP;
  u64* hi = LARGE ? glist + 2 + 2 * (size_t)K : sm;      // LARGE: the pair arrays live behind the list (dlpd_topk_glist_bytes)
  u64* lo = hi + CAP;
  __shared__ int scr[32];
  u64* ghi = glist + 2;
  u64* glo = glist + 2 + K;
  int count = (int)glist[0];
  for (int i = tid; i < CAP; i += nt) {
    hi[i] = (i < count) ? ghi[i] : ~(u64)0;
    lo[i] = (i < count) ? glo[i] : ~(u64)0;
  }
  __syncthreads();
  int nnew = 0;                                           // appended, not yet merged (uniform)
  // (score key, rotation) of the current K-th entry: a candidate survives iff its own pair sorts before it --
  // on the score alone when rotations arrive in ascending order (the reference's stable append), on the
  // rotation id too when a caller visits them in another order (DockingEngine.search groups them)
  u64 tau = (count == K) ? hi[K - 1] : ~(u64)0;
  for (int r = 0; r < nb; r++) {
    const u64 rot = (u64)(unsigned)rot_ids[r];
    int nsurv = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
      // count survivors of this rotation against tau
      int local = 0;
      for (int i = tid; i < K; i += nt) {
        const u64 sk = f2key(cs[(size_t)r * K + i]);
        if (((sk << 32) | rot) < tau) local++;
      }
      nsurv = block_sum(local, scr, tid, nt);
      if (nnew + nsurv <= CAP - KP || attempt == 1) break;
      // flush: merge what is pending so the new rotation fits
      if (LARGE && nnew <= TOPK_LDSK) rank_merge_pairs(hi, lo, KP, count, nnew, K, sm, tid, nt);
      else bitonic_sort_pairs(hi, lo, CAP, tid, nt);
      count = (count + nnew < K) ? count + nnew : K;
      nnew = 0;
      for (int i = KP + tid; i < CAP; i += nt) { hi[i] = ~(u64)0; lo[i] = ~(u64)0; }
      for (int i = count + tid; i < KP; i += nt) { hi[i] = ~(u64)0; lo[i] = ~(u64)0; }
      __syncthreads();
      tau = (count == K) ? hi[K - 1] : ~(u64)0;
    }
    if (nsurv == 0) continue;                                // (the usual case once the list has settled: nothing to append)
    for (int i0 = 0; i0 < K; i0 += nt) {                     // (every thread runs every round: block_rank has barriers)
      const int i = i0 + tid;
      const float s = i < K ? cs[(size_t)r * K + i] : 0.f;
      const u64 sk = f2key(s);
      const bool survives = i < K && ((sk << 32) | rot) < tau;
      int added;
      const int slot = KP + nnew + block_rank(survives, scr, tid, nt, added);
      if (survives) {
        const u64 negzero = (__float_as_uint(s) == 0x80000000u) ? 1 : 0;
        hi[slot] = (sk << 32) | rot;
        lo[slot] = ((u64)i << 32) | (negzero << 31) | (u64)(unsigned)ci[(size_t)r * K + i];
      }
      nnew += added;
    }
    __syncthreads();
  }
  if (nnew > 0) {
    if (LARGE && nnew <= TOPK_LDSK) rank_merge_pairs(hi, lo, KP, count, nnew, K, sm, tid, nt);
    else bitonic_sort_pairs(hi, lo, CAP, tid, nt);
    count = (count + nnew < K) ? count + nnew : K;
  }
  __syncthreads();
  for (int i = tid; i < count; i += nt) { ghi[i] = hi[i]; glo[i] = lo[i]; }
  if (tid == 0) {
    glist[0] = (u64)count;
    // score key of the K-th entry for K3's candidate filter: only once the list is full and that score is negative
    // (0 = no filter yet)
    if (tau_out) {
      const unsigned tk = (count == K) ? (unsigned)(hi[K - 1] >> 32) : 0u;
      *tau_out = (tk != 0u && tk < 0x80000000u) ? tk : 0u;
    }
  }
}

extern "C" {

static size_t topk_pow2(int K) {
  size_t KP = 1;
  while (KP < (size_t)K) KP <<= 1;
  return KP;
}
size_t dlpd_topk_workspace_bytes(int nb, int K) {
  const size_t sortbuf = (K > TOPK_LDSK) ? (size_t)nb * topk_pow2(K) * sizeof(u64) : 0;
  return (size_t)nb * sizeof(TopkState) + (size_t)nb * (size_t)K * sizeof(u64) + sortbuf + 256;
}
// header + hi[K] + lo[K] (+ the merge's 2 x 2 KP pair scratch when it does not fit a block's LDS)
size_t dlpd_topk_glist_bytes(int K) {
  return (size_t)(2 + 2 * (size_t)K + (K > TOPK_LDSK ? 4 * topk_pow2(K) : 0)) * sizeof(u64);
}

int dlpd_topk_glist_reset(void* glist, int K, void* stream) {
  if (!glist || K <= 0) return DLPD_ERR_ARG;
  return hipMemsetAsync(glist, 0, dlpd_topk_glist_bytes(K), (hipStream_t)stream) == hipSuccess ? DLPD_OK
                                                                                                : DLPD_ERR_LAUNCH;
}

int dlpd_topk_select_cand(const float* V, int nb, long long nvox, int K, float* out_score, int* out_idx, void* ws,
                          const void* cand_keys, void* cand_count, int cap, void* stream);

// V (nb, nvox) -> per rotation the reference's K picks in pick order: out_score/out_idx (nb, K)
int dlpd_topk_select(const float* V, int nb, long long nvox, int K, float* out_score, int* out_idx, void* ws,
                     void* stream) {
  return dlpd_topk_select_cand(V, nb, nvox, K, out_score, out_idx, ws, nullptr, nullptr, 0, stream);
}

// Same, with the candidate lists K3 filled for this batch (dlpd_zifft_filter_cand): rotations whose list is complete
// skip the radix select.  cand_keys (nb, cap) u64, cand_count 2*nb u32 (counters, then need-full flags); both are
// consumed and the counters reset.
int dlpd_topk_select_cand(const float* V, int nb, long long nvox, int K, float* out_score, int* out_idx, void* ws,
                          const void* cand_keys, void* cand_count, int cap, void* stream) {
  if (!V || !out_score || !out_idx || !ws || nb <= 0 || nvox <= 0 || K <= 0) return DLPD_ERR_ARG;
  if (K > TOPK_MAXK || (long long)K > nvox || nvox >= (1ll << 31)) return DLPD_ERR_UNSUPPORTED;
  if ((cand_keys != nullptr) != (cand_count != nullptr) || (cand_keys && (cap < 64 || cap > 8192))) return DLPD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  TopkState* state = (TopkState*)ws;
  u64* cand = (u64*)((char*)ws + (((size_t)nb * sizeof(TopkState) + 255) / 256) * 256);
  DLPD_LAUNCH(k_topk_init, dim3(nb), dim3(256), 0, st, state, nb, (unsigned)K);
  if (cand_keys) {
    int NP = 64;
    while (NP < cap) NP <<= 1;
    const size_t shmem = (size_t)NP * sizeof(u64);
    int rc = dlpd_set_max_dyn_shared((const void*)k_topk_from_cand, shmem);
    if (rc) return rc;
    DLPD_LAUNCH(k_topk_from_cand, dim3(nb), dim3(256), shmem, st, state, (const u64*)cand_keys, (unsigned*)cand_count, nb,
                cap, K, out_score, out_idx);
  }
  for (int pass = 0; pass < TOPK_NPASS; pass++) {
#if !(DLPD_TOPK_DIAG & 1)
    DLPD_LAUNCH(k_topk_hist, dim3(TOPK_HIST_BLOCKS, nb), dim3(TOPK_HIST_THREADS), 0, st, V, nvox, state, pass);
#endif
#if !(DLPD_TOPK_DIAG & 2)
    DLPD_LAUNCH(k_topk_scan, dim3(nb), dim3(256), 0, st, state, pass);
#endif
  }
#if !(DLPD_TOPK_DIAG & 4)
  DLPD_LAUNCH(k_topk_collect, dim3(TOPK_HIST_BLOCKS, nb), dim3(TOPK_HIST_THREADS), 0, st, V, nvox, state, cand, K);
#endif
  int KP = 1;
  while (KP < K) KP <<= 1;
  u64* sortbuf = cand + (size_t)nb * K;                 // (nb, KP), allocated for K > TOPK_LDSK only
#if !(DLPD_TOPK_DIAG & 8)
  if (K > TOPK_LDSK)
    DLPD_LAUNCH(k_topk_sort<true>, dim3(nb), dim3(1024), 0, st, V, nvox, (const TopkState*)state, (const u64*)cand, K,
                out_score, out_idx, sortbuf, KP);
  else
    DLPD_LAUNCH(k_topk_sort<false>, dim3(nb), dim3(1024), 0, st, V, nvox, (const TopkState*)state, (const u64*)cand, K,
                out_score, out_idx, (u64*)nullptr, 0);
#endif
  (void)sortbuf;
  return dlpd_check_launch();
}

int dlpd_topk_merge_tau(const float* cand_score, const int* cand_idx, const int* rot_ids, int nb, int K, void* glist,
                        void* tau_out, void* stream);

// fold nb rotations' picks (rotation ids rot_ids, ascending) into the running global list
int dlpd_topk_merge(const float* cand_score, const int* cand_idx, const int* rot_ids, int nb, int K, void* glist,
                    void* stream) {
  return dlpd_topk_merge_tau(cand_score, cand_idx, rot_ids, nb, K, glist, nullptr, stream);
}

// Same, also publishing the candidate filter of the K3 kernels of LATER batches: *tau_out (u32) = order-preserving key
// of the list's K-th score once the list is full and that score is negative, else 0 (no filter).
int dlpd_topk_merge_tau(const float* cand_score, const int* cand_idx, const int* rot_ids, int nb, int K, void* glist,
                        void* tau_out, void* stream) {
  if (!cand_score || !cand_idx || !rot_ids || !glist || nb <= 0 || K <= 0) return DLPD_ERR_ARG;
  if (K > TOPK_MAXK) return DLPD_ERR_UNSUPPORTED;
  int KP = 1;
  while (KP < K) KP <<= 1;
  if (K > TOPK_LDSK) {
    const size_t shl = (size_t)2 * TOPK_LDSK * sizeof(u64);
    int rcl = dlpd_set_max_dyn_shared((const void*)k_topk_merge<true>, shl);
    if (rcl) return rcl;
    DLPD_LAUNCH(k_topk_merge<true>, dim3(1), dim3(1024), shl, (hipStream_t)stream, cand_score, cand_idx, rot_ids, nb, K,
                (u64*)glist, (unsigned*)tau_out);
    return dlpd_check_launch();
  }
  const size_t shmem = (size_t)4 * KP * sizeof(u64);
  int rc = dlpd_set_max_dyn_shared((const void*)k_topk_merge<false>, shmem);
  if (rc) return rc;
  DLPD_LAUNCH(k_topk_merge<false>, dim3(1), dim3(1024), shmem, (hipStream_t)stream, cand_score, cand_idx, rot_ids, nb, K,
              (u64*)glist, (unsigned*)tau_out);
  return dlpd_check_launch();
}

}  // extern "C"
