// K3, role-split formulation: z-axis C2R + clip + SimpleFilter MLP + clash mask (gfx950 / CDNA4).
//
// Reference path being replaced (file:line in /root/reference):
//   src/Models/DockingModels.py:70-83   per-channel correlation (its z-inverse), upsample + concat, SimpleFilter MLP
//   src/Docker/Docker.py:226,232        clash threshold, mask multiply
//
// Same inputs, outputs and arithmetic (same products, same summation order) as k_zifft_filter /
// k_zifft_filter_tiles of dlpd_corr.hip.  There EVERY wave alternates between the two halves of the work -- pack +
// wave-local z transform of "its" channel (LDS-bound), then, behind a block barrier, the first-layer multiply-adds of
// its voxels over all channels of the group (VALU-bound) -- so the LDS pipe idles while the vector units work and
// vice versa (rocprofv3, round 2: SQ_WAIT_ANY 30 % of the wave cycles at N = 128, 53 % at N = 160, where in addition
// half of the waves own no channel and sit out every transform phase).  Here the waves of a block have FIXED ROLES:
//   * F transform waves: LDS-DMA of their channel's raw spectra, pack two rows per complex pencil, the two wave-local
//     FFT passes -- nothing else, no accumulators;
//   * M filter waves: own the tile's voxels (4 per thread at the default shapes) and fold the group's channels into
//     their hidden units from REGISTERS: after the "pencils ready" barrier they copy their voxels' values of the
//     group out of the pencils (G x EPT 8-byte LDS reads), a second barrier hands the pencils back, and the
//     multiply-adds of group g then run beside pack + transform of group g + 1.
// On one SIMD a transform wave (LDS latency, address arithmetic) and filter waves (back-to-back packed FMAs) are
// complementary, which is the pairing MI355X_MICROARCH.md ("Two waves per SIMD", item 5) says a rendezvous pays for.
// A block walks all y-tiles of an x' plane, so the pipeline fills once per plane, not once per tile.
// Measured and rejected (round 3): the same two roles paced by flags in LDS instead of block barriers -- every transform
// wave a ring slot of its own (full / done words, polled with s_sleep), no channel groups, 8 + 8 waves at N = 128: 2.19 ms
// against 1.84 ms for this kernel (2.11 with 4 + 8 waves), 2.06 against 2.09 at N = 160.  The per-channel hand-off costs
// the filter waves more (poll + copy + atomic per channel: 11 % + 37 % waiting for "full") than the extra transform
// waves give back, and at 16 waves the 128-register ceiling spills the radix-16 / radix-20 passes.  Wave priorities
// (s_setprio 1..3 for the transform waves) change nothing.  THREE roles at N = 128 -- first-pass waves, second-pass waves and
// filter waves pipelined over three pencil buffers with one barrier per group (4 + 4 + 8 waves, 128 registers, the filter
// waves reading their values from LDS as in the channel-owning kernel): bit-identical, 2.13 ms against 1.82.
// At N = 160, where the filter waves are the longer role (prologue = 48 global loads of coarse pre-activations per thread, 15 %
// of their time): those planes staged in LDS one tile ahead by the transform waves' LDS-DMA (30 KB, waited for before the
// tile's last hand-back barrier): bit-identical, 2.21 ms against 2.08 -- the extra wait in front of the barrier costs the
// filter waves more than the shorter prologue saves.  Round 4: the same planes CHANNELS-LAST (pre[b][x][y][z][HP]: twelve
// 16-byte loads per thread instead of 48 4-byte ones at plane stride; git show 9fa2243): bit-identical, K3 1.97 against
// 1.98 ms -- the 4-byte loads were coalesced 128-byte requests already -- and the coarse kernel that writes them 1.375
// against 1.30 (16-byte stores at a 96-byte lane stride); removed.  The filter waves' first layer on the MATRIX pipe
// (v_mfma_f32_4x4x1_16b_f32: each lane its own voxel times the four weights of its lane quad, one exact fmaf per
// accumulator: bit-identical, registers as here; weight quads from an LDS table): K3 2.17 against 2.10 ms (real shapes),
// 2.03 / 1.81 (48 ch x 64^3), 4.95 / 4.93 (48 ch x 80^3) -- the 4x4x1 form issues no faster than the vector FMAs of one
// wave and serialises the two filter waves of a SIMD; with the weights from global memory 2.56 / 2.21 / 6.0; removed.
#include <dlpd_platform.h>
#include "dlpd_fft.h"
#include "dlpd_internal.h"
#include "dlpd_k3.h"

template <int N> DLPD_D void init_twiddles_k3r(cplx* tw, int tid, int nthreads) {
  for (int k = tid; k < N; k += nthreads) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    tw[k] = c_make((float)c, (float)s);
  }
}

// F transform waves, M filter waves, TY rows per tile (16: a transform wave owns one channel = 8 two-row pencils;
// 8: two channels of 4 pencils -- 64-byte DMA runs: only where two voxels per filter thread leave no room for 16 rows),
// RAWBUF raw staging buffers per transform wave (2: the next group's DMA is issued before this group is packed), PBUF
// pencil buffers (2: the filter waves read their values straight from the pencils while the transform waves fill the
// other buffer -- "TWO PENCIL BUFFERS" in the kernel)
#ifndef DLPD_K3R_F128
#define DLPD_K3R_F128 4
#endif
#ifndef DLPD_K3R_M160
#define DLPD_K3R_M160 10
#endif
#ifndef DLPD_K3R_TY160
#define DLPD_K3R_TY160 16
#endif
#ifndef DLPD_K3R_PBUF160W
#define DLPD_K3R_PBUF160W 2
#endif
#ifndef DLPD_K3R_DMA_BEHIND_LOADS
#define DLPD_K3R_DMA_BEHIND_LOADS 1
#endif
#ifndef DLPD_K3R_TWREG128
#define DLPD_K3R_TWREG128 1
#endif
#ifndef DLPD_K3R_PBUF128W
#define DLPD_K3R_PBUF128W 2
#endif
#ifndef DLPD_K3R_PBUF128
#define DLPD_K3R_PBUF128 1
#endif
#ifndef DLPD_K3R_PBUF160
#define DLPD_K3R_PBUF160 2
#endif
#ifndef DLPD_K3R_RAWBUF128
#define DLPD_K3R_RAWBUF128 1
#endif
#ifndef DLPD_K3R_FFT_PRIO
#define DLPD_K3R_FFT_PRIO 0
#endif
// WIDE: hidden widths 33..48 (the reference class default: multiplier 16 -> [32, 64] channels -> hidden 48,
// ProteinRepresentationModels.py:24,35-36): 96 accumulators are two voxels x 48 hidden units, so the filter waves
// take two voxels per thread and the tile shrinks to 8 rows where 16 rows would need 16 filter waves.
// (round 4: the WIDE configurations run two pencil buffers as well -- N = 128, 48 channels, width 48: K3 3.86-3.94 -> 3.81-3.83 ms;
// N = 160: K3 3.40 -> 3.05 ms at width 48 on the real
// shapes' channels, no spill left; 2.19 / 2.20 at width 32 --, on 8-row tiles still: 16 rows x 160 voxels x 48 accumulators are
// 480 KB of the CU's 512 KB of registers)
// Hidden widths above this take two voxels per filter thread (the WIDE configurations below).  Width 32 on four voxels
// is 128 accumulators: at N = 128 that spills 9 registers and still beats the two-voxel blocks (K3 2.03 vs 2.56 ms at
// 48 ch x 64^3); at N = 160 it spills 35 and loses (2.92 vs 2.38 ms on the real shapes) -- measured in round 4.
template <int N> struct K3rWideAbove { static constexpr int value = (N == 160) ? 24 : 32; };
template <int N, bool WIDE> struct K3rCfg;
template <> struct K3rCfg<64, false> { static constexpr int F = 4, M = 4, TY = 16, RAWBUF = 2, PBUF = 1; };
template <> struct K3rCfg<80, false> { static constexpr int F = 5, M = 5, TY = 16, RAWBUF = 2, PBUF = 1; };
template <> struct K3rCfg<128, false> { static constexpr int F = DLPD_K3R_F128, M = 8, TY = 16, RAWBUF = DLPD_K3R_RAWBUF128, PBUF = DLPD_K3R_PBUF128; };
template <> struct K3rCfg<160, false> { static constexpr int F = 5, M = DLPD_K3R_M160, TY = DLPD_K3R_TY160, RAWBUF = 1, PBUF = DLPD_K3R_PBUF160; };
template <> struct K3rCfg<80, true> { static constexpr int F = 5, M = 10, TY = 16, RAWBUF = 1, PBUF = 1; };
template <> struct K3rCfg<128, true> { static constexpr int F = 4, M = 8, TY = 8, RAWBUF = 1, PBUF = DLPD_K3R_PBUF128W; };
template <> struct K3rCfg<160, true> { static constexpr int F = 5, M = 10, TY = 8, RAWBUF = 1, PBUF = DLPD_K3R_PBUF160W; };

#ifdef DLPD_STAMPS
__device__ unsigned long long dlpd_stamps_k3r[32];
extern "C" int dlpd_debug_read_stamps_k3r(unsigned long long* host32) {
  if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(dlpd_stamps_k3r), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[32] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(dlpd_stamps_k3r), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

// Raw staging layout (per channel): slot NPAIR*k + msl, k = 0 .. N/2, holds {A[k].re, A[k].im, B[k].re, B[k].im} of row
// pair m = msl ^ k3r_pair_swz(k).  The swizzle (8 pairs per bin only) makes the 16-byte reads of the first pass --
// lane = 8*pencil + t reads bin t + 8r of its pencil -- bank-conflict free: a ds_read_b128 is served in groups of 16
// lanes {0-3,12-15,20-27}, ..., and 8k + m alone puts lanes t, t+2 of one pencil on the same 16-byte column.
// WHERE K3's LDS BANK CONFLICTS COME FROM (round 6; SQ_LDS_BANK_CONFLICT = 15 % of its LDS-active cycles): every access the
// source writes is conflict free by the bank model, but the compiler turns the two special-cased first-pass reads of a
// pencil (bins 0 and N/2: `dc ? (x, z) : (x - w, y + z)`) into ds_read2_b32 pairs -- x, z for every lane, y, w under the branch
// of the general case -- and the 32-lane groups of ds_read2_b32 put four lanes on a bank where the 16-lane groups of the
// ds_read_b128 it replaces are conflict free by the pair swizzle: 4 instructions x 12 extra cycles = 48 per channel and
// tile, which IS the counter (3.85e7 per launch / 16,384 tiles / 49 channels).  DLPD_K3R_DC_OPAQUE = 1 keeps the two loads
// whole (16 ds_read_b128, no ds_read2_b32, conflict counter at zero, 143 instead of 163 registers) -- and measured SLOWER,
// same box, alternating libraries, twice: K3 1.750 / 1.762 against 1.710 / 1.713 ms with a volatile keep-alive, 1.752 / 1.744
// against 1.707 / 1.706 with a plain one (EXPERIMENTS.md R6): the kernel is vector-issue bound and the conflicts sit in its
// slack.  Default 0: the compiler's form stays.
#ifndef DLPD_K3R_DC_OPAQUE
#define DLPD_K3R_DC_OPAQUE 0
#endif
// ROUND 6, the first filter layer and the matrix core (EXPERIMENTS.md R6; the code: git show 7504e3b).  With the layer's
// multiply-adds reduced to one hidden unit (a timing probe, wrong results) K3<128, 24> takes 3.11 instead of 3.66 ms per 32
// rotations: the transform waves alone set 85 % of the kernel's time.  The layer as chains of v_mfma_f32_16x16x4_f32 -- a
// k-ascending fmaf chain bit for bit (scripts/micro/mfma_f32_order.hip), so the SAME scores and list hash -- was built (filter
// wave = one row pair, lane (k, n) = channel k of the group at 16 z, hidden units 16..23 of both rows in one tile through a
// v_permlane32_swap of the B registers, the second layer as matrix chains as well, 162 registers) and measured 4.78 against
// 3.61 ms (4.57 with every operand of the role in an LDS table).  Why: scripts/micro/mfma_f32_rate.hip -- beside a stream of
// f32-input matrix instructions a vector wave on the same SIMD issues one packed FMA per 22 cycles instead of one per 5: the
// f32 matrix work does not run BESIDE the transform waves' vector work, it displaces it.  Removed.
// all four components of a loaded float4 needed at ONE point (a plain, non-volatile asm: a data dependence, no ordering
// against the kernel's other inline assembly)
#if defined(DLPD_CPU_EMU)
#define DLPD_K3R_KEEP4(q) ((void)0)
#else
#define DLPD_K3R_KEEP4(q) asm("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w))
#endif
template <int NPAIR> DLPD_HD int k3r_pair_swz(int k) { return NPAIR == 8 ? ((k & 2) << 1) : 0; }

// first pass of the inverse z transform of one pencil (thread t of 8): inputs from the raw channel `rj`, radix-R1
// butterfly in registers (k3r_first_pass_load), outputs to the pencil at S + rowoff in the layout the second pass
// expects (k3r_first_pass_store; fft_wave / fft_wave_pencils of dlpd_fft.h)
template <int N> struct K3rFirst { typedef FftPassW<N, N / 8, 1, +1, 8> Pass; };
template <int N, int NPAIR> DLPD_D void k3r_first_pass_load(typename K3rFirst<N>::Pass& ps, int t, int m, const float4* rj) {
  constexpr int NH = N / 2, R1 = N / 8, RH = R1 / 2;
  static_assert(NH % 8 == 0 && R1 % 2 == 0, "bins t + 8r: the first R1/2 direct, the others mirrored");
  // lane-constant slots; k & 2 == t & 2 for the direct bins, (N/2 - t) & 2 for the mirrored ones
  const float4* lo = rj + NPAIR * t + (m ^ k3r_pair_swz<NPAIR>(t));
  const float4* hi = rj + NPAIR * (NH - t) + (m ^ k3r_pair_swz<NPAIR>(NH - t));
  {
    float4 q[RH];
#pragma unroll
    for (int r = 0; r < RH; r++) q[r] = lo[NPAIR * 8 * r];
    // (DLPD_K3R_DC_OPAQUE, above: keeps the special-cased bin's load ONE 16-byte read; measured slower, off)
    if (DLPD_K3R_DC_OPAQUE) DLPD_K3R_KEEP4(q[0]);
#pragma unroll
    for (int r = 0; r < RH; r++) {
      // k = 0: the purely real bin of both rows
      const bool dc = (r == 0) && (t == 0);
      ps.v[0][r] = dc ? c_make(q[r].x, q[r].z) : c_make(q[r].x - q[r].w, q[r].y + q[r].z);
    }
  }
  {
    float4 q[RH];
#pragma unroll
    for (int s = 0; s < RH; s++) q[s] = *(hi - NPAIR * 8 * s);
    if (DLPD_K3R_DC_OPAQUE) DLPD_K3R_KEEP4(q[0]);
#pragma unroll
    for (int s = 0; s < RH; s++) {
      // k = N/2: purely real as well
      const bool ny = (s == 0) && (t == 0);
      ps.v[0][RH + s] = ny ? c_make(q[s].x, q[s].z) : c_make(q[s].x + q[s].w, q[s].z - q[s].y);
    }
  }
  SmallDft<R1, +1>::run(ps.v[0]);
}
template <int N> DLPD_D void k3r_first_pass_store(cplx* S, int rowoff, int t, const typename K3rFirst<N>::Pass& ps) {
  constexpr int R1 = N / 8;
  cplx* P = S + rowoff;
  if constexpr (N == 160) {
#pragma unroll
    for (int r = 0; r < R1; r++) lds_st(P + 21 * t + r, ps.v[0][r]);        // "blocks of 21" (dlpd_fft.h)
  } else if constexpr (N == 80) {
    const RowMid<R1 + 1> md = {rowoff};
    ps.store_blk(S, md, t);
  } else {
    const RowAddr<0> ad = {rowoff};
    ps.store(S, ad, t);
  }
}
template <int N, int NPAIR> DLPD_D void k3r_first_pass(cplx* S, int rowoff, int t, int m, const float4* rj) {
  typename K3rFirst<N>::Pass ps;
  k3r_first_pass_load<N, NPAIR>(ps, t, m, rj);
  k3r_first_pass_store<N>(S, rowoff, t, ps);
}

// second pass of the transform waves.  K3rTwReg<N>: the pass' twiddles (14 complex values per thread at N = 128) stay in the
// transform waves' registers for the whole kernel instead of being read from the LDS table in every step -- the block's register
// allocation is set by the filter waves' accumulators, the transform waves have room (K3<128> 1.78 -> 1.74 ms; at N = 160 the
// 21 values per thread do not fit next to the radix-20 pass under the 128 registers of 15 waves: 4 spilled, not enabled)
template <int N> struct K3rTwReg { static constexpr bool value = (N == 128) && (DLPD_K3R_TWREG128 != 0); };
template <int N> struct K3rSecond { typedef FftPassW<N, 8, N / 8, +1, 8> Pass; };
template <int N> DLPD_D void k3r_second_pass(cplx* S, int rowoff, int t, const cplx* tw,
                                             const cplx (&twr)[K3rSecond<N>::Pass::PER][7]) {
  constexpr int R1 = N / 8;
  DLPD_WAVE_SYNC();
  if constexpr (K3rTwReg<N>::value) {
    const RowAddr<0> ad = {rowoff};
    typename K3rSecond<N>::Pass ps;
    ps.load_twr(S, ad, t, twr);
    DLPD_WAVE_SYNC();
    ps.store(S, ad, t);
  } else if constexpr (N == 160) {
    cplx* P = S + rowoff;
    FftPassW<N, 8, 20, +1, 8> ps;                    // 20 butterflies of radix 8: three rounds, the last half full
#pragma unroll
    for (int i = 0; i < 3; i++)
      if (t + 8 * i < 20) {
#pragma unroll
        for (int r = 0; r < 8; r++) ps.v[i][r] = lds_ld(P + t + 8 * i + 21 * r);
        ps.twiddle_and_run(i, t + 8 * i, tw);
      }
    DLPD_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < 3; i++)
      if (t + 8 * i < 20) {
#pragma unroll
        for (int r = 0; r < 8; r++) lds_st(P + t + 8 * i + 21 * r, ps.v[i][r]);
      }
  } else {
    const RowAddr<0> ad = {rowoff};
    FftPassW<N, 8, R1, +1, 8> ps;
    if constexpr (N == 80) {
      const RowMid<R1 + 1> md = {rowoff};
      ps.load_blk(S, md, t, tw);
    } else {
      ps.load(S, ad, t, tw);
    }
    DLPD_WAVE_SYNC();
    ps.store(S, ad, t);
  }
}

//   Bw   (nb, CT, NZ, N, N) complex [kz][x'][y']
//   MODE 1: V (nb, N,N,N) = mask * (W2 . relu(W1 . clamp(corr) + b1) + b2); score channels [0,C), clash channel C
//           if has_clash (mask = corr_C < thr); aux: HP first-layer pre-activation planes on the coarse grid (or none)
//   MODE 2: out (nb, HP, N,N,N) = b1 + W1rows^T clamp(corr): the coarse resolution's half of the first layer
//   W1t  (C, HP) transposed + zero padded, b1 (HP), W2 (HP);  G channels per group (<= F * CPW)
template <int N, int HP, int MODE> __global__ void __launch_bounds__(64 * (K3rCfg<N, (HP > K3rWideAbove<N>::value)>::F + K3rCfg<N, (HP > K3rWideAbove<N>::value)>::M))
k_zifft_filter_rs(const cplx* __restrict__ Bw, float* __restrict__ out, int CT, int C, int has_clash, int G,
                  const float* __restrict__ W1t, const float* __restrict__ b1, const float* __restrict__ W2,
                  float b2, int has_clip, float clip, float thr, K3Aux aux, int ntiles, int tpb, K3Cand cd) {
  typedef K3rCfg<N, (HP > K3rWideAbove<N>::value)> Cfg;
  constexpr int F = Cfg::F, M = Cfg::M, TY = Cfg::TY, RAWBUF = Cfg::RAWBUF;
  constexpr int NZ = N / 2 + 1, RS = N + 8, NPAIR = TY / 2, NYT = N / TY;
  constexpr int CPW = 8 / NPAIR;               // channels per transform wave: its 8 pencils = CPW channels x NPAIR row pairs
  constexpr int LPK = 64 / NPAIR;              // kz rows per 64-lane DMA instruction
  constexpr int NFULL = (N / 2) / LPK;         // full 64-lane DMA instructions per channel (bins 0..N/2-1)
  constexpr int GMAX = F * CPW;                // channel slots per group
  constexpr int NTM = 64 * M;                  // filter threads
  constexpr int EPT = (NPAIR * N) / NTM;       // complex values (voxel pairs: rows 2m, 2m+1) per filter thread and channel
  constexpr int MSTEP = NTM / N;               // pair stride between a thread's values
  constexpr int PBUF = Cfg::PBUF;              // pencil buffers (2: see "TWO PENCIL BUFFERS" below)
  constexpr int PSZ = F * 8 * RS;              // complex elements per pencil buffer
  // float4 slots per raw channel: whole waves, or -- where two pencil buffers leave no room -- exactly the channel, the last
  // DMA instruction then running on NPAIR lanes only
  constexpr int RAWC = (PBUF == 2) ? NZ * NPAIR : ((NZ * NPAIR + 63) / 64) * 64;
  static_assert(NPAIR == 8 || NPAIR == 4, "one transform wave = 8 pencils x 8 threads");
  static_assert(EPT >= 1 && EPT * NTM == NPAIR * N && NTM % N == 0, "the filter waves tile the voxels exactly");
  static_assert(NZ * NPAIR == NFULL * 64 + NPAIR, "raw channel = NFULL full DMA instructions + one short one");
  DLPD_DYN_SHARED(cplx, S);
  cplx* tw = S + PBUF * PSZ;
  float4* raw = reinterpret_cast<float4*>(tw + N);        // [RAWBUF][F][CPW][RAWC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_fft = wave < F;
  // N = 160, ROUND 4.  What bounded this kernel was the SHAPE of its reads: 8-row tiles fetch the spectra in 64-byte runs
  // (one (channel, kz) line of a tile), which HBM serves at 3.6 TB/s where 128-byte runs get 6.4-6.9 (scripts/micro/
  // dma_gather.hip); a build with neither transforms nor multiply-adds took 1.92 of the kernel's 2.04 ms.  Hence 16-row
  // tiles: five transform waves x one channel, TEN filter waves x four voxels, 15 waves at 128 registers -- 1.75 ms -- and
  // two pencil buffers (PBUF, below) -- 1.47-1.55 ms.  The experiments of the following paragraph were all measured UNDER
  // the 64-byte bound: their "no gain" says nothing about the kernel as it is now.  Measured on the new kernel and not kept:
  // transform waves at s_setprio 1 / 3 (1.65-1.68 against 1.51-1.54 on that box), the younger filter waves raised (1.60
  // against 1.51), touches of the next tile's pre-activation planes by the transform waves (one LDS-DMA per line into a junk
  // kilobyte: 1.68 with the non-temporal hint, 1.507 against 1.51-1.55 without).  The same two buffers at N = 128: 2.04
  // against 1.87 ms (that kernel is vector-issue bound; PBUF stays a per-box choice).
  // (first half of round 4: TEN half-size filter waves at N = 160 on 8-row tiles -- 15 waves, 128 registers, 3 spilled -- with the transform waves dealt
  // one per SIMD, hardware waves {0, 1, 2, 3, 7}, since a workgroup's waves go to the SIMDs cyclically: K3 2.07-2.09 ms
  // against 2.00-2.03 for 5 + 5 on the real shapes, 5.5 against 4.8 at 48 ch x 80^3; transform waves first: 2.07-2.20 / 5.4;
  // bit-identical; not kept.  Also round 4: FOUR transform + FOUR filter waves at N = 160 -- one wave of each role per SIMD,
  // 256 registers per thread, a filter thread owning five voxels (two row pairs and half of a third: the tile's 1,280
  // voxels over 256 threads; 212 registers, no spill), 8 channels per group -- K3 2.03-2.08 against 2.00-2.02 ms on the
  // real shapes, 4.86-4.95 against 4.59-4.66 at 48 ch x 80^3, bit-identical: balancing the SIMDs buys nothing when each
  // holds only two waves to hide the other's LDS round trips.  Every block shape tried at N = 160 (5 + 5, 5 + 10 in two
  // role maps, 4 + 4) lands within 4 % of 2.0 ms on the real shapes.  So does TWO PENCIL BUFFERS (4 transform + 5 filter
  // waves, 8 channels per group, 136 KB: the transform waves write group s + 1 while the filter waves copy group s, one
  // barrier per group instead of two, the transform waves up to a group ahead): 2.14 against 2.115 ms, 5.01 against 4.88 at
  // 48 ch x 80^3, bit-identical -- the filter waves' wait at "pencils ready" (22 % in the stamps) is not what a decoupled
  // producer removes.)
  const int twave = wave, fwave = wave - F;
  const int t_beg = blockIdx.x * tpb, t_end = (t_beg + tpb < ntiles) ? t_beg + tpb : ntiles;
  if (t_beg >= t_end) return;
  // a block's tiles lie in ONE x' plane (tpb divides the NYT tiles of a plane): plane, x' and rotation are decoded once
  const int plane = t_beg / NYT, plane_x = plane % N, plane_b = plane / N;
  init_twiddles_k3r<N>(tw, tid, 64 * (F + M));
  const unsigned cand_tau = (MODE == 1 && cd.keys) ? *cd.tau : 0u;
  // no clip = a clamp to +-infinity, which returns its argument: one v_med3 per value instead of a v_med3 and a select
  const float clampv = has_clip ? clip : __builtin_inff();

  // ---- transform role: this wave's channels of group `cb` of tile `t` -> raw staging buffer `buf`
  //      raw[k][m] <- Bw[b][cb + wave*CPW + j][k][xo][y0+2m .. +1]   (lane = NPAIR*(k % LPK) + m: LPK runs of NPAIR*16
  //      bytes per DMA instruction; k = N/2 is the last, short one)
  float4* rawg = raw + twave * CPW * RAWC;
  auto issue_channel = [&](int t, int cb, int buf, int lane) {
    const int ty0 = (t - plane * NYT) * TY, txo = plane_x, tb = plane_b;
#pragma unroll
    for (int j = 0; j < CPW; j++) {
      const int g = twave * CPW + j;
      if (g < G && cb + g < CT) {
        // (wave-uniform: the channel's base as a scalar pair, the lane's place in a DMA instruction as one 32-bit byte offset)
        const cplx* src = Bw + (((size_t)tb * CT + cb + DLPD_UNIFORM(g)) * NZ * N + txo) * N + ty0;
        // slot NPAIR*k + msl of the raw channel holds row pair msl ^ k3r_pair_swz(k) of bin k (see k3r_first_pass)
        const unsigned lane_off = (unsigned)(((lane / NPAIR) * N * N + 2 * ((lane % NPAIR) ^ k3r_pair_swz<NPAIR>(lane / NPAIR))) * sizeof(cplx));
        const dlpd_lds_t rj = DLPD_LDS_ADDR(rawg + buf * (F * CPW * RAWC) + j * RAWC);      // LDS address, taken once
#pragma unroll
        for (int it = 0; it < NFULL; it++) DLPD_GLDS16_SOA(src + (size_t)it * LPK * N * N, lane_off, rj + it * 1024);
        const int mt = lane % NPAIR;                        // tail lanes re-read valid elements
        if (PBUF == 1 || lane < NPAIR) DLPD_GLDS16_SOA(src + (size_t)(N / 2) * N * N, 2 * mt * sizeof(cplx), rj + NFULL * 1024);
      }
    }
  };
  // ---- filter role: voxel ownership (rows y0 + 2m, y0 + 2m + 1, column zz)
  const int tm = 64 * fwave + lane;
  const int zz_ = is_fft ? 0 : tm % N, m0_ = is_fft ? 0 : tm / N;

  if (is_fft) {
    if (DLPD_K3R_FFT_PRIO) DLPD_SET_PRIO(DLPD_K3R_FFT_PRIO);
    issue_channel(t_beg, 0, 0, lane);
  }
  DLPD_LDS_BARRIER();                          // twiddle table visible
  DLPD_STAMP_DECL;
  const int ngroups = (CT + G - 1) / G;
  const int nsteps = (t_end - t_beg) * ngroups;
  // The two roles are two separate loops over the same (tile, group) steps -- separate, so that the filter waves'
  // accumulators are not live (and allocated) across the transform code -- that meet at two block barriers per step:
  //   B1  "pencils of this group complete"   (transform waves arrive after their last store has landed)
  //   B2  "pencils copied into registers"    (filter waves arrive after their loads have returned)
  if (is_fft) {
    // ================= transform waves: raw -> pencils of group (t, cbase) =================
    int t = t_beg, cbase = 0, rb = 0;
    cplx* P = S;                               // this step's pencil buffer
    cplx twr[K3rSecond<N>::Pass::PER][7];      // (K3rTwReg: the second pass' twiddles, fetched once)
    if constexpr (K3rTwReg<N>::value) {
      typename K3rSecond<N>::Pass ps0;
      ps0.fetch_twiddles(lane & 7, tw, twr);
    }
#pragma unroll 1
    for (int step = 0; step < nsteps; step++) {
      const int gn = (CT - cbase) < G ? (CT - cbase) : G;
      const bool last_group = cbase + G >= CT;
      const bool mine = twave * CPW < gn;
      // (the lane index re-enters every step as an opaque value, as the voxel coordinates of the filter waves do: the
      // addresses derived from it are recomputed per step instead of being kept -- at 128 registers: spilled -- across it)
      int ln = lane;
      if (F + M > 12) DLPD_OPAQUE_V(ln);
      const int tr = ln & 7, qr = ln >> 3;     // FFT: lane = 8*pencil + thread
      if (mine) {
        DLPD_WAIT_VMEM();                      // this wave's own DMA has landed
        DLPD_WAVE_SYNC();
      }
      DLPD_STAMP(0);
      // the next group of this tile, or the first group of the next tile
      const int nt = last_group ? t + 1 : t, ncb = last_group ? 0 : cbase + G;
      if (RAWBUF == 2 && nt < t_end) issue_channel(nt, ncb, rb ^ 1, ln);
      // FIRST PASS STRAIGHT FROM THE RAW SPECTRA.  A pencil holds two real rows as one complex sequence,
      // Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k]) (k <= N/2; A, B the Hermitian half-spectra of rows
      // 2m, 2m+1).  Thread t of the first pass (radix R1 = N/8, butterfly t) needs Z[t + 8r], r < R1: the first half
      // of them are raw bins t + 8r, the second half the mirrored bins N/2 - t - 8s -- both are 16-byte reads at
      // lane-constant addresses plus immediates, so the separate pack phase (a ds_write_b64 per element and as many
      // VALU as the transform itself) and the first pass' own reads disappear.
      // (a wave whose second channel lies beyond the group transforms stale staging data into pencils nobody reads)
      if (mine) {
        const int m = qr % NPAIR, j = qr / NPAIR;
        k3r_first_pass<N, NPAIR>(P, (twave * 8 + qr) * RS, tr, m, rawg + rb * (F * CPW * RAWC) + j * RAWC);
        DLPD_WAVE_SYNC();                      // every lane's raw values are in registers: the staging buffer is free
      }
      DLPD_STAMP(1);
      if constexpr (K3rTwReg<N>::value && DLPD_K3R_DMA_BEHIND_LOADS) {
        // the second pass' 16 LDS reads first, the next group's DMA (scalar address arithmetic, nine to eleven instructions)
        // behind them while they are in flight, then twiddles, butterflies and stores
        const RowAddr<0> ad = {(twave * 8 + qr) * RS};
        typename K3rSecond<N>::Pass ps;
        DLPD_WAVE_SYNC();
        if (mine) ps.load_only(P, ad, tr);
        if (RAWBUF == 1 && nt < t_end) issue_channel(nt, ncb, 0, ln);
        if (mine) {
          ps.run_twr(tr, twr);
          DLPD_WAVE_SYNC();
          ps.store(P, ad, tr);
        }
      } else {
        if (RAWBUF == 1 && nt < t_end) issue_channel(nt, ncb, 0, ln);
        if (mine) k3r_second_pass<N>(P, (twave * 8 + qr) * RS, tr, tw, twr);
      }
      DLPD_STAMP(2);
      if (RAWBUF == 2) rb ^= 1;
      DLPD_LDS_BARRIER();                      // B1
      DLPD_STAMP(3);
      if (PBUF == 2) {
        // the other buffer: the filter waves finished reading it before they arrived at this barrier
        P = (P == S) ? S + PSZ : S;
      } else {
        DLPD_LDS_BARRIER();                    // B2
      }
      DLPD_STAMP(4);
      if (last_group) { cbase = 0; t++; } else cbase += G;
    }
  } else {
    // ================= filter waves =================
    float nrm[EPT * 2];
    float h[EPT * 2][HP];
    cplx vals[PBUF == 2 ? 1 : GMAX][EPT] = {};
    const cplx* P = S;
    int t = t_beg, cbase = 0;
#pragma unroll 1
    for (int step = 0; step < nsteps; step++) {
      const int y0 = (t - plane * NYT) * TY, xo = plane_x, b = plane_b;
      const int gn = (CT - cbase) < G ? (CT - cbase) : G;
      const bool last_group = cbase + G >= CT;
      // the lane's voxel coordinates re-enter every step as opaque values: what is derived from them (LDS offsets, global
      // offsets) is then recomputed where it is used -- a few vector instructions -- instead of being hoisted out of the
      // step loop and, at the 128 registers of the 15-wave blocks, spilled (35 spilled registers without this at width 24)
      int zz = zz_, m0 = m0_;
      DLPD_OPAQUE_V(zz);
      DLPD_OPAQUE_V(m0);
      if (cbase == 0) {
        // first group of a tile: hidden pre-activations
        if (MODE == 1 && cd.keys && !cand_tau && tm == 0) cd.count[cd.nb + b] = 1u;
#pragma unroll
        for (int e = 0; e < EPT * 2; e++) nrm[e] = 0.f;
        if (MODE == 1 && aux.C > 0) {
          // rows 2m, 2m+1 and columns z, z^1 of the fine grid share one coarse voxel
          // (wave-uniform 64-bit plane bases + one 32-bit lane offset per row pair: scalar address arithmetic, so that
          // no per-plane vector addresses are kept -- or spilled -- across the step loop)
          const int Na = aux.N;
          size_t cstride = (size_t)Na * Na * Na;
          DLPD_OPAQUE_S(cstride);                // (its 24 multiples are not worth 48 scalar registers across the loop either)
          const float* ub = aux.p + (size_t)b * HP * cstride + ((size_t)(xo >> 1) * Na + (y0 >> 1)) * Na;
          unsigned lo[EPT];
#pragma unroll
          for (int e = 0; e < EPT; e++) lo[e] = (unsigned)((m0 + e * MSTEP) * Na + (zz >> 1));
#pragma unroll
          for (int e = 0; e < EPT; e++)
#pragma unroll
            for (int j = 0; j < HP; j++) {
              const float v = (ub + (size_t)j * cstride)[lo[e]];
              h[2 * e][j] = v;
              h[2 * e + 1][j] = v;
            }
        } else {
#pragma unroll
          for (int e = 0; e < EPT * 2; e++)
#pragma unroll
            for (int j = 0; j < HP; j++) h[e][j] = b1[j];
        }
      }
      DLPD_STAMP(0);
      DLPD_LDS_BARRIER();                      // B1: all pencils of the group transformed
      DLPD_STAMP(1);
      // score channels of this group; the clash channel (index C, always last) is peeled off
      const int gs = (cbase + gn <= C) ? gn : (C - cbase > 0 ? C - cbase : 0);
      if (PBUF == 1) {
#pragma unroll
        for (int g = 0; g < GMAX; g++)
          if (g < gs) {
#pragma unroll
            for (int e = 0; e < EPT; e++) vals[g][e] = S[(g * NPAIR + m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
          }
      }
      if (MODE == 1 && has_clash && cbase + gn > C) {
        const int g = C - cbase;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
          const cplx v = P[(g * NPAIR + m0 + e * MSTEP) * RS + pencil_out_pos<N>(zz)];
          nrm[2 * e] = v.x;
          nrm[2 * e + 1] = v.y;
        }
      }
      if (PBUF == 1) {
        DLPD_LDS_BARRIER();                    // B2: values held in registers, pencils free for the next group
        // (the barrier's own wait is inline assembly the compiler does not see: consumed here, the copied values are not
        // "pending" when the channel blocks start -- see "THE FIRST CHANNEL'S OPERANDS ARE CONSUMED HERE" below)
        // (unconditionally: a consumption under `g < gs` leaves the value pending on the other path of the join)
#pragma unroll
        for (int g = 0; g < GMAX; g++) {
#pragma unroll
          for (int e = 0; e < EPT; e++) { DLPD_SINK_V(vals[g][e].x); DLPD_SINK_V(vals[g][e].y); }
        }
      }
      DLPD_STAMP(2);
      if (PBUF == 2) {
        // TWO PENCIL BUFFERS: the values are read where they are used -- this buffer stays untouched until these waves
        // arrive at the next step's barrier (the transform waves work on the other one) -- so no copy phase, no second
        // barrier and no values held in registers (20 of them at 5 channels per group); channel g + 1's values and weights
        // are requested before channel g's multiply-adds
        const cplx* pv = P + (m0 * RS + pencil_out_pos<N>(zz));
        float wcur[HP], wnxt[HP];
        cplx vcur[EPT], vnxt[EPT];
        if (gs > 0) {
#pragma unroll
          for (int j = 0; j < HP; j++) wcur[j] = W1t[(size_t)cbase * HP + j];
#pragma unroll
          for (int e = 0; e < EPT; e++) vcur[e] = pv[e * MSTEP * RS];
          // THE FIRST CHANNEL'S OPERANDS ARE CONSUMED HERE (empty asm): scalar loads return out of order, so any wait on
          // them is a wait for all of them; left pending into the channel blocks, the first row forces a full wait behind
          // EVERY block's requests for the next channel -- the multiply-adds of channel g then start only when channel
          // g + 1's operands have arrived (seen in the ISA: s_load, ds_read, s_waitcnt lgkmcnt(0), 48 packed FMAs).  With
          // nothing pending on entry the compiler waits where the hand-over needs it: behind the multiply-adds.
#pragma unroll
          for (int j = 0; j < HP; j++) DLPD_SINK_S(wcur[j]);
#pragma unroll
          for (int e = 0; e < EPT; e++) { DLPD_SINK_V(vcur[e].x); DLPD_SINK_V(vcur[e].y); }
        }
#pragma unroll
        for (int g = 0; g < GMAX; g++) {
          if (g < gs) {
            const int gn1 = (g + 1 < gs ? g + 1 : g);
#pragma unroll
            for (int j = 0; j < HP; j++) wnxt[j] = W1t[(size_t)(cbase + gn1) * HP + j];
#pragma unroll
            for (int e = 0; e < EPT; e++) vnxt[e] = pv[(gn1 * NPAIR + e * MSTEP) * RS];
            DLPD_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < EPT; e++) {
              float v0 = vcur[e].x, v1 = vcur[e].y;
              v0 = DLPD_CLAMP(v0, clampv);
              v1 = DLPD_CLAMP(v1, clampv);
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
        }
        P = (P == S) ? S + PSZ : S;
      } else {
        // first-layer weights are wave-uniform (scalar loads): channel g+1's row is requested before channel g's FMAs
        // (round 4: the row in two halves for widths >= 32, HP scalar registers for the two buffers instead of 2 HP: the
        // vector spills of <128, 32> / <160, 48> are the 128 / 96 accumulators themselves, 11 / 23 against 9 / 21 -- not kept)
        float wcur[HP], wnxt[HP];
        if (gs > 0) {
#pragma unroll
          for (int j = 0; j < HP; j++) wcur[j] = W1t[(size_t)cbase * HP + j];
#pragma unroll
          for (int j = 0; j < HP; j++) DLPD_SINK_S(wcur[j]);      // (see "THE FIRST CHANNEL'S OPERANDS ARE CONSUMED HERE")
        }
#pragma unroll
        for (int g = 0; g < GMAX; g++) {
          if (g < gs) {
            const int gn1 = (g + 1 < gs ? g + 1 : g);
#pragma unroll
            for (int j = 0; j < HP; j++) wnxt[j] = W1t[(size_t)(cbase + gn1) * HP + j];
            DLPD_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < EPT; e++) {
              float v0 = vals[g][e].x, v1 = vals[g][e].y;
              v0 = DLPD_CLAMP(v0, clampv);
              v1 = DLPD_CLAMP(v1, clampv);
#pragma unroll
              for (int j = 0; j < HP; j++) {
                h[2 * e][j] = fmaf(wcur[j], v0, h[2 * e][j]);
                h[2 * e + 1][j] = fmaf(wcur[j], v1, h[2 * e + 1][j]);
              }
            }
            DLPD_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < HP; j++) wcur[j] = wnxt[j];
          }
        }
      }
      DLPD_STAMP(3);
      if (last_group) {
        if (MODE == 2) {
          // first-layer pre-activations of these channels as HP planes (bias included): the coarse resolution's half
          // of SimpleFilter's first layer, which the fine grid's kernel picks up by index (DockingModels.py:74-83)
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const int m = m0 + e * MSTEP;
#pragma unroll
            for (int u = 0; u < 2; u++) {
              {
#pragma unroll
                for (int j = 0; j < HP; j++)
                  (out + ((((size_t)b * HP + j) * N + xo) * N + y0) * N)[(unsigned)((2 * m + u) * N + zz)] = h[2 * e + u][j];
              }
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const int m = m0 + e * MSTEP;
#pragma unroll
            for (int u = 0; u < 2; u++) {
              float acc = b2;
#pragma unroll
              for (int j = 0; j < HP; j++) acc = fmaf(W2[j], fmaxf(h[2 * e + u][j], 0.f), acc);
              if (has_clash) acc = acc * ((nrm[2 * e + u] < thr) ? 1.0f : 0.0f);
              (out + (((size_t)b * N + xo) * N + y0) * N)[(unsigned)((2 * m + u) * N + zz)] = acc;
              if (cd.keys && cand_tau) k3_emit(cd, cand_tau, b, (unsigned)((xo * N + y0 + 2 * m + u) * N + zz), acc);
            }
          }
        }
        DLPD_STAMP(4);
      }
      if (last_group) { cbase = 0; t++; } else cbase += G;
    }
  }
#ifdef DLPD_STAMPS
  // (-DDLPD_STAMPS=<w>: transform wave w % F and filter wave w % M report)
  if (lane == 0 && ((is_fft && twave == (DLPD_STAMPS) % F) || (!is_fft && fwave == (DLPD_STAMPS) % M))) {
    const int o = is_fft ? 0 : 16;
    for (int i_ = 0; i_ < 8; i_++) atomicAdd(&dlpd_stamps_k3r[o + i_], st_sum[i_]);
    atomicAdd(&dlpd_stamps_k3r[o + 15], 1ull);
  }
#endif
}

#ifndef DLPD_K3R_TPB_DIV
#define DLPD_K3R_TPB_DIV 1                   // tiles per block = (y-tiles of an x' plane) / DIV
#endif
static int k3r_group(int CT, int maxg, bool balanced) {
  if (!balanced) return CT < maxg ? CT : maxg;
  const int ng = (CT + maxg - 1) / maxg;
  return (CT + ng - 1) / ng;
}

#ifndef DLPD_K3R_TPB_DIV
#define DLPD_K3R_TPB_DIV 1                   // tiles per block = (y-tiles of an x' plane) / DIV
#endif
template <int N, int HP, int MODE> static int launch_k3r(const cplx* Bw, float* out, int CT, int C, int has_clash, int nb,
                                                         const float* W1t, const float* b1, const float* W2, float b2,
                                                         int has_clip, float clip, float thr, hipStream_t st, K3Aux aux,
                                                         K3Cand cd) {
  typedef K3rCfg<N, (HP > K3rWideAbove<N>::value)> Cfg;
  constexpr int RS = N + 8, NZ = N / 2 + 1, NPAIR = Cfg::TY / 2, CPW = 8 / NPAIR;
  constexpr int RAWC = (Cfg::PBUF == 2) ? NZ * NPAIR : ((NZ * NPAIR + 63) / 64) * 64;
  const size_t shmem = (size_t)(Cfg::PBUF * Cfg::F * 8 * RS + N) * sizeof(cplx) + (size_t)Cfg::RAWBUF * Cfg::F * CPW * RAWC * 16;
  int rc = dlpd_set_max_dyn_shared((const void*)k_zifft_filter_rs<N, HP, MODE>, shmem);
  if (rc) return rc;
  const int G = k3r_group(CT, Cfg::F * CPW, true);
  const int ntiles = (N / Cfg::TY) * N * nb, tpb = (N / Cfg::TY) / DLPD_K3R_TPB_DIV;
  DLPD_LAUNCH((k_zifft_filter_rs<N, HP, MODE>), dim3((ntiles + tpb - 1) / tpb), dim3(64 * (Cfg::F + Cfg::M)), shmem, st, Bw,
              out, CT, C, has_clash, G, W1t, b1, W2, b2, has_clip, clip, thr, aux, ntiles, tpb, cd);
  return dlpd_check_launch();
}

// hidden widths the role-split kernel is compiled for (96 accumulators: 4 voxels x <= 24..32, or 2 voxels x 48)
int dlpd_k3r_supported(int L, int HP, int mode) {
  if (HP != 2 && HP != 4 && HP != 8 && HP != 16 && HP != 24 && HP != 32 && HP != 48) return 0;
  if (mode == 1) return (L == 64 || L == 80) ? 1 : 0;
  if (mode == 2) return (L == 40) ? 1 : 0;
  return 0;
}

template <int N, int MODE> static int k3r_dispatch(int HP, const cplx* Bw, float* out, int CT, int C, int has_clash, int nb,
                                                   const float* W1t, const float* b1, const float* W2, float b2,
                                                   int has_clip, float clip, float thr, hipStream_t st, K3Aux aux, K3Cand cd) {
  switch (HP) {
    case 2: return launch_k3r<N, 2, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 4: return launch_k3r<N, 4, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 8: return launch_k3r<N, 8, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 16: return launch_k3r<N, 16, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 24: return launch_k3r<N, 24, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 32: return launch_k3r<N, 32, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 48: return launch_k3r<N, 48, MODE>(Bw, out, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// wsB -> V (nb, N^3); aux: HP pre-activation planes of the coarse grid (is_preact) or none
int dlpd_k3r_filter(const cplx* Bw, float* V, int CT, int C, int has_clash, int nb, int L, const float* W1t, int HP,
                    const float* b1, const float* W2, float b2, int has_clip, float clip, float thr, K3Aux aux, K3Cand cd,
                    hipStream_t st) {
  switch (L) {
    case 64: return k3r_dispatch<128, 1>(HP, Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    case 80: return k3r_dispatch<160, 1>(HP, Bw, V, CT, C, has_clash, nb, W1t, b1, W2, b2, has_clip, clip, thr, st, aux, cd);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}

// wsB (nb, C, NZ, N, N) -> pre (nb, HP, N^3): z C2R fused with the (linear) first layer over these C channels
int dlpd_k3r_preact(const cplx* Bw, float* pre, int C, int nb, int L, const float* W1rows, int HP, const float* b1,
                    int has_clip, float clip, hipStream_t st) {
  const K3Aux ax = {nullptr, 0, 0, 0};
  const K3Cand cd = {nullptr, nullptr, nullptr, 0, 0};
  switch (L) {
    case 40: return k3r_dispatch<80, 2>(HP, Bw, pre, C, C, 0, nb, W1rows, b1, b1, 0.f, has_clip, clip, 0.f, st, ax, cd);
    default: return DLPD_ERR_UNSUPPORTED;
  }
}
