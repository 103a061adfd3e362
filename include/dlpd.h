/* dlpd.h -- C ABI of libdlpd.so: MI355X (gfx950) kernels for the exhaustive rotation x translation
 * correlation search of DeepLocalProteinDocking.
 *
 * The reference has no FFI; its boundary for this path is Python duck typing onto
 * TorchProteinLibrary operator objects.  Each entry point below names the
 * reference call site (file:line under /root/reference) whose arithmetic it replaces.  All
 * functions: plain pointers and sizes, device pointers are borrowed for the duration of the call,
 * work is enqueued on `stream` (a hipStream_t passed as void*), no allocation, no host
 * synchronisation, return 0 on success (DLPD_ERR_* otherwise), never throw.
 *
 * Layouts: volumes are (.., L, L, L) float32, index [x][y][z], z contiguous.  N = 2L,
 * NZ = N/2 + 1.  Spectra are (.., NZ, N, N) complex64 indexed [kz][kx][ky].
 *
 * THE PATH -- the minimal call set of one search (Docker.py:184-238; what DockingEngine issues in steady state):
 *   once per pair    dlpd_rfft3d_padded        receptor spectrum            (DockingModels.py:71, hoisted out of the loop)
 *                    dlpd_receptor_pack        ... in K2's read order       (boxes 80 / 40 only)
 *                    dlpd_make_channels_last   ligand copy K1 gathers from
 *   per launch       dlpd_zfft_channels_last   K1: rotation + z transform  (Docker.py:218)
 *                    dlpd_project_atoms + dlpd_zfft_into   clash channel from rotated atoms (Docker.py:221-224)
 *                    dlpd_xy_correlate_packed / dlpd_xy_correlate   K2     (DockingModels.py:70-71, Docker.py:225)
 *                    dlpd_zifft_filter_cand    K3: z inverse + clip + MLP + clash mask + candidates (DockingModels.py:74-83, Docker.py:226-232)
 *                    dlpd_topk_select_cand, dlpd_topk_merge_tau   the ranked list (Docker.py:86-105)
 *   two resolutions  the same K1 / K2 on the coarse grid, then dlpd_zifft_preact (coarse) and dlpd_zifft_filter_cand(aux = its planes)
 * Everything else in this header is a VARIANT of one of these stages -- suffix _ext (embedded box), _oriented / _quads (launch
 * variants of the per-channel K1), _form (kernel formulation named: test cross-checks), _occ (occupancy maps: sparse ligands),
 * _aux / un-suffixed (fewer features) -- a stand-alone operator of the plugin surface (dlpd_rotate_trilinear,
 * dlpd_correlate_generic, dlpd_filter_*, dlpd_conv3d*, dlpd_maxpool3d_5s2*), or a size / capability query.  Test hooks
 * (dlpd_debug_*) are declared in dlpd_debug.h, not here.
 */
#ifndef DLPD_H
#define DLPD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DLPD_OK 0
#define DLPD_ERR_ARG 1
#define DLPD_ERR_UNSUPPORTED 2
#define DLPD_ERR_LAUNCH 3

int dlpd_version(void);
/* sha256 (64 hex digits) of the sources, headers and flags this library was built from; the build
 * script refuses a library whose hash differs from the sources next to it */
const char* dlpd_source_hash(void);
/* 1 if box size L has a compiled pipeline: L in {32, 40, 64, 80} (grids N = 2L = 64, 80, 128, 160) */
int dlpd_grid_supported(int L);
/* 1 if dlpd_xy_correlate_oriented(transposed = 1) exists for box L (slabs the per-channel K1 stored transposed): every
 * compiled box except 80, whose transposed reader is a test variant (-DDLPD_TEST_VARIANTS), not part of libdlpd.so. */
int dlpd_orientation_supported(int L);
/* hidden width the filter kernel pads H to (2,4,8,16,24,32), -1 if H > 32 */
int dlpd_hidden_pad(int H);
/* Hidden width the FUSED pipeline (dlpd_zifft_filter*, dlpd_zifft_preact) pads H to on a fine grid of L^3 voxels
 * (two_res: with a coarse (L/2)^3 grid, ProteinRepresentationModels.py:72-76): dlpd_hidden_pad's widths plus 48 -- the
 * reference class default SimpleFilter([32, 64]) has hidden width 48 (DockingModels.py:25-27) -- where the role-split
 * K3 is compiled (L = 64, 80; coarse 40); -1: no fused kernel for this width, use dlpd_filter_mask. */
int dlpd_fused_hidden_pad(int H, int L, int two_res);

/* TPL VolumeRotation call, src/Docker/Docker.py:218 (ctor :40; DockingModels.py:49).
 * out[b,c](i) = trilinear vol[b,c]( center + R_b^T (i - center) ), zeros outside.
 * R_b is used as a GENERAL 3x3 map by every rotation entry point of this header (nothing assumes orthonormality): a
 * sampling scale s and a reversed axis order (matrix axis 0 <-> last spatial index) -- conventions TorchProteinLibrary's
 * VolumeRotation may have (its source is absent) -- are passed as R' = s P R P, P the axis reversal
 * (deeplocalproteindocking_amd/Utils/Conventions.py: kernel_matrices), the pivot as `center`.
 * vol (B,C,L^3) with batch stride vol_bstride floats (0: one volume set shared by all b). */
int dlpd_rotate_trilinear(const float* vol, const float* R, float* out, int B, int C, int L,
                          long long vol_bstride, float center, void* stream);

/* Stage K1 of TPL VolumeConvolution (src/Models/DockingModels.py:71, src/Docker/Docker.py:225),
 * optionally fused with the rotation of Docker.py:218: z-axis R2C of (rotated) volumes.
 * wsA: nb*CT*NZ*L*L complex64. */
int dlpd_zfft(const float* vol, const float* R, void* wsA, int nb, int CT, int L, long long vol_bstride,
              int do_rotate, float center, void* stream);

/* Same, writing channels [c_base, c_base+CT) of a (nb, CT_out, NZ, L, L) workspace: lets the clash
 * channel come from per-rotation re-projected atoms (Docker.py:221-224) while the score channels are
 * rotated volumes (Docker.py:218). */
int dlpd_zfft_into(const float* vol, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                   long long vol_bstride, int do_rotate, float center, void* stream);
/* dlpd_zfft_into for GIVEN volumes (no rotation; Docker.dockE3's per-batch representation, Docker.py:163-172) that come with
 * occupancy maps: occ (nb, ceil(L/4)^3) bytes, one map per batch entry for all its CT channels, non-zero where the 4 x 4 x 4
 * cell holds a non-zero value (the maps dlpd_conv3d_split_sparse / dlpd_maxpool3d_5s2_sparse hand on).  Voxels of empty
 * cells are taken as zero and NOT read (unwritten activations), x-planes without an occupied cell are written as zeros
 * without a transform -- or, with skip_empty != 0, not written at all: for a consumer that goes by the pencil map (the maps'
 * OR over z; dlpd_xy_correlate_packed_occ).  Same spectra as dlpd_zfft_into on the dense tensor. */
int dlpd_zfft_volumes_occ(const float* vol, const unsigned char* occ, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                          long long vol_bstride, int skip_empty, void* stream);

/* Slab orientation (speed only, results identical).  With transposed = 1 every rotation of the call is
 * processed with the roles of x and y exchanged and its slabs are stored as [kz][y][x]: the caller groups the
 * rotations whose sample matrix has |R[0][2]| > |R[1][2]| (source z axis closer to the output x axis) into such
 * calls, so that the gather of Docker.py:218 runs along contiguous memory for them as well.  Channels written
 * by separate calls (e.g. the re-projected clash volume) must use the same value.
 * dlpd_xy_correlate_oriented undoes the transposition while it stages each slab. */
int dlpd_zfft_oriented(const float* vol, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                       long long vol_bstride, int do_rotate, float center, int transposed, void* stream);
int dlpd_xy_correlate_oriented(const void* wsA, const void* rec, void* wsB, int nb, int CT, int L,
                               long long rec_bstride, int transposed, void* stream);

/* Quad layout of a volume set for the rotation gather: quads[v][x][y][z] (y, z < L-1) = {vol(x,y,z), vol(x,y,z+1),
 * vol(x,y+1,z), vol(x,y+1,z+1)}: the eight trilinear corners are two 16-byte gathers instead of four 8-byte
 * ones (the gather is bound by cache lines touched per instruction).  dlpd_zfft_quads = dlpd_zfft_oriented with
 * do_rotate = 1 on one volume set shared by all rotations; results are bit-identical to the plain path. */
size_t dlpd_quads_floats(int nvol, int L);
int dlpd_make_quads(const float* vol, float* quads, int nvol, int L, void* stream);
int dlpd_zfft_quads(const float* quads, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                    float center, int transposed, void* stream);

/* Channels-last copy of a ligand's C score volumes for the rotation gather of Docker.py:218: cl[x][y][z][Cp], Cp = C
 * rounded up to 16 (zero padded).  Every channel is sampled at the same rotated position, so one 16-byte load
 * serves four channels of a corner and the gather's cache-line requests per sample stop depending on the rotation
 * (no slab orientation, no quad layout on this path).  dlpd_zfft_channels_last = dlpd_zfft_into with do_rotate = 1
 * for channels [c_base, c_base + C) of a (nb, CT_out, NZ, L, L) workspace; samples are bit-identical to the plain
 * path.  A clash channel is written by a separate dlpd_zfft_into call (transposed = 0). */
size_t dlpd_channels_last_floats(int C, int L);
int dlpd_make_channels_last(const float* vol, float* cl, int C, int L, void* stream);
int dlpd_zfft_channels_last(const float* cl, const float* R, void* wsA, int nb, int C, int CT_out, int c_base, int L,
                            float center, void* stream);

/* The same two rotation + z-transform entries for a volume that is an extent^3 box in the corner of the L^3 one (zeros
 * around it): a box_size without a compiled plan inside the next compiled box (box_size is a free argument of
 * Docker.py:18; Docker._dock_volumes_embedded).  `center` is the pivot of the SMALL box; output voxels with any index
 * >= extent are written as zero -- the reference crops the rotated volume to its own box (Docker.py:218).
 * extent = 0 or L: the plain entries above. */
int dlpd_zfft_oriented_ext(const float* vol, const float* R, void* wsA, int nb, int CT, int CT_out, int c_base, int L,
                           long long vol_bstride, int do_rotate, float center, int transposed, int extent, void* stream);
int dlpd_zfft_channels_last_ext(const float* cl, const float* R, void* wsA, int nb, int C, int CT_out, int c_base, int L,
                                float center, int extent, void* stream);
/* ... with the kernel formulation named: 0 = the library's default, 1 = every wave gathers, transforms and stores in turn
 * (k_rotate_zfft_cl), 2 = gather waves and transform / store waves with fixed roles, one block per CU walking a range of
 * work items (k_rotate_zfft_cl_rs).  Same samples, same butterflies: the spectra are bit-identical.  Form 2 is a TEST VARIANT:
 * it exists in -DDLPD_TEST_VARIANTS builds only (tests/variants/libdlpd_variants.so, boxes 64 and 80); libdlpd.so returns
 * DLPD_ERR_UNSUPPORTED for it at every box -- dlpd_k1_form_supported(L, form) says what the loaded library holds. */
int dlpd_k1_form_supported(int L, int form);
int dlpd_zfft_channels_last_form(const float* cl, const float* R, void* wsA, int nb, int C, int CT_out, int c_base, int L,
                                 float center, int extent, int form, void* stream);
/* The same gather with per-rotation OCCUPANCY MAPS (round 6; Docker.py:218 for a ligand whose representation is zero away
 * from the protein -- any real one).  dlpd_rotated_occupancy turns the stored ligand's cell map (occ_src, ceil(L/4)^3 bytes,
 * dlpd_conv3d_tile_occupancy over all its channels) into one conservative map per rotation (occ_out, nb maps: 0 = every
 * trilinear sample of that 4 x 4 x 4 cell of the rotated volume is certainly zero); R are the matrices K1 samples with.
 * dlpd_zfft_channels_last_occ is dlpd_zfft_channels_last_ext that skips the loads of samples in empty cells and the
 * transform of blocks without an occupied cell (their zeros are written).  Same spectra.
 * pencil_out (may be null; nb x ceil(L/4) 32-bit words, [rotation][x cell]): bit (y cell) set where some z cell of that column
 * is marked; dlpd_pencil_bits makes the same words from nb given cell maps (the plugin's own, dlpd_zfft_volumes_occ).
 * skip_empty != 0: blocks without an occupied cell write NOTHING -- for a consumer that goes by the pencil map and never reads
 * those pencils: dlpd_xy_correlate_packed_occ (boxes with a packed receptor: dlpd_pencil_map_supported(L)), which takes the
 * pencils the map marks empty as zeros for channels [0, nmasked) (the clash channel behind them is read as it is). */
int dlpd_rotated_occupancy(const unsigned char* occ_src, const float* R, unsigned char* occ_out, unsigned* pencil_out, int nb,
                           int L, float center, void* stream);
int dlpd_pencil_bits(const unsigned char* occ, unsigned* pencil_out, int nb, int L, void* stream);
int dlpd_zfft_channels_last_occ(const float* cl, const float* R, const unsigned char* occ, void* wsA, int nb, int C, int CT_out,
                                int c_base, int L, float center, int extent, int skip_empty, void* stream);
int dlpd_pencil_map_supported(int L);
int dlpd_xy_correlate_packed_occ(const void* wsA, const void* rec_packed, void* wsB, int nb, int CT, int L,
                                 const unsigned* pencil_map, int nmasked, void* stream);

/* CoordsRotate + CoordsTranslate + TypedCoords2Volume (+ channel sum) of src/Docker/Docker.py:204,
 * 208,221-224 in one kernel: p' = R_b p + shift, density exp(-|r - p'|^2 / 2) on the 5^3 voxels
 * around each atom (build-defined shape).  coords (B, 3*stride_atoms) ordered by type,
 * num_atoms_of_type / offsets (B, ntypes) int32.  out (B, ntypes, L^3) or (B, 1, L^3) if sum_types. */
int dlpd_project_atoms(const float* coords, const int* num_atoms_of_type, const int* offsets, const float* R,
                       float shift_x, float shift_y, float shift_z, float* out, int B, int stride_atoms,
                       int ntypes, int L, float resolution, int sum_types, void* stream);
/* The same with the density shape as parameters (TypedCoords2Volume's is not known: TorchProteinLibrary is absent from
 * the reference tree; scripts/calibrate_tpl.py determines them where it is installed): exp(-|r - p'|^2 / (2 sigma^2)) on
 * the (2 window + 1)^3 voxels around each atom, times `norm` (0 < norm <= 64; the fixed-point accumulator holds
 * 1023 per voxel), voxel (i,j,k) at ((i,j,k) + voxel_offset) * resolution; window <= 6.
 * dlpd_project_atoms = (sigma 1, window 2, voxel_offset 0, norm 1). */
int dlpd_project_atoms_ext(const float* coords, const int* num_atoms_of_type, const int* offsets, const float* R,
                           float shift_x, float shift_y, float shift_z, float* out, int B, int stride_atoms,
                           int ntypes, int L, float resolution, int sum_types, float sigma, int window,
                           float voxel_offset, float norm, void* stream);
/* The same projection (all types, no sum) CELL-WISE, for consumers that go by occupancy maps (Docker.dockE3's per-batch
 * projection, Docker.py:163-165, in front of dlpd_conv3d_split_sparse(unwritten != 0)): only the 4 x 4 x 4 cells an atom's
 * window reaches are cleared, accumulated into and converted; occ (B, ceil(L/4)^3 bytes, written in full) marks them; out
 * (B, ntypes, L^3) holds the same values as dlpd_project_atoms_ext in marked cells and is NOT WRITTEN elsewhere. */
int dlpd_project_atoms_cells(const float* coords, const int* num_atoms_of_type, const int* offsets, const float* R,
                             float shift_x, float shift_y, float shift_z, float* out, unsigned char* occ, int B, int stride_atoms,
                             int ntypes, int L, float resolution, float sigma, int window, float voxel_offset, float norm,
                             void* stream);

/* Zero-padded 3-D R2C spectrum (receptor side of VolumeConvolution, DockingModels.py:71):
 * spec (nvol, NZ, N, N) = scale * rfftn(pad(vol)).  wsA: nvol*NZ*L*L complex64 scratch. */
int dlpd_rfft3d_padded(const float* vol, void* spec, void* wsA, int nvol, int L, float scale, void* stream);

/* Stage K2: per (b,c,kz) slab 2-D FFT, multiply rec * conj(lig), 2-D inverse.
 * rec (CT spectra, or nb*CT with rec_bstride = CT*NZ*N*N); wsB: nb*CT*NZ*N*N complex64. */
int dlpd_xy_correlate(const void* wsA, const void* rec, void* wsB, int nb, int CT, int L,
                      long long rec_bstride, void* stream);

/* The same stage for ONE receptor shared by the whole batch (the search: Docker.py:213-232 scores every rotation of the
 * ligand against the same receptor) on the boxes whose K2 re-reads the receptor spectrum for every rotation (80 and 40:
 * their slab does not stay in registers): dlpd_receptor_pack re-orders the spectrum ONCE into the order K2's column
 * phase consumes it (contiguous kilobytes per instruction instead of 64- / 32-byte runs), dlpd_xy_correlate_packed reads
 * that copy.  Same arithmetic, same results bit for bit.  dlpd_receptor_packed_floats: floats of the packed copy (as
 * many as the spectrum), 0 for boxes that take the natural layout only. */
long long dlpd_receptor_packed_floats(int CT, int L);
int dlpd_receptor_pack(const void* rec, void* packed, int CT, int L, void* stream);
int dlpd_xy_correlate_packed(const void* wsA, const void* rec_packed, void* wsB, int nb, int CT, int L, void* stream);

/* VolumeConvolution (Docker.py:32,225; DockingModels.py:48,71) for ANY box size -- `box_size` is a free constructor
 * argument of the reference's Docker (Docker.py:18,22-24,31): out (nvol, N^3)[t mod N] = sum_r v1[r + t] v2[r], N = 2L,
 * optional clamp to +-clip.  Direct O(n N) transforms, no radix plan: the slow path for boxes without a compiled
 * pipeline (dlpd_grid_supported(L) == 0), L <= 128.  ws: dlpd_correlate_generic_ws_bytes(nvol, L) bytes of scratch. */
int dlpd_generic_box_supported(int L);
size_t dlpd_correlate_generic_ws_bytes(int nvol, int L);
int dlpd_correlate_generic(const float* v1, const float* v2, float* out, int nvol, int L, int has_clip, float clip,
                           void* ws, void* stream);

/* Stage K3, plain: real correlation volumes out (nb, CT, N^3) [+ clamp to +-clip]
 * (output of VolumeConvolution(clip), DockingModels.py:48,71). */
int dlpd_zifft_real(const void* wsB, float* out, int nb, int CT, int L, int has_clip, float clip,
                    void* stream);

/* Stage K3, fused scoring: z C2R + clip + SimpleFilter MLP (DockingModels.py:28-32,79-83) +
 * clash mask and multiply (Docker.py:226,232).  W1t (C,HP) transposed zero-padded first layer,
 * b1 (HP), W2 (HP).  V (nb, N^3). */
int dlpd_zifft_filter(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                      const float* b1, const float* W2, float b2, int HP, int has_clip, float clip,
                      float thr, void* stream);

/* Same for the reference's two-resolution layout (ProteinRepresentationModels.py:72-76): the Caux
 * channels of the coarser resolution arrive as clipped real correlation volumes aux (nb, Caux, L^3)
 * (grid N/2 = L) and are nearest-upsampled by index (DockingModels.py:74-76); W1t has C + Caux rows.
 * aux_is_preact: aux is (nb, HP, L^3) from dlpd_filter_preact -- the coarse half of the (linear) first
 * layer, bias included, evaluated once per coarse voxel -- and seeds the hidden units instead. */
int dlpd_zifft_filter_aux(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                          const float* b1, const float* W2, float b2, int HP, int has_clip, float clip,
                          float thr, const float* aux, int Caux, int aux_is_preact, void* stream);

/* One batch of the hot loop, Docker.py:211-232 (single-resolution model): K1 + K2 + K3. */
int dlpd_score_rotations(const float* lig, const void* recF, const float* R, int nb, int C, int has_clash,
                         int L, float center, const float* W1t, const float* b1, const float* W2, float b2,
                         int HP, int has_clip, float clip, float thr, void* wsA, void* wsB, float* V,
                         void* stream);

/* dlpd_score_rotations with oriented slabs (see dlpd_zfft_oriented). */
int dlpd_score_rotations_oriented(const float* lig, const void* recF, const float* R, int nb, int C, int has_clash,
                                  int L, float center, const float* W1t, const float* b1, const float* W2, float b2,
                                  int HP, int has_clip, float clip, float thr, void* wsA, void* wsB, float* V,
                                  int transposed, void* stream);

/* Per-voxel filter over materialised correlation volumes incl. nearest upsample of a second,
 * coarser resolution (DockingModels.py:74-83) and mask multiply (Docker.py:232). */
int dlpd_filter_mask(const float* conv0, int C0, int N0, const float* conv1, int C1, int N1,
                     const float* mask_norm, float thr, int has_clash, const float* W1t, const float* b1,
                     const float* W2, float b2, int H, float* V, int nb, void* stream);

/* Same with strided layouts and weights padded to HP = dlpd_hidden_pad(H): 4 voxels per thread,
 * float4 traffic.  conv0 (nb, *, N0^3) uses its first C0 channels (batch stride conv0_bstride);
 * mask_norm has batch stride mask_bstride (so it may be a channel of conv0).
 * conv1_is_preact: conv1 is (nb, HP, N1^3) from dlpd_filter_preact instead of (nb, C1, N1^3). */
int dlpd_filter_volumes(const float* conv0, int C0, long long conv0_bstride, int N0, const float* conv1, int C1,
                        int N1, int conv1_is_preact, const float* mask_norm, long long mask_bstride, float thr, int has_clash,
                        const float* W1t, const float* b1, const float* W2, float b2, int HP, float* V, int nb,
                        void* stream);

/* The coarse-resolution half of SimpleFilter's first layer (DockingModels.py:28, after the concat of
 * :77) evaluated on the coarse grid: pre (nb, HP, N1^3) = b1 + W1rows^T conv1, W1rows (C1, HP) = the
 * rows of W1t that belong to the coarse channels.  The first layer is linear, so upsampling these
 * HP planes by index equals applying it to the upsampled channels (8x fewer multiply-adds). */
int dlpd_filter_preact(const float* conv1, int C1, int N1, const float* W1rows, const float* b1, int HP,
                       float* pre, int nb, void* stream);

/* dlpd_zifft_filter_aux that also feeds the candidate lists of the top-K stage (dlpd_topk_select_cand): every score
 * whose order-preserving key is <= *tau is appended to the rotation's list cand_keys (nb, cap) u64 (key << 32 | flat
 * index), counted in cand_count[0..nb); *tau == 0 ("no valid filter yet", see dlpd_topk_merge_tau) flags the rotation
 * in cand_count[nb..2nb) for the full select instead.  tau / cand_keys / cand_count null: plain dlpd_zifft_filter_aux.
 * Replaces nothing in the reference: it is Docker.update_top's pick loop (Docker.py:89-98) seen from the producer. */
int dlpd_zifft_filter_cand(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                           const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                           const float* aux, int Caux, int aux_is_preact, const void* tau, void* cand_keys,
                           void* cand_count, int cap, void* stream);

/* dlpd_zifft_filter_cand with the kernel formulation named (same arithmetic, bit-identical V): form 0 = the
 * library's default, 1 = every wave owns a channel of the group and the transform / filter phases alternate behind
 * block barriers, 2 = role-split blocks -- dedicated transform waves (LDS-DMA, pack, z C2R) and filter waves (the
 * MLP of DockingModels.py:79-83 from registers) that overlap each other; falls back to 1 where 2 is not compiled.
 * libdlpd.so holds ONE formulation per box -- 1 at boxes 32 / 40, 2 at boxes 64 / 80; form 1 at boxes 64 / 80 (and raw
 * aux channels there, aux_is_preact = 0) is DLPD_ERR_UNSUPPORTED unless the library was built with -DDLPD_TEST_VARIANTS
 * (tests/variants: the bit-exactness reference of the tests). */
int dlpd_zifft_filter_form(const void* wsB, float* V, int nb, int C, int has_clash, int L, const float* W1t,
                           const float* b1, const float* W2, float b2, int HP, int has_clip, float clip, float thr,
                           const float* aux, int Caux, int aux_is_preact, const void* tau, void* cand_keys,
                           void* cand_count, int cap, int form, void* stream);

/* Stage K3 for the COARSER resolution of a two-resolution model: z C2R + clip fused with that resolution's half of
 * SimpleFilter's first layer (DockingModels.py:28 after the concat of :77, linear): pre (nb, HP, N^3) = b1 +
 * W1rows^T clamp(corr), W1rows (C, HP) = the rows of W1t that belong to these channels.  The fine grid's
 * dlpd_zifft_filter_aux(aux = pre, aux_is_preact = 1) picks the planes up by index (nearest upsample, :74-76);
 * the C real correlation volumes of this resolution are never written. */
int dlpd_zifft_preact(const void* wsB, float* pre, int nb, int C, int L, const float* W1rows, const float* b1, int HP,
                      int has_clip, float clip, void* stream);
/* dlpd_zifft_preact with the kernel formulation named (see dlpd_zifft_filter_form). */
int dlpd_zifft_preact_form(const void* wsB, float* pre, int nb, int C, int L, const float* W1rows, const float* b1,
                           int HP, int has_clip, float clip, int form, void* stream);

/* dlpd_zifft_real with the clamp restricted to channels [0, nclip). */
int dlpd_zifft_real_part(const void* wsB, float* out, int nb, int CT, int nclip, int L, int has_clip, float clip,
                         void* stream);

/* Representation-plugin convolution (ProteinRepresentationModels.py:85-114, the per-batch cost of
 * Docker.dockE3, Docker.py:166-167): y (B, cout, D^3) = [relu] conv3d(x (B, cin, D^3), w (cout, cin, ks^3)),
 * padding ks/2, stride 1, no bias, exact f32 on the matrix cores.  Supported: ks in {3,5}, cout a multiple
 * of 16, D <= 80 (dlpd_conv3d_supported); anything else is the plugin's own torch convolution. */
int dlpd_conv3d_supported(int cin, int cout, int ks, int D);
/* weights are passed packed: wp = dlpd_conv3d_pack(w (cout, cin, ks^3)), dlpd_conv3d_packed_floats() floats */
size_t dlpd_conv3d_packed_floats(int cin, int cout, int ks);
int dlpd_conv3d_pack(const float* w, float* wp, int cin, int cout, int ks, void* stream);
int dlpd_conv3d(const float* x, const float* wp, float* y, int B, int cin, int cout, int D, int ks, int relu,
                void* stream);
/* Same with stride 1 or 2 (padding ks/2): the stride-2 layer of SE3MultiResReprScalar
 * (ProteinRepresentationModels.py:51).  y (B, cout, Do^3), Do = (D - 1) / stride + 1. */
int dlpd_conv3d_strided(const float* x, const float* wp, float* y, int B, int cin, int cout, int D, int ks, int relu,
                        int stride, void* stream);

/* The same convolution on the BF16 matrix cores with f32-grade results: inputs and weights are split into three bf16
 * terms each and every product is summed from six bf16 products in f32 (the three dropped ones are below f32's own
 * rounding): 16 / 6 = 2.7 x the matrix throughput of the exact-f32 form, results equal to it to a few 1e-7 relative.
 * Weights are passed packed AND split: wp = dlpd_conv3d_split_pack(w (cout, cin, ks^3)), dlpd_conv3d_split_packed_bytes()
 * bytes.  Same supported shapes as dlpd_conv3d (dlpd_conv3d_supported); stride 1 or 2.  Replaces the same reference
 * lines (ProteinRepresentationModels.py:85-114, Docker.py:166-167). */
size_t dlpd_conv3d_split_packed_bytes(int cin, int cout, int ks);
int dlpd_conv3d_split_pack(const float* w, void* wp, int cin, int cout, int ks, void* stream);
int dlpd_conv3d_split(const float* x, const void* wp, float* y, int B, int cin, int cout, int D, int ks, int relu,
                      int stride, void* stream);
/* Tile occupancy: the plugins' convolutions have no bias (ProteinRepresentationModels.py:85-96: bias=False), so an output
 * tile whose receptive field is all zero is zero, and a protein fills a fraction of its box.  occ (B, ceil(D/4), ceil(D/4),
 * ceil(D/4)) bytes, non-zero where the 4 x 4 x 4 cell of the volume holds a non-zero value in some channel:
 * dlpd_conv3d_tile_occupancy computes it for any (B, cin, D^3) tensor; dlpd_conv3d_split_sparse skips the 4 x 4 x 16 output
 * tiles whose neighbouring input cells (a superset of the halo) are empty (occ_in; null = dense) and writes the map of ITS output (occ_out; null = none; not
 * written for stride 2) -- the next layer's occ_in.  Results are bit-identical to dlpd_conv3d_split.
 * unwritten != 0 (needs both maps, stride 1): the activations travel WITH their maps -- an output tile whose neighbourhood is
 * empty is not written at all (its cells stay 0 in occ_out) and no voxel of a cell that occ_in marks empty is read (it is
 * taken as the zero the map stands for): y holds the same values as before in every cell occ_out marks and is undefined
 * elsewhere; every consumer must go by the map (the next layer with unwritten != 0, dlpd_maxpool3d_5s2_sparse,
 * dlpd_zfft_volumes_occ).  Docker.dockE3's per-batch representation (Docker.py:163-167) spent most of its time writing the
 * zeros of skipped tiles. */
size_t dlpd_conv3d_tile_occupancy_bytes(int B, int D);
int dlpd_conv3d_tile_occupancy(const float* x, unsigned char* occ, int B, int cin, int D, void* stream);
int dlpd_conv3d_split_sparse(const float* x, const void* wp, float* y, const unsigned char* occ_in, unsigned char* occ_out,
                             int B, int cin, int cout, int D, int ks, int relu, int stride, int unwritten, void* stream);

/* MaxPool3d(kernel 5, stride 2, padding 2) of the E3 plugin (ProteinRepresentationModels.py:101):
 * x (nvol, D^3) -> y (nvol, Do^3), Do = (D - 1) / 2 + 1. */
int dlpd_maxpool3d_5s2(const float* x, float* y, int nvol, int D, void* stream);
/* The same with occupancy maps (dlpd_conv3d_tile_occupancy): x (B, C, D^3); occ_in (the input's cells; null = read everything):
 * output tiles whose inputs lie in empty cells are written as +0.0 without reading them; occ_out (null = none): the cells of
 * the (B, C, Do^3) output that hold a non-zero value in some channel -- the next convolution's occ_in.
 * unwritten != 0 (needs occ_in): as dlpd_conv3d_split_sparse -- input voxels of empty cells are not read (zero), output tiles
 * over empty cells are not written. */
int dlpd_maxpool3d_5s2_sparse(const float* x, float* y, const unsigned char* occ_in, unsigned char* occ_out, int B, int C, int D,
                              int unwritten, void* stream);

/* Docker.update_top pick loop, src/Docker/Docker.py:89-98: per rotation the K picks in pick order
 * (incl. the zero-fill behaviour).  V (nb, nvox); out (nb, K). */
size_t dlpd_topk_workspace_bytes(int nb, int K);
int dlpd_topk_select(const float* V, int nb, long long nvox, int K, float* out_score, int* out_idx,
                     void* ws, void* stream);

/* dlpd_topk_select with the candidate lists K3 filled for this batch (dlpd_zifft_filter_cand).  Once the running list
 * is full and its K-th score negative, a later pick can enter it only with a score <= that K-th score, and every
 * voxel of a rotation scoring below a candidate is a candidate too, so a candidate's rank in its sorted list IS its
 * pick order: rotations with a complete list (flag clear, count <= cap) skip the radix select over V; the others take
 * it as before.  Consumes the lists and resets cand_count.  cap in [64, 8192]. */
int dlpd_topk_select_cand(const float* V, int nb, long long nvox, int K, float* out_score, int* out_idx, void* ws,
                          const void* cand_keys, void* cand_count, int cap, void* stream);

/* Docker.update_top list maintenance, src/Docker/Docker.py:100-105: append, stable sort by score,
 * truncate -- on a device-resident list.  glist: u64 count, u64 pad, u64 hi[K], u64 lo[K] with
 * hi = score_key<<32 | rotation, lo = pick<<32 | negzero<<31 | flat_index. */
size_t dlpd_topk_glist_bytes(int K);
int dlpd_topk_glist_reset(void* glist, int K, void* stream);
int dlpd_topk_merge(const float* cand_score, const int* cand_idx, const int* rot_ids, int nb, int K,
                    void* glist, void* stream);
/* Same, publishing the candidate filter for the K3 launches of later batches: *tau_out (u32) = order-preserving key of
 * the list's K-th score once the list holds K entries and that score is negative, else 0. */
int dlpd_topk_merge_tau(const float* cand_score, const int* cand_idx, const int* rot_ids, int nb, int K,
                        void* glist, void* tau_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
