"""Host-side driver of the HIP pipeline: owns the device workspaces (as torch tensors) and
enqueues the C-ABI calls of include/dlpd.h on torch's current stream.

It mirrors the body of the reference's hot loop, src/Docker/Docker.py:211-236 (rotate ligand
volumes, clash correlation + threshold, GlobalDockingModel.forward, mask multiply, update_top)
for a single-resolution representation, without any host synchronisation inside the loop.
torch is plumbing only (memory + streams); all arithmetic is in libdlpd.so.
"""
import numpy as np
import torch

from ._lib import get_lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(device):
    if device.type == "cuda":
        return torch.cuda.current_stream(device).cuda_stream
    return 0


def key_to_float(key):
    """Inverse of the order-preserving float key used in the device top-K list."""
    key = np.asarray(key, dtype=np.uint32)
    neg = (key & np.uint32(0x80000000)) == 0
    bits = np.where(neg, ~key, key & np.uint32(0x7FFFFFFF)).astype(np.uint32)
    return bits.view(np.float32)


class DeviceTopList:
    """Device-resident running top list: the state Docker.update_top keeps in ``self.top_list``
    (Docker.py:100-105), plus the per-rotation pick buffers (Docker.py:89-98)."""

    def __init__(self, K, batch, device, lib):
        self.K, self.batch, self.device, self.lib = int(K), int(batch), torch.device(device), lib
        dev = self.device
        self.cand_score = torch.empty(batch, self.K, dtype=torch.float32, device=dev)
        self.cand_idx = torch.empty(batch, self.K, dtype=torch.int32, device=dev)
        self.ws = torch.empty(lib.call("dlpd_topk_workspace_bytes", batch, self.K), dtype=torch.uint8, device=dev)
        self.glist = torch.zeros(lib.call("dlpd_topk_glist_bytes", self.K) // 8, dtype=torch.int64, device=dev)
        # candidate filter of the scoring kernel (include/dlpd.h, dlpd_topk_merge_tau): key of the list's K-th score
        # once the list is full and that score negative, else 0
        self.tau = torch.zeros(4, dtype=torch.int32, device=dev)

    CAND_CAP = 4096

    def new_candidate_set(self):
        """Per-batch candidate lists the scoring kernel fills (dlpd_zifft_filter_cand) and select() consumes."""
        return {"keys": torch.empty(self.batch * self.CAND_CAP, dtype=torch.int64, device=self.device),
                "count": torch.zeros(2 * self.batch, dtype=torch.int32, device=self.device), "cap": self.CAND_CAP}

    def reset(self):
        self.lib.call("dlpd_topk_glist_reset", _ptr(self.glist), self.K, _stream(self.device))
        self.tau.zero_()

    def select(self, V, nb, cset=None):
        """V (nb, nvox) contiguous -> (scores, flat indices) (nb, K) in the reference's pick order.
        cset: the candidate set the scoring kernel filled for exactly these nb rotations (or None).

        Contract WITH a cset (the steady-state path of ``DockingEngine.step``): a rotation whose candidate list is
        valid comes back with only the picks that can still enter the running list -- the scores <= tau (the K-th
        score published by the last merge), in pick order, the rest of its row padded with +inf / index 0 -- which is
        exactly what ``merge()`` needs and NOT the reference's full K picks (Docker.py:89-98); rotations without a valid
        list (list not yet full, tau >= 0, overflow) get the full radix select.  A cset is SINGLE-USE: the select
        consumes and resets its counters, so it must be refilled by the scoring kernel before the next select.  Callers
        that want the reference's K picks per rotation pass ``cset=None``."""
        nvox = V[0].numel()
        if cset is None:
            self.lib.call("dlpd_topk_select", _ptr(V), nb, nvox, self.K, _ptr(self.cand_score), _ptr(self.cand_idx),
                          _ptr(self.ws), _stream(self.device))
        else:
            self.lib.call("dlpd_topk_select_cand", _ptr(V), nb, nvox, self.K, _ptr(self.cand_score), _ptr(self.cand_idx),
                          _ptr(self.ws), _ptr(cset["keys"]), _ptr(cset["count"]), cset["cap"], _stream(self.device))
        return self.cand_score[:nb], self.cand_idx[:nb]

    def merge(self, rot_ids, nb):
        """rot_ids int32 (nb,) on the device, ascending."""
        self.lib.call("dlpd_topk_merge_tau", _ptr(self.cand_score), _ptr(self.cand_idx), _ptr(rot_ids), nb, self.K,
                      _ptr(self.glist), _ptr(self.tau), _stream(self.device))

    def entries(self, glist=None):
        """-> (rot, flat_idx, score, pick) numpy arrays sorted as the reference's top_list."""
        g = (self.glist if glist is None else glist).cpu().numpy().view(np.uint64)
        count = int(g[0])
        hi = g[2:2 + count]
        lo = g[2 + self.K:2 + self.K + count]
        score = key_to_float((hi >> np.uint64(32)).astype(np.uint32)).copy()
        negzero = ((lo >> np.uint64(31)) & np.uint64(1)).astype(bool)
        score[negzero] = np.float32(-0.0)
        rot = (hi & np.uint64(0xFFFFFFFF)).astype(np.int64)
        idx = (lo & np.uint64(0x7FFFFFFF)).astype(np.int64)
        pick = (lo >> np.uint64(32)).astype(np.int64)
        return rot, idx, score, pick

    @staticmethod
    def merge_entries(parts, K):
        """Deterministic merge of several ranks' lists: sort by (score, rotation, pick), keep K.
        Exact because the global top-K is a subset of the union of per-shard top-Ks and the key
        reproduces the reference's stable insertion order."""
        rot = np.concatenate([p[0] for p in parts])
        idx = np.concatenate([p[1] for p in parts])
        score = np.concatenate([p[2] for p in parts])
        pick = np.concatenate([p[3] for p in parts])
        order = np.lexsort((pick, rot, score + np.float32(0.0)))[:K]
        return rot[order], idx[order], score[order], pick[order]

    @staticmethod
    def to_top_list(entries, N):
        rot, idx, score, _ = entries
        x, y, z = idx // (N * N), (idx // N) % N, idx % N
        return [(int(rot[i]), int(x[i]), int(y[i]), int(z[i]), float(score[i])) for i in range(len(rot))]


class DockingEngine:
    """Exhaustive translation scoring for batches of rotations of ONE receptor/ligand pair.

    Parameters follow the reference objects: ``W1,b1,W2,b2`` are SimpleFilter.fc[0]/fc[2]
    (DockingModels.py:28-32), ``clip`` the VolumeConvolution(clip) of DockingModels.py:48,
    ``threshold_clash`` Docker.py:226, ``max_conf`` Docker.py:26.
    """

    def __init__(self, L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=300.0, has_clash=True,
                 max_conf=1000, batch=8, device="cuda", lib=None, center=None, coarse_channels=0,
                 fine_unfused=None, channels_last=None, k3_form=0, coarse_center=None, extent=None,
                 orient=True, quads=True, prefilter=True, packed_receptor=True,
                 rotation_scale=1.0, coarse_rotation_scale=None, rotation_axis_order="xyz", clip_mode="output",
                 rotation_transpose=False, keep_receptor_spectrum=False, k1_form=0, sparse_k1=None):
        """coarse_channels > 0: the reference's two-resolution layout -- C channels at L^3 plus
        ``coarse_channels`` at (L/2)^3 (ProteinRepresentationModels.py:72-76); W1 is (H, C+coarse).
        extent < L: the volumes are extent^3 boxes in the corner of the L^3 ones (a box size without a compiled plan
        inside this one, Docker._dock_volumes_embedded): rotated ligand volumes are cropped to that box (coarse grid:
        extent / 2), ``center`` / ``coarse_center`` are the pivots of the small boxes."""
        self.device = torch.device(device)
        if lib is None:
            if self.device.type != "cuda":
                raise RuntimeError("dlpd: the product path runs on an AMD GPU only "
                                   "(no CPU fallback); got device %s" % self.device)
            lib = get_lib()
        self.lib = lib
        if not lib.call("dlpd_grid_supported", int(L)):
            raise RuntimeError("dlpd: box size L=%d has no compiled fused pipeline" % L)
        self.L, self.N, self.C = int(L), 2 * int(L), int(C)
        self.NZ = self.N // 2 + 1
        self.has_clash = bool(has_clash)
        self.CT = self.C + (1 if self.has_clash else 0)
        self.center = float(L) / 2.0 if center is None else float(center)
        # Conventions of the volume rotation and of VolumeConvolution(clip) that TorchProteinLibrary may define
        # differently (Utils/Conventions.py): scale + axis order are folded into the 3x3 maps K1 samples with.  The clash
        # channel follows where it comes from: re-projected ATOMS (clash_provider, the reference's own path,
        # Docker.py:221-224) are rotated geometrically by the TRUE R; a stored forbidden VOLUME (dock_volumes without a
        # provider -- the synthetic configs) is a volume like the others and turns with the mapped matrix; clip_mode "input" clamps the receptor once
        # and every rotated ligand batch before its transform (through the volumes path: exact, one extra round trip of
        # the rotated volumes -- the price of a convention nobody has confirmed), "none" drops the clamp.
        self.rot_scale = float(rotation_scale)
        self.rot_scale1 = float(rotation_scale if coarse_rotation_scale is None else coarse_rotation_scale)
        self.rot_axis_order, self.rot_transpose = rotation_axis_order, bool(rotation_transpose)
        self._rot_mapped = self.rot_scale != 1.0 or self.rot_scale1 != 1.0 or rotation_axis_order != "xyz" or self.rot_transpose
        if clip_mode not in ("output", "input", "none"):
            raise RuntimeError("dlpd: clip_mode %r" % (clip_mode,))
        self.clip_mode = clip_mode
        self.clip = clip
        self.threshold = float(threshold_clash)
        self.K = int(max_conf)
        self.batch = int(batch)
        dev = self.device
        f32 = torch.float32
        self.C1 = int(coarse_channels)
        self.k3_form = int(k3_form)
        # kernel formulation of the channels-last K1 (include/dlpd.h, dlpd_zfft_channels_last_form): 0 = the library's
        # default, 1 = every wave gathers / transforms / stores in turn, 2 = role-split (boxes 64 and 80); same bits
        self.k1_form = int(k1_form)
        if self.k1_form not in (0, 1, 2):
            raise RuntimeError("dlpd: k1_form %r (0 = library default, 1 = phased, 2 = role-split)" % (k1_form,))
        if self.k1_form == 2 and int(L) in (64, 80) and not lib.call("dlpd_k1_form_supported", int(L), 2):
            # form 2 is a TEST-VARIANT kernel (-DDLPD_TEST_VARIANTS builds: tests/variants/libdlpd_variants.so): the product
            # library refuses it at the first launch -- say so when the engine is built, not in the middle of a search
            raise RuntimeError("dlpd: k1_form=2 (role-split K1) exists in -DDLPD_TEST_VARIANTS builds only; this library "
                               "(%s) does not hold it" % getattr(lib, "path", "?"))
        self.fine_unfused = bool(fine_unfused)
        self.set_filter(W1, b1, W2, b2)
        nb, CT, NZ, N = self.batch, self.CT, self.NZ, self.N
        self.lig = torch.zeros(CT, L, L, L, dtype=f32, device=dev)
        self.recF = torch.zeros(CT, NZ, N, N, 2, dtype=f32, device=dev)
        # boxes whose K2 re-reads the receptor spectrum for every rotation (80, 40) take a copy in the order its column
        # phase consumes it (include/dlpd.h: dlpd_receptor_pack); written by set_receptor, read by untransposed launches
        npk = lib.call("dlpd_receptor_packed_floats", CT, int(L)) if packed_receptor else 0
        self.recP = torch.zeros(npk, dtype=f32, device=dev) if npk else None
        self.recP1 = None
        self.wsA = torch.empty(nb * CT * NZ * L * L * 2, dtype=f32, device=dev)
        self.wsB = torch.empty(nb * CT * NZ * N * N * 2, dtype=f32, device=dev)
        self.V = torch.empty(nb, N, N, N, dtype=f32, device=dev)
        # channels-last gather (include/dlpd.h): one 16-byte load serves four channels of a corner, so the rotation
        # costs the same for every rotation; slab orientation and the quad layout are the per-channel kernel's
        # remedies and stay for ligands with few channels (and as diagnostic switches)
        # Kernel choices are CONSTRUCTOR ARGUMENTS only (no switch is read from outside the call): channels_last (default: ligands with
        # >= 8 channels), orient / quads (the per-channel K1's launch variants), prefilter (top-K candidate lists from
        # K3).  They select equivalent kernels, never less work; ``switches()`` reports what is active.
        if channels_last is None:
            channels_last = self.C >= 8
        self.use_cl = bool(channels_last)
        self.extent = int(extent) if extent and int(extent) < int(L) else 0
        self.extent1 = self.extent // 2
        if self.extent and self.clip_mode == "input":
            raise RuntimeError("dlpd: clip_mode 'input' is not combined with embedded boxes (use a compiled box size)")
        # (slab orientation needs K2's transposed reader: every compiled box except 80 in libdlpd.so)
        self.orient = bool(orient) and not self.use_cl and bool(lib.call("dlpd_orientation_supported", int(L)))
        self.use_quads = bool(quads) and not self.use_cl and not self.extent
        if self.use_cl:
            self.ligcl = torch.empty(lib.call("dlpd_channels_last_floats", self.C, int(L)), dtype=f32, device=dev)
        # With the channels-last K1 every launch is untransposed and goes through the staged path, so where K2 reads the
        # PACKED receptor copy (boxes 80 / 40) the natural-layout spectrum is dead once it has been packed: it becomes a
        # temporary of set_receptor instead of a second resident copy (282 MB at 17 channels, 813 MB at 49, box 80).
        self._spectrum_is_temporary = bool(self.use_cl and not keep_receptor_spectrum)
        if self._spectrum_is_temporary and self.recP is not None:
            self.recF = None
        self.prefilter = bool(prefilter)
        # Search-side sparsity (round 6): a ligand whose channels are zero in most 4^3 cells of its box (any real protein's
        # representation) gets per-rotation occupancy maps, and the channels-last K1 skips what they mark empty -- same
        # spectra.  None: decided per ligand and grid in set_ligand (on where the ligand's cells, dilated by one, stay below
        # SPARSE_K1_MAX_FILL of the box); True / False force it.
        self.sparse_k1_wanted = sparse_k1
        self.sparse_k1 = self.sparse_k1_coarse = self.k2_pencil_map = self.k2_pencil_map_coarse = False
        self._k2_by_map = {False: False, True: False}
        self.lig_fill = self.lig_fill_coarse = None
        self.window = None
        if self.extent:
            # the reference's (2 extent)^3 translation grid inside this engine's (2L)^3 one: index t for 0 <= t <= extent
            # (t = extent: no overlap), 2L + t for -extent < t < 0 -- monotonic, so equal scores keep the reference's order.
            # The top-K takes the gathered grid; K3's candidate lists carry indices of the large grid and are not used.
            e = self.extent
            self.window = torch.tensor(list(range(0, e + 1)) + list(range(2 * int(L) - (e - 1), 2 * int(L))), dtype=torch.long, device=dev)
            # ... as ONE flat gather index over the (2L)^3 volume (the three per-axis index_selects it replaces made
            # three passes over the score volume per batch)
            w, Nn = self.window, 2 * int(L)
            self.window_flat = ((w[:, None, None] * Nn + w[None, :, None]) * Nn + w[None, None, :]).reshape(-1).contiguous()
            self.prefilter = False
        if self.use_quads:
            self.ligq = torch.empty(lib.call("dlpd_quads_floats", CT, int(L)), dtype=f32, device=dev)
        if self.C1:
            L1 = self.L // 2
            if self.L % 2 or not lib.call("dlpd_grid_supported", L1):
                raise RuntimeError("dlpd: coarse box size %d has no compiled pipeline" % L1)
            N1, NZ1, C1 = 2 * L1, L1 + 1, self.C1
            self.L1 = L1
            self.center1 = float(L1) / 2.0 if coarse_center is None else float(coarse_center)
            self.lig1 = torch.zeros(C1, L1, L1, L1, dtype=f32, device=dev)
            self.recF1 = torch.zeros(C1, NZ1, N1, N1, 2, dtype=f32, device=dev)
            npk1 = lib.call("dlpd_receptor_packed_floats", C1, L1) if packed_receptor else 0
            self.recP1 = torch.zeros(npk1, dtype=f32, device=dev) if npk1 else None
            if self._spectrum_is_temporary and self.recP1 is not None:
                self.recF1 = None
            self.wsA1 = torch.empty(nb * C1 * NZ1 * L1 * L1 * 2, dtype=f32, device=dev)
            self.wsB1 = torch.empty(nb * C1 * NZ1 * N1 * N1 * 2, dtype=f32, device=dev)
            if self.use_quads:
                self.ligq1 = torch.empty(lib.call("dlpd_quads_floats", C1, L1), dtype=f32, device=dev)
            if self.use_cl:
                self.ligcl1 = torch.empty(lib.call("dlpd_channels_last_floats", C1, L1), dtype=f32, device=dev)
        # fine_unfused (diagnostic / A-B): materialise the real correlations of the fine grid and run the
        # vectorised filter kernel behind a plain z-inverse, instead of the fused z-inverse + MLP (K3)
        self.fine_unfused = bool(fine_unfused)
        if self.fine_unfused:
            self.conv = torch.empty(nb, CT, N, N, N, dtype=f32, device=dev)
        if self.C1:
            # first-layer pre-activations of the coarse channels on the coarse grid (dlpd_zifft_preact): HP planes
            self.pre = torch.empty(nb, self.HP, 2 * self.L1, 2 * self.L1, 2 * self.L1, dtype=f32, device=dev)
        self.top = DeviceTopList(self.K, nb, dev, lib)
        # optional: callable(R (nb,3,3) f32 device) -> (nb,L,L,L) f32 device ligand forbidden volumes
        # re-projected from rotated ATOMS (Docker.py:221-224) instead of the rotated volume
        self.clash_provider = None
        # kernel formulation of the fused K3 (include/dlpd.h, dlpd_zifft_filter_form): 0 = the library's default
        # (role-split transform / filter waves where compiled), 1 = channel-owning waves; same results bit for bit
        self.k3_form = int(k3_form)

    @property
    def _out_clip(self):
        """(has_clip, clip) of the kernels' clamp on the correlation OUTPUT."""
        on = self.clip is not None and self.clip_mode == "output"
        return (1 if on else 0), float(self.clip if on else 0.0)

    @property
    def _in_clip(self):
        return float(self.clip) if (self.clip is not None and self.clip_mode == "input") else None

    def _kernel_R(self, R, coarse=False):
        """The 3x3 maps K1 samples with: R itself, or scale * P R P (Utils/Conventions.kernel_matrices)."""
        if not self._rot_mapped:
            return R
        from .Utils.Conventions import kernel_matrices
        return kernel_matrices(R, self.rot_scale1 if coarse else self.rot_scale, self.rot_axis_order, self.rot_transpose)

    def _k1_form_at(self, L):
        """The requested K1 formulation where the library holds both (boxes 64, 80); its only one elsewhere."""
        return self.k1_form if int(L) in (64, 80) else 0

    def switches(self):
        """Which of the equivalent kernel formulations and which conventions this engine launches with (bench.py records it)."""
        return {"k1": "channels_last" if self.use_cl else "per_channel",
                "k1_slab_orientation": bool(self.orient), "k1_quad_layout": bool(self.use_quads),
                "k1_form": {0: "library default", 1: "phased", 2: "role-split"}[self.k1_form],
                "k3_form": {0: "library default (role-split where compiled)", 1: "channel-owning", 2: "role-split"}[self.k3_form],
                "k3_unfused": bool(self.fine_unfused), "topk_candidate_lists": bool(self.prefilter),
                "k2_packed_receptor": {"fine": self.recP is not None, "coarse": self.recP1 is not None},
                "embedded_extent": self.extent or None,
                "k1_occupancy_maps": {"fine": bool(self.sparse_k1), "coarse": bool(self.sparse_k1_coarse),
                                      "k2_pencil_map": {"fine": bool(self.k2_pencil_map), "coarse": bool(self.k2_pencil_map_coarse)},
                                      "ligand_cells_occupied": {"fine": self.lig_fill, "coarse": self.lig_fill_coarse}},
                "rotation": {"center": self.center, "scale": self.rot_scale, "axis_order": self.rot_axis_order,
                             "transpose": self.rot_transpose},
                "clip_mode": self.clip_mode}

    # ---- inputs ------------------------------------------------------------------------
    def set_filter(self, W1, b1, W2, b2):
        """SimpleFilter parameters (DockingModels.py:28-32) in the kernels' layout: W1 transposed and zero-padded to
        the hidden width the filter kernel is compiled for.  May be called again on a live engine (same shapes)."""
        f32, dev, lib = torch.float32, self.device, self.lib
        W1 = torch.as_tensor(W1, dtype=f32).reshape(-1, self.C + self.C1)
        H = W1.shape[0]
        HP = lib.call("dlpd_fused_hidden_pad", int(H), self.L, int(self.C1 > 0))
        if HP < 0:
            raise RuntimeError("dlpd: hidden width %d has no fused filter kernel at box %d" % (H, self.L))
        if HP > 32 and (getattr(self, "fine_unfused", False) or self.k3_form == 1):
            raise RuntimeError("dlpd: hidden width %d runs on the role-split K3 only" % H)
        if getattr(self, "HP", HP) != HP:
            raise RuntimeError("dlpd: a live engine keeps its hidden width (%d), got %d" % (self.HP, HP))
        self.H, self.HP = H, HP
        W1t = torch.zeros(self.C + self.C1, HP, dtype=f32)
        W1t[:, :H] = W1.t()
        b1p = torch.zeros(HP, dtype=f32)
        b1p[:H] = torch.as_tensor(b1, dtype=f32).reshape(-1)
        W2p = torch.zeros(HP, dtype=f32)
        W2p[:H] = torch.as_tensor(W2, dtype=f32).reshape(-1)
        self.W1t, self.b1, self.W2 = W1t.to(dev), b1p.to(dev), W2p.to(dev)
        self.b2 = float(torch.as_tensor(b2).reshape(-1)[0])

    def set_receptor(self, rec_volumes, rec_forbidden=None, rec_coarse=None):
        """rec_volumes (C,L,L,L); rec_forbidden (L,L,L); rec_coarse (C1,L/2,..).  Spectra precomputed
        once per pair (the reference recomputes them every batch inside VolumeConvolution,
        DockingModels.py:71)."""
        L, N, CT = self.L, self.N, self.CT
        if self.C1:
            L1 = self.L1
            r1 = torch.as_tensor(rec_coarse, dtype=torch.float32).reshape(self.C1, L1, L1, L1).to(self.device).contiguous()
            if self._in_clip is not None:
                r1 = r1.clamp(-self._in_clip, self._in_clip)
            recF1 = self.recF1 if self.recF1 is not None else \
                torch.empty(self.C1, L1 + 1, 2 * L1, 2 * L1, 2, dtype=torch.float32, device=self.device)
            self.lib.call("dlpd_rfft3d_padded", _ptr(r1), _ptr(recF1), _ptr(self.wsA1), self.C1, L1,
                          1.0 / float(2 * L1) ** 3, _stream(self.device))
            if self.recP1 is not None:
                self.lib.call("dlpd_receptor_pack", _ptr(recF1), _ptr(self.recP1), self.C1, L1, _stream(self.device))
        rec = torch.zeros(CT, L, L, L, dtype=torch.float32, device=self.device)
        rec[: self.C] = torch.as_tensor(rec_volumes, dtype=torch.float32).reshape(self.C, L, L, L).to(self.device)
        if self._in_clip is not None:
            rec[: self.C].clamp_(-self._in_clip, self._in_clip)
        if self.has_clash:
            rec[self.C] = torch.as_tensor(rec_forbidden, dtype=torch.float32).reshape(L, L, L).to(self.device)
        scale = 1.0 / float(N) ** 3
        recF = self.recF if self.recF is not None else \
            torch.empty(CT, self.NZ, N, N, 2, dtype=torch.float32, device=self.device)
        self.lib.call("dlpd_rfft3d_padded", _ptr(rec), _ptr(recF), _ptr(self.wsA), CT, L, scale,
                      _stream(self.device))
        if self.recP is not None:
            self.lib.call("dlpd_receptor_pack", _ptr(recF), _ptr(self.recP), CT, L, _stream(self.device))

    def _k2(self, coarse, nb, tr, st):
        """Stage K2 of the fine or the coarse grid on what K1 left in its wsA; tr: the slab orientation K1 used."""
        wsA, rec, recP, wsB, CT, L = ((self.wsA1, self.recF1, self.recP1, self.wsB1, self.C1, self.L1) if coarse else
                                      (self.wsA, self.recF, self.recP, self.wsB, self.CT, self.L))
        by_map = self._k2_by_map[bool(coarse)]
        if by_map is not False:
            # K1 left the pencils of empty blocks unwritten: K2 takes them as zeros by the map (score channels only).  The map
            # is the engine's own (rotation path: True) or the one made from the batch's volumes (a tensor)
            assert recP is not None and not tr
            self._k2_by_map[bool(coarse)] = False
            pen = by_map if torch.is_tensor(by_map) else (self.pen_rot1 if coarse else self.pen_rot)
            self.lib.call("dlpd_xy_correlate_packed_occ", _ptr(wsA), _ptr(recP), _ptr(wsB), nb, CT, L, _ptr(pen),
                          self.C1 if coarse else self.C, st)
        elif recP is not None and not tr:
            self.lib.call("dlpd_xy_correlate_packed", _ptr(wsA), _ptr(recP), _ptr(wsB), nb, CT, L, st)
        else:
            if rec is None:          # (packed-receptor boxes drop the natural-layout spectrum: keep_receptor_spectrum=True keeps it)
                raise RuntimeError("dlpd: this engine holds only the packed receptor spectrum; a transposed launch needs "
                                   "keep_receptor_spectrum=True")
            self.lib.call("dlpd_xy_correlate_oriented", _ptr(wsA), _ptr(rec), _ptr(wsB), nb, CT, L, 0, tr, st)

    SPARSE_K1_MAX_FILL = 0.7

    def _ligand_occupancy(self):
        """Cell maps of the stored ligand (all score channels) and the decision whether K1 goes by per-rotation maps."""
        from . import ops
        self.sparse_k1 = self.sparse_k1_coarse = False
        if not self.use_cl or self.sparse_k1_wanted is False:
            return
        nc = (self.L + 3) // 4

        def reach(occ):
            """Fraction of the cells a ROTATED copy can mark: the per-rotation maps are conservative by about one cell per
            side, so the decision goes by the ligand's cells dilated by one (a 60 %-full coarse grid marks everything)."""
            o = occ.to(torch.float32).reshape(1, 1, *occ.shape[-3:])
            return float(torch.nn.functional.max_pool3d(o, kernel_size=3, stride=1, padding=1).mean())
        self.occ_src = ops.tile_occupancy(self.lig[: self.C].unsqueeze(0), lib=self.lib)
        self.lig_fill = float(self.occ_src.float().mean())
        self.sparse_k1 = bool(self.sparse_k1_wanted) or reach(self.occ_src) < self.SPARSE_K1_MAX_FILL
        if self.sparse_k1 and not hasattr(self, "occ_rot"):
            self.occ_rot = torch.empty(self.batch, nc, nc, nc, dtype=torch.uint8, device=self.device)
            self.pen_rot = torch.empty(self.batch, nc, dtype=torch.int32, device=self.device)
        if self.C1:
            nc1 = (self.L1 + 3) // 4
            self.occ_src1 = ops.tile_occupancy(self.lig1.unsqueeze(0), lib=self.lib)
            self.lig_fill_coarse = float(self.occ_src1.float().mean())
            self.sparse_k1_coarse = bool(self.sparse_k1_wanted) or reach(self.occ_src1) < self.SPARSE_K1_MAX_FILL
            if self.sparse_k1_coarse and not hasattr(self, "occ_rot1"):
                self.occ_rot1 = torch.empty(self.batch, nc1, nc1, nc1, dtype=torch.uint8, device=self.device)
                self.pen_rot1 = torch.empty(self.batch, nc1, dtype=torch.int32, device=self.device)
        # Where K2 reads the packed receptor (boxes 80 / 40) it can go by a per-rotation PENCIL map: K1 then does not write
        # the blocks without an occupied cell and K2 does not read the pencils the map marks empty (same spectra, same lists)
        self.k2_pencil_map = bool(self.sparse_k1 and self.recP is not None and self.lib.call("dlpd_pencil_map_supported", self.L))
        self.k2_pencil_map_coarse = bool(self.C1 and self.sparse_k1_coarse and self.recP1 is not None and
                                         self.lib.call("dlpd_pencil_map_supported", self.L1))

    def _k1_channels_last(self, coarse, R, nb, st):
        """Rotation + z transform of the score channels from the channels-last copy, by occupancy maps where the ligand is sparse."""
        call = self.lib.call
        if coarse:
            cl, wsA, C, CT, L, c0, ext, sparse = self.ligcl1, self.wsA1, self.C1, self.C1, self.L1, self.center1, self.extent1, self.sparse_k1_coarse
            occ_src, occ_rot, pen, skip = (self.occ_src1, self.occ_rot1, self.pen_rot1, self.k2_pencil_map_coarse) if sparse else (None,) * 4
        else:
            cl, wsA, C, CT, L, c0, ext, sparse = self.ligcl, self.wsA, self.C, self.CT, self.L, self.center, self.extent, self.sparse_k1
            occ_src, occ_rot, pen, skip = (self.occ_src, self.occ_rot, self.pen_rot, self.k2_pencil_map) if sparse else (None,) * 4
        self._k2_by_map[bool(coarse)] = False
        if sparse and self._k1_form_at(L) in (0, 1):
            call("dlpd_rotated_occupancy", _ptr(occ_src), _ptr(R), _ptr(occ_rot), _ptr(pen) if skip else 0, nb, L, c0, st)
            call("dlpd_zfft_channels_last_occ", _ptr(cl), _ptr(R), _ptr(occ_rot), _ptr(wsA), nb, C, CT, 0, L, c0, ext, int(skip), st)
            self._k2_by_map[bool(coarse)] = bool(skip)        # this launch's K2 must go by the pencil map (wsA holds unwritten pencils)
        else:
            call("dlpd_zfft_channels_last_form", _ptr(cl), _ptr(R), _ptr(wsA), nb, C, CT, 0, L, c0, ext, self._k1_form_at(L), st)

    def set_ligand(self, lig_volumes, lig_forbidden=None, lig_coarse=None):
        L = self.L
        if self.C1:
            self.lig1.copy_(torch.as_tensor(lig_coarse, dtype=torch.float32).reshape(self.lig1.shape))
        self.lig[: self.C] = torch.as_tensor(lig_volumes, dtype=torch.float32).reshape(self.C, L, L, L).to(self.device)
        if self.has_clash:
            self.lig[self.C] = torch.as_tensor(lig_forbidden, dtype=torch.float32).reshape(L, L, L).to(self.device)
        if self.use_cl:
            st = _stream(self.device)
            self.lib.call("dlpd_make_channels_last", _ptr(self.lig), _ptr(self.ligcl), self.C, L, st)
            if self.C1:
                self.lib.call("dlpd_make_channels_last", _ptr(self.lig1), _ptr(self.ligcl1), self.C1, self.L1, st)
        self._ligand_occupancy()
        # quad layout for the rotation gather (include/dlpd.h), once per pair: 4x the ligand's bytes
        if self.use_quads:
            st = _stream(self.device)
            self.lib.call("dlpd_make_quads", _ptr(self.lig), _ptr(self.ligq), self.CT, L, st)
            if self.C1:
                self.lib.call("dlpd_make_quads", _ptr(self.lig1), _ptr(self.ligq1), self.C1, self.L1, st)

    @staticmethod
    def prefers_quads(R):
        """(n,3,3) -> bool (n,): after the slab orientation the source z axis is carried by the in-plane
        x/y axis rather than by the output z axis.  For those rotations the quad-layout gather
        (include/dlpd.h) is ~1.4x faster; for the others the plain volume (4x fewer bytes, L2 resident)
        is as good or better inside the full pipeline."""
        R = np.asarray(R)
        inplane = np.maximum(np.abs(R[:, 0, 2]), np.abs(R[:, 1, 2]))
        return inplane > np.abs(R[:, 2, 2])

    @staticmethod
    def prefers_transposed(R):
        """(n,3,3) rotation matrices -> bool (n,): the source z axis is closer to the output x axis than to
        the output y axis, i.e. the x-plane gather of K1 would run across memory rows (include/dlpd.h)."""
        R = np.asarray(R)
        return np.abs(R[:, 0, 2]) > np.abs(R[:, 1, 2])

    # ---- hot loop ------------------------------------------------------------------------
    def score_batch(self, R, mark=None, out=None, volumes=None, transposed=False, quads=False, cset=None, occupancy=None):
        """R (nb,3,3) float32 on the device, nb <= batch.  Returns V[:nb] (view of the engine's
        buffer, overwritten by the next call): Docker.py:218-232.  mark(name): optional callback
        after each stage (timing).
        volumes = (lig (nb,C,L,L,L), forbidden (nb,L,L,L) | None, coarse (nb,C1,L/2,..) | None): the
        batch's ligand volumes are given as they are (dockE3: re-projected and re-represented per
        rotation, Docker.py:163-172) instead of rotating the stored ligand by R.
        occupancy = (fine map, coarse map | None) with ``volumes``: uint8 (nb, ceil(L/4)^3) maps of the cells that hold a
        non-zero value (ops.conv3d(return_occupancy=True)); cells marked empty are never read -- the volumes may be
        unwritten there (ops.conv3d(unwritten=True)).
        transposed: slab orientation for ALL rotations of the batch (include/dlpd.h); search() groups the
        rotations for which it pays (prefers_transposed) into batches of their own."""
        self._cset_used = None
        if volumes is not None:
            return self._score_volumes(volumes, mark, out, cset, occupancy)
        nb = R.shape[0]
        assert nb <= self.batch and R.dtype == torch.float32 and R.is_contiguous()
        tr = int(bool(transposed) and self.orient)
        use_quads = bool(quads) and self.use_quads
        has_clip, clip = self._out_clip
        V = self.V if out is None else out
        call, st, L = self.lib.call, _stream(self.device), self.L
        provider = self.clash_provider if self.has_clash else None
        if self._in_clip is not None:
            return self._score_rotated_then_clamped(R, mark, out, cset, provider)
        R_true, R, R1 = R, self._kernel_R(R), (self._kernel_R(R, coarse=True) if self.C1 else None)
        self._keepR = (R, R1)                          # mapped copies stay alive until the stream has consumed them
        if not (self.C1 or provider or self.fine_unfused or mark or use_quads or self.use_cl or cset is not None or self.k3_form
                or self.extent):
            call("dlpd_score_rotations_oriented", _ptr(self.lig), _ptr(self.recF), _ptr(R), nb, self.C,
                 int(self.has_clash), L, self.center, _ptr(self.W1t), _ptr(self.b1), _ptr(self.W2), self.b2,
                 self.HP, has_clip, clip, self.threshold, _ptr(self.wsA), _ptr(self.wsB), _ptr(V), tr, st)
            return V[:nb]
        mark = mark or (lambda name: None)
        mark("begin")
        if self.C1:
            # coarse resolution first: rotate + correlate + clip -> real volumes the fine filter reads
            L1 = self.L1
            if self.use_cl:
                self._k1_channels_last(True, R1, nb, st)
            elif use_quads:
                call("dlpd_zfft_quads", _ptr(self.ligq1), _ptr(R1), _ptr(self.wsA1), nb, self.C1, self.C1, 0, L1,
                     self.center1, tr, st)
            else:
                call("dlpd_zfft_oriented_ext", _ptr(self.lig1), _ptr(R1), _ptr(self.wsA1), nb, self.C1, self.C1, 0, L1, 0, 1,
                     self.center1, tr, self.extent1, st)
            sub = getattr(mark, "sub_stages", False)     # (diagnostic callers: a mark after every kernel of the coarse stage)
            if sub:
                mark("coarse_k1")
            self._k2(True, nb, tr, st)
            if sub:
                mark("coarse_k2")
            self._coarse_preact(nb, has_clip, clip, st)
            mark("coarse")
        if provider is not None:
            # clash channel from re-projected rotated ATOMS (Docker.py:221-224), scores from rotated volumes
            forb = provider(R_true).reshape(nb, L, L, L).contiguous()
            if self.use_cl:
                self._k1_channels_last(False, R, nb, st)
            elif use_quads:
                call("dlpd_zfft_quads", _ptr(self.ligq), _ptr(R), _ptr(self.wsA), nb, self.C, self.CT, 0, L,
                     self.center, tr, st)
            else:
                call("dlpd_zfft_oriented_ext", _ptr(self.lig), _ptr(R), _ptr(self.wsA), nb, self.C, self.CT, 0, L, 0, 1,
                     self.center, tr, self.extent, st)
            call("dlpd_zfft_oriented", _ptr(forb), 0, _ptr(self.wsA), nb, 1, self.CT, self.C, L, L ** 3, 0, 0.0,
                 tr, st)                      # same orientation as the score channels
        elif self.use_cl:
            self._k1_channels_last(False, R, nb, st)
            if self.has_clash:                # the ligand's forbidden volume: one channel, per-channel kernel
                call("dlpd_zfft_oriented_ext", self.lig.data_ptr() + self.C * L ** 3 * 4, _ptr(R), _ptr(self.wsA), nb, 1,
                     self.CT, self.C, L, 0, 1, self.center, 0, self.extent, st)
        elif use_quads:
            call("dlpd_zfft_quads", _ptr(self.ligq), _ptr(R), _ptr(self.wsA), nb, self.CT, self.CT, 0, L,
                 self.center, tr, st)
        else:
            call("dlpd_zfft_oriented_ext", _ptr(self.lig), _ptr(R), _ptr(self.wsA), nb, self.CT, self.CT, 0, L, 0, 1,
                 self.center, tr, self.extent, st)
        mark("k1_rotate_zfft")
        return self._correlate_and_filter(nb, V, mark, tr, cset)

    def _score_volumes(self, volumes, mark, out, cset=None, occupancy=None):
        vl, vf, vc = volumes
        nb, L = vl.shape[0], self.L
        assert nb <= self.batch
        occ0, occ1 = occupancy if occupancy is not None else (None, None)
        if (occ0 is not None or occ1 is not None) and self._in_clip is not None:
            raise RuntimeError("dlpd: occupancy maps are not combined with clip_mode 'input' (the clamp reads every voxel)")

        def _occ(o, n, Lx):
            if o is None:
                return None
            nc = (Lx + 3) // 4
            if o.dtype != torch.uint8 or o.device.type != self.device.type or o.numel() != n * nc ** 3:
                raise RuntimeError("dlpd: occupancy map %s does not belong to %d volumes of box %d" % (tuple(o.shape), n, Lx))
            return o.contiguous()
        occ0, occ1 = _occ(occ0, nb, L), (_occ(occ1, nb, self.L1) if self.C1 else None)
        f32c = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        call, st = self.lib.call, _stream(self.device)
        has_clip, clip = self._out_clip
        V = self.V if out is None else out
        mark = mark or (lambda name: None)
        mark("begin")
        if self._in_clip is not None:                  # VolumeConvolution(clip) clamping its INPUTS (clip_mode "input")
            vl = f32c(vl).clamp(-self._in_clip, self._in_clip)
            vc = f32c(vc).clamp(-self._in_clip, self._in_clip) if self.C1 else vc
        if self.C1:
            L1 = self.L1
            vc = f32c(vc).reshape(nb, self.C1, L1, L1, L1)
            if occ1 is not None:
                # (boxes whose K2 reads the packed receptor: empty x-planes are not written, K2 goes by the maps' OR over z)
                skip1 = self.recP1 is not None and bool(self.lib.call("dlpd_pencil_map_supported", L1))
                if skip1:
                    self.pen_vol1 = torch.empty(nb, (L1 + 3) // 4, dtype=torch.int32, device=self.device)
                    call("dlpd_pencil_bits", _ptr(occ1), _ptr(self.pen_vol1), nb, L1, st)
                call("dlpd_zfft_volumes_occ", _ptr(vc), _ptr(occ1), _ptr(self.wsA1), nb, self.C1, self.C1, 0, L1, self.C1 * L1 ** 3,
                     int(skip1), st)
                self._k2_by_map[True] = self.pen_vol1 if skip1 else False
            else:
                call("dlpd_zfft", _ptr(vc), 0, _ptr(self.wsA1), nb, self.C1, L1, self.C1 * L1 ** 3, 0, 0.0, st)
            self._k2(True, nb, 0, st)
            self._coarse_preact(nb, has_clip, clip, st)
            mark("coarse")
        vl = f32c(vl).reshape(nb, self.C, L, L, L)
        if occ0 is not None:
            skip0 = self.recP is not None and bool(self.lib.call("dlpd_pencil_map_supported", L))
            if skip0:
                self.pen_vol = torch.empty(nb, (L + 3) // 4, dtype=torch.int32, device=self.device)
                call("dlpd_pencil_bits", _ptr(occ0), _ptr(self.pen_vol), nb, L, st)
            call("dlpd_zfft_volumes_occ", _ptr(vl), _ptr(occ0), _ptr(self.wsA), nb, self.C, self.CT, 0, L, self.C * L ** 3, int(skip0), st)
            self._k2_by_map[False] = self.pen_vol if skip0 else False
        else:
            call("dlpd_zfft_into", _ptr(vl), 0, _ptr(self.wsA), nb, self.C, self.CT, 0, L, self.C * L ** 3, 0, 0.0, st)
        if self.has_clash:
            vf = f32c(vf).reshape(nb, L, L, L)
            call("dlpd_zfft_into", _ptr(vf), 0, _ptr(self.wsA), nb, 1, self.CT, self.C, L, L ** 3, 0, 0.0, st)
        mark("k1_rotate_zfft")
        self._keep = (vl, vf, vc, occ0, occ1)          # inputs stay alive until the stream has consumed them
        return self._correlate_and_filter(nb, V, mark, 0, cset)

    def _score_rotated_then_clamped(self, R, mark, out, cset, provider):
        """clip_mode "input": the reference order is rotate (Docker.py:218) -> VolumeConvolution(clip) (DockingModels.py:71),
        so the clamp acts on the ROTATED ligand volumes: they are materialised by the stand-alone rotation kernel, clamped
        (in _score_volumes) and fed to the pipeline as given volumes.  The clash channel is never clamped (Docker.py:32,225:
        VolumeConvolution() without clip)."""
        nb, L, call, st = R.shape[0], self.L, self.lib.call, _stream(self.device)
        f32 = torch.float32
        Rk, Rk1 = self._kernel_R(R), (self._kernel_R(R, coarse=True) if self.C1 else None)
        self._keepR = (Rk, Rk1)                        # alive until the stream has consumed them
        vl = torch.empty(nb, self.C, L, L, L, dtype=f32, device=self.device)
        call("dlpd_rotate_trilinear", _ptr(self.lig), _ptr(Rk), _ptr(vl), nb, self.C, L, 0, self.center, st)
        vf = vc = None
        if self.has_clash:
            if provider is not None:
                vf = provider(R).reshape(nb, L, L, L).contiguous()
            else:
                vf = torch.empty(nb, 1, L, L, L, dtype=f32, device=self.device)
                call("dlpd_rotate_trilinear", self.lig.data_ptr() + self.C * L ** 3 * 4, _ptr(Rk), _ptr(vf), nb, 1, L, 0,
                     self.center, st)
        if self.C1:
            L1 = self.L1
            vc = torch.empty(nb, self.C1, L1, L1, L1, dtype=f32, device=self.device)
            call("dlpd_rotate_trilinear", _ptr(self.lig1), _ptr(Rk1), _ptr(vc), nb, self.C1, L1, 0,
                 self.center1, st)
        return self._score_volumes((vl, vf, vc), mark, out, cset)

    def _coarse_preact(self, nb, has_clip, clip, st):
        """Coarse grid, last stage: z-inverse + clip fused with the coarse half of the (linear) first layer
        (DockingModels.py:74-83): HP pre-activation planes on the coarse grid instead of C1 correlation volumes."""
        self.lib.call("dlpd_zifft_preact_form", _ptr(self.wsB1), _ptr(self.pre), nb, self.C1, self.L1,
                      self.W1t.data_ptr() + self.C * self.HP * 4, _ptr(self.b1), self.HP, has_clip, clip, self.k3_form, st)

    def _correlate_and_filter(self, nb, V, mark, tr, cset=None):
        """K2 + K3 (+ filter) on whatever K1 left in wsA (and the coarse result in aux); tr: the slab
        orientation K1 used."""
        has_clip, clip = self._out_clip
        call, st, L = self.lib.call, _stream(self.device), self.L
        self._k2(False, nb, tr, st)
        mark("k2_xy_corr")
        aux, C1, N1 = (_ptr(self.pre), self.C1, 2 * self.L1) if self.C1 else (0, 0, 0)
        if self.fine_unfused:
            N3 = self.N ** 3
            call("dlpd_zifft_real_part", _ptr(self.wsB), _ptr(self.conv), nb, self.CT, self.C, L, has_clip, clip, st)
            mark("k3_zifft")
            mask = self.conv.data_ptr() + self.C * N3 * 4 if self.has_clash else 0
            call("dlpd_filter_volumes", _ptr(self.conv), self.C, self.CT * N3, self.N, aux, C1, N1, int(C1 > 0), mask,
                 self.CT * N3, self.threshold, int(self.has_clash), _ptr(self.W1t), _ptr(self.b1), _ptr(self.W2),
                 self.b2, self.HP, _ptr(V), nb, st)
            mark("filter")
        else:
            # fused K3; with a candidate set it also appends every score below the running K-th one to the batch's
            # candidate lists, which the top-K select then takes instead of a radix select over V
            tau, ck, cc, cap = (_ptr(self.top.tau), _ptr(cset["keys"]), _ptr(cset["count"]), cset["cap"]) if cset else (0, 0, 0, 0)
            call("dlpd_zifft_filter_form", _ptr(self.wsB), _ptr(V), nb, self.C, int(self.has_clash), L,
                 _ptr(self.W1t), _ptr(self.b1), _ptr(self.W2), self.b2, self.HP, has_clip, clip, self.threshold,
                 aux, C1, int(C1 > 0), tau, ck, cc, cap, self.k3_form, st)
            self._cset_used = cset
            mark("k3_zifft_filter")
        return V[:nb]

    def reset_top(self):
        self.finish()                                   # nothing of the previous pair still in flight
        self.top.reset()

    def select_batch(self, V, nb, cset=None):
        """Per-rotation picks of Docker.update_top (Docker.py:89-98) for V (nb, N^3).  cset: the candidate set the
        scoring kernel filled for this batch -- then the rows hold only the picks that can still enter the running list
        (+inf padded) and the cset is consumed; see ``DeviceTopList.select``."""
        if self.window is not None:
            V = self.gather_window(V, nb)
        return self.top.select(V.reshape(nb, -1), nb, cset)

    def gather_window(self, V, nb):
        """(nb, N^3) scores of this engine's grid -> (nb, (2 extent)^3): the reference's grid of the embedded box."""
        return V.reshape(nb, -1).index_select(1, self.window_flat)

    def merge_batch(self, rot_ids, nb):
        """Docker.py:100-105 on the device-resident list.  rot_ids int32 (nb,) ascending."""
        self.top.merge(rot_ids, nb)

    # ---- two-stream pipeline: top-K of batch i overlaps K1/K2 of batch i+1 -----------------
    def step(self, R, rot_ids, mark=None, volumes=None, transposed=False, quads=False, occupancy=None):
        """One batch: score on the current stream; select + merge on a side stream (they are
        latency-bound one-block kernels that fit beside the FFT blocks).  V is double-buffered;
        call finish() before reading the list."""
        nb = R.shape[0] if volumes is None else volumes[0].shape[0]
        if self.device.type != "cuda":                  # emulated library (tests): same calls, one stream
            if self.prefilter and not hasattr(self, "_cset_cpu"):
                self._cset_cpu = self.top.new_candidate_set()
            V = self.score_batch(R, mark=mark, volumes=volumes, transposed=transposed, quads=quads,
                                 cset=getattr(self, "_cset_cpu", None), occupancy=occupancy)
            self.select_batch(V, nb, self._cset_used)
            self.merge_batch(rot_ids, nb)
            return
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(device=self.device)
            self._Vbuf = [self.V, torch.empty_like(self.V)]
            self._csets = [self.top.new_candidate_set(), self.top.new_candidate_set()] if self.prefilter else [None, None]
            self._consumed = [None, None]
            self._ids_alive = [None, None]
            self._k = 0
        k = self._k
        self._k ^= 1
        main = torch.cuda.current_stream(self.device)
        if self._consumed[k] is not None:
            main.wait_event(self._consumed[k])          # V[k] free again
        # (Holding the previous batch's select + merge back until K1 of THIS batch has been issued was measured:
        # K1 -0.06 ms, K2 +0.11 ms -- not kept.)
        V = self.score_batch(R, mark=mark, out=self._Vbuf[k], volumes=volumes, transposed=transposed, quads=quads,
                             cset=self._csets[k], occupancy=occupancy)
        cset = self._cset_used
        self._launch_pending(main)
        # the side stream reads rot_ids later: keep the caller's tensor alive (and its memory out of the
        # allocator's reach) until this buffer slot comes round again
        self._ids_alive[k] = rot_ids
        ready = torch.cuda.Event()
        ready.record(main)
        self._pending = (V, nb, rot_ids, ready, k, cset)
        self._launch_pending(main)

    def _launch_pending(self, main):
        """Enqueue select + merge of the batch that finished last on the side stream: after its scores are
        complete (ready) and after everything issued on the main stream so far."""
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is None:
            return
        V, nb, rot_ids, ready, k, cset = pend
        gate = torch.cuda.Event()
        gate.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            self._side.wait_event(gate)
            self.select_batch(V, nb, cset)
            self.merge_batch(rot_ids, nb)
            done = torch.cuda.Event()
            done.record(self._side)
        self._consumed[k] = done

    def finish(self):
        if hasattr(self, "_side"):
            main = torch.cuda.current_stream(self.device)
            self._launch_pending(main)
            main.wait_stream(self._side)

    def search(self, R_all, rot_ids=None, progress=None):
        """Score every rotation in R_all (nrot,3,3) and fold it into the running top list.
        rot_ids: global rotation indices (ascending) for this shard; default arange.
        The rotations are visited in groups -- slab orientation (prefers_transposed) x gather layout
        (prefers_quads) -- because both are per-launch choices; the ranked list does not
        depend on the visiting order (merge key = (score, rotation, pick))."""
        dev = self.device
        R_host = torch.as_tensor(R_all).detach().cpu().numpy()
        nrot = R_host.shape[0]
        ids_host = np.arange(nrot) if rot_ids is None else torch.as_tensor(rot_ids).detach().cpu().numpy()
        flags = self.prefers_transposed(R_host) if (self.orient and nrot) else np.zeros(nrot, dtype=bool)
        qflags = self.prefers_quads(R_host) if (self.use_quads and nrot) else np.zeros(nrot, dtype=bool)
        done = 0
        for tr, qd in ((False, False), (False, True), (True, False), (True, True)):
            sel = np.nonzero((flags == tr) & (qflags == qd))[0]
            if len(sel) == 0:
                continue
            R_grp = torch.from_numpy(np.ascontiguousarray(R_host[sel])).to(device=dev, dtype=torch.float32).contiguous()
            ids_grp = torch.from_numpy(np.ascontiguousarray(ids_host[sel]).astype(np.int32)).to(dev)
            for beg in range(0, len(sel), self.batch):
                end = min(beg + self.batch, len(sel))
                self.step(R_grp[beg:end], ids_grp[beg:end], transposed=tr, quads=qd)
                done += end - beg
                if progress is not None:
                    progress(done)
        self.finish()

    # ---- results ------------------------------------------------------------------------
    def top_entries(self):
        self.finish()
        return self.top.entries()

    def top_list(self):
        """[(rotation_index, x, y, z, score)] exactly as Docker.top_list (Docker.py:100-105)."""
        return DeviceTopList.to_top_list(self.top_entries(), self.N)
