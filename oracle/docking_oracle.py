"""CPU oracle for the rotation x translation correlation search.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deeplocalproteindocking_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker.

This is a restatement (torch-CPU / numpy, no TorchProteinLibrary) of the reference path

    /root/reference/src/Docker/Docker.py:184-238     dockSE3 loop
    /root/reference/src/Docker/Docker.py:86-105      update_top
    /root/reference/src/Docker/Docker.py:107-133     write_conformations
    /root/reference/src/Models/DockingModels.py:63-84  GlobalDockingModel.forward
    /root/reference/src/Models/MultiplyVolumes.py:13-60  definition of the correlation
    /root/reference/src/Utils/Rotations.py:14-66     Euler -> matrix

Parity status
-------------
* correlation sign/index/per-channel behaviour, Euler convention, update_top (incl. the
  zero-fill quirk and stable tie order), the .dat format and the upsample/concat/MLP order of
  GlobalDockingModel.forward are PINNED to the reference: ``tests/golden/*.npz`` were produced
  by importing those reference pieces (``tests/golden/make_golden.py``) and
  ``tests/test_oracle_golden.py`` checks this module against them.
* The arithmetic of TorchProteinLibrary's VolumeRotation / VolumeConvolution(clip) lives in an
  un-vendored dependency (TorchProteinLibrary @ 16166ce4847ad3ba95cd5afdb8bb4503cad32caa,
  reference README.md:5) that is absent here and has no golden vectors in the reference:
  for the trilinear rotation conventions and the ``clip`` semantics **parity is unpinned**;
  they are build-defined below from the in-repo geometric contract (Docker.py:221-223: the
  rotation is about the box centre) and exposed as parameters.
"""
import math
import numpy as np
import torch

# --------------------------------------------------------------------------------------
# Rotations  (reference: src/Utils/Rotations.py:14-32 writeMatrix, :59-66 file parsing)
# --------------------------------------------------------------------------------------

def euler_to_matrix(phi, theta, psi):
    """R = Rz(psi) . Rx(theta) . Rz(phi), element by element as Rotations.py:16-32 (float64)."""
    phi = np.asarray(phi, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64)
    psi = np.asarray(psi, dtype=np.float64)
    cpsi, spsi = np.cos(psi), np.sin(psi)
    cth, sth = np.cos(theta), np.sin(theta)
    cphi, sphi = np.cos(phi), np.sin(phi)
    R = np.empty(phi.shape + (3, 3), dtype=np.float64)
    R[..., 0, 0] = cpsi * cphi - spsi * cth * sphi
    R[..., 0, 1] = -cpsi * sphi - spsi * cth * cphi
    R[..., 0, 2] = spsi * sth
    R[..., 1, 0] = spsi * cphi + cpsi * cth * sphi
    R[..., 1, 1] = -spsi * sphi + cpsi * cth * cphi
    R[..., 1, 2] = -cpsi * sth
    R[..., 2, 0] = sth * sphi
    R[..., 2, 1] = sth * cphi
    R[..., 2, 2] = cth
    return R


def load_eul(filename):
    """Rows 'phi theta psi' -> (N,3,3) float64 (Rotations.py:59-66)."""
    ang = np.loadtxt(filename, dtype=np.float64).reshape(-1, 3)
    return euler_to_matrix(ang[:, 0], ang[:, 1], ang[:, 2])


# --------------------------------------------------------------------------------------
# Volume rotation (TPL VolumeRotation at Docker.py:218) -- build-defined conventions
# --------------------------------------------------------------------------------------

def rotate_volume(vol, R, center=None, dtype=torch.float32, scale=1.0, axis_order="xyz", transpose=False):
    """Trilinear resampling of ``vol`` (B,C,L,L,L) under rotation ``R`` (B,3,3).

    out[b,c,i] = vol[b,c]( c0 + s R_b^T (i - c0) ), zero outside the box, i = (x,y,z) index
    vector, axis 0 of the matrix <-> first spatial index.  ``c0`` defaults to L/2 in index
    units: the reference rotates ligand coordinates about the origin and then translates them
    to box_length/2 (Docker.py:199-201,221-222), i.e. about the box centre.
    Weights are computed in ``dtype``.

    The conventions TorchProteinLibrary's VolumeRotation may differ in (its source is absent; they are what
    scripts/calibrate_tpl.py determines on a machine that has it), each written out here, not folded into R:
      center      pivot index c0 (L/2; (L-1)/2 for a grid_sample(align_corners=True)-style sampler; any number)
      scale       s: the sample offset is stretched by s -- (L-1)/L when normalised coordinates are GENERATED with one
                  align_corners convention and SAMPLED with the other (PyTorch 1.1, README.md:4, has only one)
      axis_order  "xyz": matrix axis 0 <-> first spatial index; "zyx": matrix axis 0 <-> LAST spatial index
                  (grid_sample's own convention: its grid's last dimension is (x, y, z) = (W, H, D) order)
      transpose   True: R^T in place of R (the inverse rotation)
    """
    vol = torch.as_tensor(vol)
    B, C, L = vol.shape[0], vol.shape[1], vol.shape[2]
    R = torch.as_tensor(R).to(dtype)
    if transpose:
        R = R.transpose(-1, -2)
    c0 = (L / 2.0) if center is None else float(center)
    if axis_order not in ("xyz", "zyx"):
        raise ValueError("axis_order must be 'xyz' or 'zyx'")
    ar = torch.arange(L, dtype=dtype) - c0
    gx, gy, gz = torch.meshgrid(ar, ar, ar, indexing="ij")
    d = torch.stack([gx, gy, gz], dim=-1).reshape(-1, 3)          # (L^3, 3) offsets in (first, second, third) index order
    out = torch.empty(B, C, L, L, L, dtype=dtype)
    v = vol.to(dtype)
    for b in range(B):
        # p = c0 + s R^T d  ->  row-vector form d @ R; "zyx": the matrix acts on (third, second, first)
        if axis_order == "zyx":
            p = (d.flip(-1) @ R[b]).flip(-1) * float(scale) + c0
        else:
            p = (d @ R[b]) * float(scale) + c0
        p0 = torch.floor(p)
        f = p - p0
        i0 = p0.to(torch.int64)
        acc = torch.zeros(C, L * L * L, dtype=dtype)
        flat = v[b].reshape(C, -1)
        for dx in (0, 1):
            wx = f[:, 0] if dx else (1 - f[:, 0])
            ix = i0[:, 0] + dx
            for dy in (0, 1):
                wy = f[:, 1] if dy else (1 - f[:, 1])
                iy = i0[:, 1] + dy
                for dz in (0, 1):
                    wz = f[:, 2] if dz else (1 - f[:, 2])
                    iz = i0[:, 2] + dz
                    ok = (ix >= 0) & (ix < L) & (iy >= 0) & (iy < L) & (iz >= 0) & (iz < L)
                    idx = (ix.clamp(0, L - 1) * L + iy.clamp(0, L - 1)) * L + iz.clamp(0, L - 1)
                    w = (wx * wy * wz) * ok.to(dtype)
                    acc += flat[:, idx] * w
        out[b] = acc.reshape(C, L, L, L)
    return out


# --------------------------------------------------------------------------------------
# Correlation (TPL VolumeConvolution at DockingModels.py:71 / Docker.py:225)
# semantics anchored by MultiplyVolumes.py:13-47
# --------------------------------------------------------------------------------------

def correlate_direct(v1, v2):
    """Brute force: out[b,c,t mod 2L] = sum_r v1[b,c,r+t] v2[b,c,r], t in [-(L-1), L-1]^3.

    Same slices as MultiplyVolumes.multiply (MultiplyVolumes.py:13-47); index -> translation
    wrap as Docker.py:115-120.  Entries with any |t| == L stay 0.  float64, tiny L only.
    """
    v1 = np.asarray(v1, dtype=np.float64)
    v2 = np.asarray(v2, dtype=np.float64)
    B, C, L = v1.shape[:3]
    N = 2 * L
    out = np.zeros((B, C, N, N, N), dtype=np.float64)

    def sl(d):
        return (slice(d, L), slice(0, L - d)) if d >= 0 else (slice(0, L + d), slice(-d, L))

    for dx in range(-(L - 1), L):
        ax, bx = sl(dx)
        for dy in range(-(L - 1), L):
            ay, by = sl(dy)
            for dz in range(-(L - 1), L):
                az, bz = sl(dz)
                out[:, :, dx % N, dy % N, dz % N] = (
                    v1[:, :, ax, ay, az] * v2[:, :, bx, by, bz]).sum(axis=(2, 3, 4))
    return out


CLIP_MODES = ("output", "input", "none")


def correlate_fft(v1, v2, clip=None, dtype=torch.float32, clip_mode="output"):
    """Per-channel circular cross-correlation on the 2L zero-padded grid (B,C,2L,2L,2L).

    irfftn( rfftn(v1, 2L) * conj(rfftn(v2, 2L)) ).  ``clip``: build-defined as clamping the
    OUTPUT to [-clip, clip] (TPL source absent; parity unpinned, see module docstring).
    ``clip_mode``: what VolumeConvolution(clip) (DockingModels.py:48) may mean instead -- "input": both input volumes
    are clamped to [-clip, clip] before they are correlated; "none": the argument is ignored
    (scripts/calibrate_tpl.py tells which one a TorchProteinLibrary build implements).
    """
    if clip_mode not in CLIP_MODES:
        raise ValueError("clip_mode must be one of %s" % (CLIP_MODES,))
    v1 = torch.as_tensor(v1).to(dtype)
    v2 = torch.as_tensor(v2).to(dtype)
    if clip is not None and clip_mode == "input":
        v1, v2 = torch.clamp(v1, -float(clip), float(clip)), torch.clamp(v2, -float(clip), float(clip))
    L = v1.shape[2]
    N = 2 * L
    f1 = torch.fft.rfftn(v1, s=(N, N, N), dim=(2, 3, 4))
    f2 = torch.fft.rfftn(v2, s=(N, N, N), dim=(2, 3, 4))
    out = torch.fft.irfftn(f1 * torch.conj(f2), s=(N, N, N), dim=(2, 3, 4))
    if clip is not None and clip_mode == "output":
        out = torch.clamp(out, -float(clip), float(clip))
    return out


# --------------------------------------------------------------------------------------
# Scoring model (GlobalDockingModel.forward, DockingModels.py:63-84; SimpleFilter :23-37)
# --------------------------------------------------------------------------------------

def filter_mlp(feat, W1, b1, W2, b2):
    """Linear(C,C/2) -> ReLU -> Linear(C/2,1) on the last dim (DockingModels.py:28-32)."""
    h = torch.relu(feat @ W1.t() + b1)
    return h @ W2.t() + b2


def score_volumes(receptor_volumes, ligand_volumes, W1, b1, W2, b2, clip=5.0,
                  dtype=torch.float32, clip_mode="output"):
    """GlobalDockingModel.forward: list of (B,C_i,L_i^3) pairs -> V (B,N,N,N), N = 2*L_0.

    Per-resolution correlation (:70-71), nearest upsample of smaller grids to N (:74-76,
    F.interpolate default == index//scale), channel concat (:79), channels-last + MLP (:80-83).
    """
    B = receptor_volumes[0].shape[0]
    N = 2 * receptor_volumes[0].shape[2]
    conv = []
    for r, l in zip(receptor_volumes, ligand_volumes):
        c = correlate_fft(r, l, clip=clip, dtype=dtype, clip_mode=clip_mode)
        if c.shape[2] < N:
            s = N // c.shape[2]
            assert c.shape[2] * s == N
            c = c.repeat_interleave(s, 2).repeat_interleave(s, 3).repeat_interleave(s, 4)
        conv.append(c)
    V = torch.cat(conv, dim=1).permute(0, 2, 3, 4, 1).reshape(B * N * N * N, -1)
    W1, b1, W2, b2 = (torch.as_tensor(t).to(dtype) for t in (W1, b1, W2, b2))
    V = filter_mlp(V, W1, b1, W2, b2).reshape(B, N, N, N)
    return V


def clash_mask(receptor_forbidden, ligand_forbidden_rotated, threshold_clash,
               dtype=torch.float32):
    """Docker.py:225-226: (corr(rec_forb, lig_forb) < threshold).float(), (B,N,N,N)."""
    norm = correlate_fft(receptor_forbidden, ligand_forbidden_rotated, clip=None, dtype=dtype)
    return torch.lt(norm[:, 0], float(threshold_clash)).to(dtype), norm[:, 0]


# --------------------------------------------------------------------------------------
# Top-K bookkeeping (Docker.update_top, Docker.py:86-105)
# --------------------------------------------------------------------------------------

def update_top(top_list, V, rotation_index, max_conf):
    """Faithful restatement: max_conf x (min over z, y, x; record; V[x,y,z] = 0), append,
    stable sort by score, truncate.  Mutates ``V`` (torch (N,N,N)) like the reference.
    Returns the new list of (rot, x, y, z, score)."""
    top = []
    for _ in range(max_conf):
        minval_z, ind_z = torch.min(V, dim=2)
        minval_y, ind_y = torch.min(minval_z, dim=1)
        minval_x, ind_x = torch.min(minval_y, dim=0)
        x = ind_x.item()
        y = ind_y[x].item()
        z = ind_z[x, y].item()
        top.append((x, y, z, V[x, y, z].item()))
        V[x, y, z] = 0.0
    for x, y, z, score in top:
        top_list.append((rotation_index, x, y, z, score))
    top_list.sort(key=lambda t: t[4])
    return top_list[:max_conf]


def rotation_picks_fast(V, max_conf):
    """Vectorised equivalent of the pick loop of update_top for ONE rotation.

    Order by (value, flat index); negatives first; then the zero-fill quirk: every picked voxel
    has been set to 0.0, so once the negatives are exhausted the minimum is the first zero in
    flat order among {original zeros} U {already picked voxels}, and it is picked again and
    again with score 0.0 (only if there is no zero at all is the smallest positive picked once
    and then repeated with 0.0).  Returns (flat_idx int64[K],
    score float32[K]).  Checked against ``update_top`` in tests/test_oracle_golden.py.
    """
    v = np.ascontiguousarray(np.asarray(V, dtype=np.float32)).reshape(-1)
    K = int(max_conf)
    key = v + np.float32(0.0)                       # -0.0 -> +0.0 for ordering
    kk = min(K + 1, v.size)
    part = np.argpartition(key, kk - 1)[:kk] if kk < v.size else np.arange(v.size)
    # all entries tied with the kk-th value must be considered for the index tie-break
    thr = key[part].max()
    cand = np.nonzero(key <= thr)[0]
    order = cand[np.lexsort((cand, key[cand]))]
    idx = np.empty(K, dtype=np.int64)
    sc = np.empty(K, dtype=np.float32)
    neg = order[key[order] < 0][:K]
    q = len(neg)
    idx[:q] = neg
    sc[:q] = v[neg]
    if q < K:
        rest = order[q:]
        first_nonneg = rest[0]                       # smallest (value, index) among the non-negatives
        picked_min = neg.min() if q > 0 else None    # picked voxels were zeroed: they are zeros now
        if key[first_nonneg] == 0 and (picked_min is None or first_nonneg < picked_min):
            # an original zero is the first zero in flat order: stored value (maybe -0.0), then +0.0
            idx[q:] = first_nonneg; sc[q] = v[first_nonneg]; sc[q + 1:] = np.float32(0.0)
        elif picked_min is not None:
            idx[q:] = picked_min; sc[q:] = np.float32(0.0)
        else:
            # no zero anywhere: the smallest positive is picked once, then it is the only zero
            idx[q:] = first_nonneg; sc[q] = v[first_nonneg]; sc[q + 1:] = np.float32(0.0)
    return idx, sc


def flat_to_xyz(flat, N):
    flat = np.asarray(flat)
    return flat // (N * N), (flat // N) % N, flat % N


# --------------------------------------------------------------------------------------
# Output (.dat) -- Docker.write_conformations, Docker.py:107-133
# --------------------------------------------------------------------------------------

def format_conformations(top_list, R_all, box_size, resolution, randR=None):
    """13 tab-separated %f columns per pose: 9 R entries row-major, 3 t, score."""
    lines = []
    for i, x, y, z, score in top_list:
        r = np.asarray(R_all[i], dtype=np.float64)
        t = np.array([x, y, z], dtype=np.float64)
        for a in range(3):
            if t[a] >= box_size:
                t[a] = -(2 * box_size - t[a])
        t = t * resolution
        if randR is not None:
            rt = np.asarray(randR, dtype=np.float64).reshape(3, 3).T
            t = rt @ t
            r = rt @ r
        s = ""
        for a in range(3):
            s += "%f\t%f\t%f\t" % (r[a, 0], r[a, 1], r[a, 2])
        s += "%f\t%f\t%f\t" % (t[0], t[1], t[2])
        s += "%f\n" % (score)
        lines.append(s)
    return "".join(lines)


# --------------------------------------------------------------------------------------
# Atom projection (TPL TypedCoords2Volume at Docker.py:204,208,223) -- build-defined shape
# --------------------------------------------------------------------------------------

def project_atoms(coords, num_atoms_of_type, offsets, L, resolution, R=None, shift=None, sum_types=False,
                  sigma=1.0, window=2, voxel_offset=0.0, norm=1.0):
    """coords (3*Nmax,) ordered by type; every atom adds exp(-|r - p'|^2 / (2 sigma^2)) to the (2 window + 1)^3 voxels
    around p' = R p + shift (voxel (i,j,k) at ((i,j,k) + voxel_offset)*resolution).  float64.  -> (T or 1, L,L,L).
    Defaults (sigma 1, window 2, offset 0) are this build's definition; TypedCoords2Volume's is unknown."""
    T = len(num_atoms_of_type)
    out = np.zeros((1 if sum_types else T, L, L, L), dtype=np.float64)
    xyz = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    shift = np.zeros(3) if shift is None else np.asarray(shift, dtype=np.float64).reshape(3)
    for t in range(T):
        for a in range(int(offsets[t]), int(offsets[t]) + int(num_atoms_of_type[t])):
            p = R @ xyz[a] + shift
            c = np.floor(p / resolution - voxel_offset).astype(int)
            w = int(window)
            for i in range(c[0] - w, c[0] + w + 1):
                for j in range(c[1] - w, c[1] + w + 1):
                    for k in range(c[2] - w, c[2] + w + 1):
                        if 0 <= i < L and 0 <= j < L and 0 <= k < L:
                            d = p - (np.array([i, j, k]) + voxel_offset) * resolution
                            out[0 if sum_types else t, i, j, k] += norm * np.exp(-0.5 * (d @ d) / sigma ** 2)
    return out


def project_atoms_fast(coords, num_atoms_of_type, offsets, L, resolution, R=None, shift=None, sum_types=False):
    """Same definition as project_atoms, vectorised over atoms (numpy, float64) for protein-sized
    inputs; tests/test_oracle_golden.py checks it against the loop on a small case."""
    T = len(num_atoms_of_type)
    out = np.zeros((1 if sum_types else T, L * L * L), dtype=np.float64)
    xyz = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    shift = np.zeros(3) if shift is None else np.asarray(shift, dtype=np.float64).reshape(3)
    off = np.arange(-2, 3)
    di, dj, dk = (a.reshape(-1) for a in np.meshgrid(off, off, off, indexing="ij"))
    for t in range(T):
        a0, n = int(offsets[t]), int(num_atoms_of_type[t])
        if n == 0:
            continue
        p = xyz[a0:a0 + n] @ R.T + shift                        # (n,3)
        c = np.floor(p / resolution).astype(np.int64)
        i, j, k = c[:, 0:1] + di[None], c[:, 1:2] + dj[None], c[:, 2:3] + dk[None]      # (n,125)
        ok = (i >= 0) & (i < L) & (j >= 0) & (j < L) & (k >= 0) & (k < L)
        d2 = (p[:, 0:1] - i * resolution) ** 2 + (p[:, 1:2] - j * resolution) ** 2 + (p[:, 2:3] - k * resolution) ** 2
        np.add.at(out[0 if sum_types else t], ((i * L + j) * L + k)[ok], np.exp(-0.5 * d2)[ok])
    return out.reshape(-1, L, L, L)


# --------------------------------------------------------------------------------------
# Whole search (Docker.dockSE3 loop, Docker.py:211-238) on volumes
# --------------------------------------------------------------------------------------

def rotation_scale(rule, L):
    """Scale of the volume rotation on a grid of L voxels per edge: a number, or "(L-1)/L" / "L/(L-1)" (the two ways a
    generate / sample mismatch of grid_sample's align_corners convention can go)."""
    if rule is None:
        return 1.0
    if isinstance(rule, str):
        if rule == "(L-1)/L":
            return (float(L) - 1.0) / float(L)
        if rule == "L/(L-1)":
            return float(L) / (float(L) - 1.0)
        raise ValueError("unknown rotation scale rule %r" % (rule,))
    return float(rule)


def dock_volumes(receptor_volumes, ligand_volumes, receptor_forbidden, ligand_forbidden,
                 rotations, W1, b1, W2, b2, threshold_clash, max_conf, clip=5.0,
                 rot_indices=None, faithful_topk=True, dtype=torch.float32, return_V=False,
                 clip_mode="output", rotation_center_offset=0.0, rotation_scale_rule=None, axis_order="xyz", transpose=False):
    """Run the reference loop for the given rotations on single-sample volumes.

    receptor_volumes / ligand_volumes: lists of (1,C_i,L_i^3); *_forbidden: (1,1,L^3).
    The ligand forbidden volume is rotated with the same trilinear op (build-defined stand-in
    for the per-rotation atom re-projection of Docker.py:221-224; synthetic inputs have no
    atoms).  Returns top_list of (rot, x, y, z, score) [and the list of V if return_V].
    """
    rotations = np.asarray(rotations, dtype=np.float64)
    if rot_indices is None:
        rot_indices = range(rotations.shape[0])
    top_list = []
    Vs = []
    N = 2 * receptor_volumes[0].shape[2]
    for ri in rot_indices:
        Rb = torch.from_numpy(rotations[ri:ri + 1]).to(dtype)
        def rot(v):
            Lv = v.shape[2]
            return rotate_volume(v, Rb, center=Lv / 2.0 + rotation_center_offset, dtype=dtype,
                                 scale=rotation_scale(rotation_scale_rule, Lv), axis_order=axis_order, transpose=transpose)
        lig_rot = [rot(v) for v in ligand_volumes]
        lig_forb_rot = rot(ligand_forbidden)
        mask, _ = clash_mask(receptor_forbidden, lig_forb_rot, threshold_clash, dtype=dtype)
        V = score_volumes(receptor_volumes, lig_rot, W1, b1, W2, b2, clip=clip, dtype=dtype, clip_mode=clip_mode)
        V = (mask * V)[0].contiguous()
        if return_V:
            Vs.append(V.clone())
        if faithful_topk:
            top_list = update_top(top_list, V, int(ri), max_conf)
        else:
            idx, sc = rotation_picks_fast(V.numpy(), max_conf)
            x, y, z = flat_to_xyz(idx, N)
            for a in range(len(idx)):
                top_list.append((int(ri), int(x[a]), int(y[a]), int(z[a]), float(sc[a])))
            top_list.sort(key=lambda t: t[4])
            top_list = top_list[:max_conf]
    return (top_list, Vs) if return_V else top_list
