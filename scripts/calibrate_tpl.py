#!/usr/bin/env python3
"""Pin the conventions of TorchProteinLibrary's operators -- run by a maintainer who HAS a TorchProteinLibrary build
(README.md:5 of the reference pins commit 16166ce4; PyTorch 1.1), never in this repository's own CI: the library is
absent from the reference tree and from the build machines, which is why the rotation pivot / scale / axis order, the
meaning of VolumeConvolution(clip), the density splat and the atom typing are build-defined here.

    python scripts/calibrate_tpl.py --out tests/golden/tpl_conventions.json [--device cuda] [--module TorchProteinLibrary]

It feeds tiny deterministic volumes and atoms through the library's

    Volume.VolumeRotation()(volume, R)                      reference call: src/Docker/Docker.py:40,218
    Volume.VolumeConvolution(clip)(v1, v2)                  src/Models/DockingModels.py:48,71 ; Docker.py:32,225
    Volume.TypedCoords2Volume(box, res)(coords, n, offs)    Docker.py:31,204,208,223
    FullAtomModel.PDB2CoordsUnordered / Coords2TypedCoords  Docker.py:37-38,51-52

compares the outputs with a small family of candidate conventions (implemented HERE in numpy, independently of the
product kernels and of the test oracle), prints which candidate reproduces each operator and by what margin, and writes a
JSON holding (i) the conventions in the vocabulary of deeplocalproteindocking_amd/Utils/Conventions.py -- pass the file to
``Docker(model, ..., conventions=path)`` -- and (ii) the probe inputs and the library's outputs, which
tests/test_tpl_conventions.py replays through the HIP kernels (``-m gpu``) once the file is committed under
tests/golden/.  Exit code 0: every operator identified unambiguously; 1: at least one was not (the table says which).

``--module`` names the package to import (default TorchProteinLibrary); the CPU test of this script passes a stand-in
package with known conventions.
"""
import argparse
import importlib
import itertools
import json
import os
import sys
import tempfile

import numpy as np


# ----------------------------------------------------------------------------------------------------------------------
# candidate implementations (numpy, float64)
# ----------------------------------------------------------------------------------------------------------------------

def euler_zyz(phi, theta, psi):
    """A generic proper rotation Rz(phi) Ry(theta) Rz(psi) (any rotation without a symmetry axis along a grid axis will do)."""
    c, s = np.cos, np.sin
    rz1 = np.array([[c(phi), -s(phi), 0], [s(phi), c(phi), 0], [0, 0, 1]])
    ry = np.array([[c(theta), 0, s(theta)], [0, 1, 0], [-s(theta), 0, c(theta)]])
    rz2 = np.array([[c(psi), -s(psi), 0], [s(psi), c(psi), 0], [0, 0, 1]])
    return rz1 @ ry @ rz2


def rotate_candidate(vol, R, center, scale, axis_order, transpose):
    """out(i) = vol(c + s M^T (i - c)), trilinear, zeros outside; vol (L,L,L)."""
    L = vol.shape[0]
    M = R.T if transpose else R
    if axis_order == "zyx":
        M = M[::-1, ::-1]
    ar = np.arange(L, dtype=np.float64) - center
    d = np.stack(np.meshgrid(ar, ar, ar, indexing="ij"), axis=-1).reshape(-1, 3)
    p = scale * (d @ M) + center
    p0 = np.floor(p)
    f = p - p0
    i0 = p0.astype(np.int64)
    out = np.zeros(L ** 3)
    flat = vol.reshape(-1)
    for dx, dy, dz in itertools.product((0, 1), repeat=3):
        w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
        ix, iy, iz = i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz
        ok = (ix >= 0) & (ix < L) & (iy >= 0) & (iy < L) & (iz >= 0) & (iz < L)
        idx = (np.clip(ix, 0, L - 1) * L + np.clip(iy, 0, L - 1)) * L + np.clip(iz, 0, L - 1)
        out += np.where(ok, flat[idx] * w, 0.0)
    return out.reshape(L, L, L)


CENTERS = {"L/2": lambda L: L / 2.0, "grid_sample": lambda L: (L - 1) / 2.0, "L/2-1": lambda L: L / 2.0 - 1.0}
SCALES = {"1": lambda L: 1.0, "(L-1)/L": lambda L: (L - 1.0) / L, "L/(L-1)": lambda L: L / (L - 1.0)}


def correlate_candidate(v1, v2, clip, mode, swapped):
    """(2L)^3 circular cross-correlation out[t mod 2L] = sum_r v1[r + t] v2[r] (src/Models/MultiplyVolumes.py:13-47);
    swapped: the roles of the two arguments exchanged."""
    if swapped:
        v1, v2 = v2, v1
    if clip is not None and mode == "input":
        v1, v2 = np.clip(v1, -clip, clip), np.clip(v2, -clip, clip)
    N = 2 * v1.shape[0]
    ax = (0, 1, 2)
    out = np.fft.irfftn(np.fft.rfftn(v1, s=(N, N, N), axes=ax) * np.conj(np.fft.rfftn(v2, s=(N, N, N), axes=ax)), s=(N, N, N), axes=ax)
    if clip is not None and mode == "output":
        out = np.clip(out, -clip, clip)
    return out


def splat_candidate(p, L, res, sigma, window, voxel_offset, norm):
    out = np.zeros((L, L, L))
    c = np.floor(p / res - voxel_offset).astype(int)
    for i in range(c[0] - window, c[0] + window + 1):
        for j in range(c[1] - window, c[1] + window + 1):
            for k in range(c[2] - window, c[2] + window + 1):
                if 0 <= i < L and 0 <= j < L and 0 <= k < L:
                    d = p - (np.array([i, j, k]) + voxel_offset) * res
                    out[i, j, k] += norm * np.exp(-0.5 * (d @ d) / sigma ** 2)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# probes
# ----------------------------------------------------------------------------------------------------------------------

def smooth_volume(L, seed):
    """A few off-centre Gaussian blobs: smooth (trilinear-friendly), no symmetry that could hide an axis swap."""
    rng = np.random.RandomState(seed)
    ar = np.arange(L, dtype=np.float64)
    x, y, z = np.meshgrid(ar, ar, ar, indexing="ij")
    v = np.zeros((L, L, L))
    for _ in range(4):
        c = rng.uniform(0.25 * L, 0.75 * L, size=3)
        w = rng.uniform(0.12 * L, 0.22 * L)
        v += rng.uniform(0.5, 1.5) * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * w * w))
    return v


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def decide(errors, what, tol=2e-4, margin=20.0):
    """errors: {candidate key: relative error} -> (best key | None, report lines)."""
    ranked = sorted(errors.items(), key=lambda kv: kv[1])
    best, second = ranked[0], (ranked[1] if len(ranked) > 1 else (None, float("inf")))
    ok = best[1] <= tol and second[1] >= margin * max(best[1], 1e-7)
    lines = ["%s: best %s (error %.2e), runner-up %s (%.2e) -> %s" % (
        what, best[0], best[1], second[0], second[1], "IDENTIFIED" if ok else "NOT identified")]
    return (best[0] if ok else None), lines, ranked


def to_torch(a, torch, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=device, dtype=dtype or torch.float32)


def calibrate_rotation(tpl, torch, device):
    op = tpl.Volume.VolumeRotation()
    report, per_L = [], {}
    probes = []
    R = euler_zyz(0.7, 0.5, -0.4)
    for L in (12, 9):
        vol = smooth_volume(L, 100 + L)
        out = op(to_torch(vol[None, None], torch, device), to_torch(R[None], torch, device)).detach().cpu().numpy()[0, 0].astype(np.float64)
        probes.append({"L": L, "volume": vol.tolist(), "R": R.tolist(), "out": out.tolist()})
        errs = {}
        for (cn, cf), (sn, sf), ax, tr in itertools.product(CENTERS.items(), SCALES.items(), ("xyz", "zyx"), (False, True)):
            errs[(cn, sn, ax, tr)] = rel_err(rotate_candidate(vol, R, cf(L), sf(L), ax, tr), out)
        per_L[L] = errs
    # a candidate must explain BOTH grid sizes (this is what tells a constant scale from an L-dependent rule)
    combined = {k: max(per_L[12][k], per_L[9][k]) for k in per_L[12]}
    best, lines, ranked = decide(combined, "VolumeRotation (center, scale, axis order, transposed)")
    report += lines
    conv = None
    if best is not None:
        cn, sn, ax, tr = best
        # every candidate pivot is a rule of Utils/Conventions.rotation_pivot ("L/2" is its default, None)
        conv = {"rotation_center": None if cn == "L/2" else cn,
                "rotation_scale": None if sn == "1" else sn, "rotation_axis_order": ax, "rotation_transpose": bool(tr)}
    return conv, report, probes, [("%s|%s|%s|%s" % k, v) for k, v in ranked[:6]]


def calibrate_convolution(tpl, torch, device):
    L, clip = 6, 0.8
    rng = np.random.RandomState(7)
    v1, v2 = rng.randn(L, L, L), rng.randn(L, L, L) * 0.7
    t1, t2 = to_torch(v1[None, None], torch, device), to_torch(v2[None, None], torch, device)
    out_clip = tpl.Volume.VolumeConvolution(clip=clip)(t1, t2).detach().cpu().numpy()[0, 0].astype(np.float64)
    out_plain = tpl.Volume.VolumeConvolution()(t1, t2).detach().cpu().numpy()[0, 0].astype(np.float64)
    report = []
    errs = {("plain", sw): rel_err(correlate_candidate(v1, v2, None, "none", sw), out_plain) for sw in (False, True)}
    arg, lines, _ = decide(errs, "VolumeConvolution() index / argument convention")
    report += lines
    swapped = bool(arg[1]) if arg is not None else False
    errs = {m: rel_err(correlate_candidate(v1, v2, clip, m, swapped), out_clip) for m in ("output", "input", "none")}
    mode, lines, ranked = decide(errs, "VolumeConvolution(clip)")
    report += lines
    probe = {"L": L, "clip": clip, "v1": v1.tolist(), "v2": v2.tolist(), "out_clip": out_clip.tolist(), "out_plain": out_plain.tolist()}
    conv = None
    if mode is not None and arg is not None:
        conv = {"clip_mode": mode}
        if swapped:
            # not representable in Utils/Conventions.py (the product follows MultiplyVolumes.py:13-47, which the reference's own
            # code pins): report it as NOT identified rather than writing a file that would run with unswapped arguments
            report.append("  the library correlates with its two arguments in the OPPOSITE roles to MultiplyVolumes.py:13-47: "
                          "this build has no such convention -> VolumeConvolution NOT identified")
            conv = None
    return conv, report, probe, [(str(k), v) for k, v in ranked]


def calibrate_splat(tpl, torch, device):
    L, res, T = 12, 1.25, 11
    op = tpl.Volume.TypedCoords2Volume(L, res)
    report, probes, fits = [], [], []
    for pos in ([7.3, 6.1, 8.45], [5.0, 9.9, 6.6]):
        p = np.array(pos)
        coords = np.zeros((1, 3), dtype=np.float64)
        coords[0] = p
        num = np.zeros((1, T), dtype=np.int32)
        num[0, 3] = 1
        offs = np.zeros((1, T), dtype=np.int32)
        vol = op(to_torch(coords, torch, device, torch.double), to_torch(num, torch, device, torch.int32),
                 to_torch(offs, torch, device, torch.int32)).detach().cpu().numpy()[0].astype(np.float64)
        if np.abs(vol[:3]).max() + np.abs(vol[4:]).max() > 0:
            report.append("TypedCoords2Volume: density found outside the atom's own type channel")
        d = vol[3]
        nz = np.argwhere(d > 0)
        if len(nz) < 8:
            report.append("TypedCoords2Volume: fewer than 8 non-zero voxels -- cannot fit")
            return None, report, probes, []
        ext = nz.max(axis=0) - nz.min(axis=0) + 1
        best = None
        for voff in (0.0, 0.5):
            r2 = np.array([np.sum((p - (ijk + voff) * res) ** 2) for ijk in nz])
            y = np.log(d[tuple(nz.T)])
            A = np.stack([r2, np.ones_like(r2)], axis=1)
            sol, resid = np.linalg.lstsq(A, y, rcond=None)[:2]
            rms = float(np.sqrt(np.mean((A @ sol - y) ** 2)))
            if best is None or rms < best[0]:
                best = (rms, voff, sol)
        rms, voff, (slope, icpt) = best
        sigma, norm = float(np.sqrt(-0.5 / slope)), float(np.exp(icpt))
        window = int((ext.max() - 1) // 2)
        fits.append((sigma, norm, voff, window, rms))
        probes.append({"L": L, "resolution": res, "position": pos, "type": 3, "volume": d.tolist()})
    sig, nrm = np.mean([f[0] for f in fits]), np.mean([f[1] for f in fits])
    voff, window = fits[0][2], max(f[3] for f in fits)
    conv = {"splat": {"sigma": round(float(sig), 5), "window": window, "voxel_offset": voff, "norm": round(float(nrm), 5)}}
    # verify the fitted shape against the library's volumes, window included
    worst = 0.0
    for pr in probes:
        cand = splat_candidate(np.array(pr["position"]), L, res, conv["splat"]["sigma"], window, voff, conv["splat"]["norm"])
        worst = max(worst, rel_err(cand, np.array(pr["volume"])))
    ok = worst <= 1e-3 and all(abs(f[0] - sig) < 1e-3 * sig and f[2] == voff for f in fits)
    report.append("TypedCoords2Volume: Gaussian fit sigma %.4f, prefactor %.4f, voxel offset %.1f, window %d (log-fit rms %.1e); "
                  "refit error %.2e -> %s" % (sig, nrm, voff, window, max(f[4] for f in fits), worst,
                                              "IDENTIFIED" if ok else "NOT identified (not a truncated Gaussian of this family)"))
    return (conv if ok else None), report, probes, [("refit", worst)]


RESIDUES = {
    "GLY": ["N", "CA", "C", "O"], "ALA": ["N", "CA", "C", "O", "CB"], "SER": ["N", "CA", "C", "O", "CB", "OG"],
    "CYS": ["N", "CA", "C", "O", "CB", "SG"], "VAL": ["N", "CA", "C", "O", "CB", "CG1", "CG2"],
    "LEU": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2"], "ILE": ["N", "CA", "C", "O", "CB", "CG1", "CG2", "CD1"],
    "THR": ["N", "CA", "C", "O", "CB", "OG1", "CG2"], "MET": ["N", "CA", "C", "O", "CB", "CG", "SD", "CE"],
    "PRO": ["N", "CA", "C", "O", "CB", "CG", "CD"], "LYS": ["N", "CA", "C", "O", "CB", "CG", "CD", "CE", "NZ"],
    "ASP": ["N", "CA", "C", "O", "CB", "CG", "OD1", "OD2"], "GLU": ["N", "CA", "C", "O", "CB", "CG", "CD", "OE1", "OE2"],
    "ASN": ["N", "CA", "C", "O", "CB", "CG", "OD1", "ND2"], "GLN": ["N", "CA", "C", "O", "CB", "CG", "CD", "OE1", "NE2"],
    "ARG": ["N", "CA", "C", "O", "CB", "CG", "CD", "NE", "CZ", "NH1", "NH2"],
    "HIS": ["N", "CA", "C", "O", "CB", "CG", "ND1", "CD2", "CE1", "NE2"],
    "PHE": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ"],
    "TYR": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ", "OH"],
    "TRP": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "NE1", "CE2", "CE3", "CZ2", "CZ3", "CH2"],
}


def calibrate_typing(tpl, torch):
    """One residue of each kind, every heavy atom at a unique position: the library's typed coordinates are grouped by
    type, so each atom's type can be read back by matching positions."""
    fa = getattr(tpl, "FullAtomModel", None)
    if fa is None or not hasattr(fa, "PDB2CoordsUnordered") or not hasattr(fa, "Coords2TypedCoords"):
        return None, ["Coords2TypedCoords: FullAtomModel not available in this module -- typing table not read"], None
    lines, where, serial = [], {}, 1
    for ri, (resn, atoms) in enumerate(sorted(RESIDUES.items())):
        for ai, name in enumerate(atoms):
            xyz = (3.0 * ri + 0.137 * ai, 1.5 * ai + 0.211 * ri, 0.731 * ai - 0.5 * ri)
            where[tuple(np.round(xyz, 3))] = (resn, name)
            name4 = (" " + name) if len(name) < 4 else name
            lines.append("ATOM  %5d %-4s %3s A%4d    %8.3f%8.3f%8.3f  1.00 20.00          %2s" % (
                serial, name4, resn, ri + 1, xyz[0], xyz[1], xyz[2], name[0]))
            serial += 1
    lines.append("END")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "typing_probe.pdb")
        with open(path, "w") as f:
            f.write("\n".join(lines) + "\n")
        coords, chains, resnames, resnums, atomnames, num_atoms = fa.PDB2CoordsUnordered()([path])
        typed, num_of_type, offsets = fa.Coords2TypedCoords()(coords, resnames, atomnames, num_atoms)
    typed = typed.detach().cpu().numpy().reshape(-1, 3)
    num_of_type = np.asarray(num_of_type.detach().cpu().numpy()).reshape(-1)
    offsets = np.asarray(offsets.detach().cpu().numpy()).reshape(-1)
    table, seen = {}, 0
    for t in range(len(num_of_type)):
        for a in range(int(offsets[t]), int(offsets[t]) + int(num_of_type[t])):
            key = tuple(np.round(typed[a], 3))
            if key in where:
                table["%s:%s" % where[key]] = t
                seen += 1
    for (resn, name) in where.values():
        table.setdefault("%s:%s" % (resn, name), -1)              # atoms the library drops
    return {"atom_types": table}, ["Coords2TypedCoords: %d of %d probe atoms typed into %d types" % (
        seen, len(where), len(num_of_type))], {"pdb": lines}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default="tpl_conventions.json")
    ap.add_argument("--module", default="TorchProteinLibrary")
    ap.add_argument("--device", default=None, help="default: cuda if the library's volume ops need it and it is available, else cpu")
    args = ap.parse_args(argv)
    import torch
    tpl = importlib.import_module(args.module)
    for sub in ("Volume", "FullAtomModel"):
        if not hasattr(tpl, sub):
            try:
                setattr(tpl, sub, importlib.import_module(args.module + "." + sub))
            except ImportError:
                pass
    device = args.device or ("cuda" if torch.cuda.is_available() else "cpu")
    conventions, evidence, probes, report, all_ok = {}, {}, {}, [], True
    for name, fn in (("rotation", lambda: calibrate_rotation(tpl, torch, device)),
                     ("convolution", lambda: calibrate_convolution(tpl, torch, device)),
                     ("splat", lambda: calibrate_splat(tpl, torch, device))):
        conv, lines, probe, ranked = fn()
        report += lines
        probes[name] = probe
        evidence[name] = {"candidates": ranked, "identified": conv is not None}
        if conv is None:
            all_ok = False
        else:
            conventions.update(conv)
    conv, lines, probe = calibrate_typing(tpl, torch)
    report += lines
    if conv is not None:
        conventions.update(conv)
        probes["typing"] = probe
    print("\n".join(report))
    print("\nconventions for Docker(..., conventions=%r):" % args.out)
    print(json.dumps({k: v for k, v in conventions.items() if k != "atom_types"}, indent=1))
    with open(args.out, "w") as f:
        json.dump({"conventions": conventions, "evidence": evidence, "probes": probes, "module": args.module,
                   "torch": torch.__version__, "all_identified": all_ok}, f)
    print("written: %s (%s)" % (args.out, "every operator identified" if all_ok else "SOME OPERATORS NOT IDENTIFIED"))
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
