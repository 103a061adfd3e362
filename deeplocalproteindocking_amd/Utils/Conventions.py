"""Conventions of the TorchProteinLibrary operators the reference calls around the hot path, as PARAMETERS.

TorchProteinLibrary (pinned at README.md:5; call sites src/Docker/Docker.py:29-40,218,221-225 and
src/Models/DockingModels.py:48,71) is not in the reference tree, so what its VolumeRotation, VolumeConvolution(clip) and
TypedCoords2Volume do at the level of a voxel index is build-defined here ("parity unpinned").  Every
convention that can plausibly differ is a parameter of the kernels / operator objects, and this module is where a
maintainer who HAS a TorchProteinLibrary build plugs in what ``scripts/calibrate_tpl.py`` found there:

    conv = VolumeConventions.load("tpl_conventions.json")      # written by scripts/calibrate_tpl.py
    Docker(model, ..., conventions=conv)

    rotation_center      pivot of the trilinear volume rotation: None = index L/2; "grid_sample" = (L-1)/2; "L/2-1" = one
                         voxel below the centre; a number = that index on the fine grid (scaled to coarser grids)
    rotation_scale       stretch of the sample offset: None / 1.0; "(L-1)/L" or "L/(L-1)" (grid_sample's normalised
                         coordinates generated with one align_corners convention and sampled with the other); a number
    rotation_axis_order  "xyz": axis 0 of the rotation matrix <-> first spatial index; "zyx": <-> last
    rotation_transpose   False: out(i) = vol(c + s R^T (i - c)) (the volume turns WITH the atoms rotated by R,
                         Docker.py:221-223); True: the inverse rotation
    clip_mode            VolumeConvolution(clip): "output" clamps the correlation, "input" clamps both input volumes,
                         "none" ignores the argument
    splat                TypedCoords2Volume: {"sigma": 1.0, "window": 2, "voxel_offset": 0.0, "norm": 1.0}:
                         norm * exp(-|r|^2 / (2 sigma^2)) on the (2 window + 1)^3 voxels around the atom, voxel i at
                         (i + voxel_offset) * resolution
    atom_types           optional {"RES:ATOM": type} table overriding Utils/FullAtom.atom_type (Coords2TypedCoords)

The kernels take the rotation as a general 3x3 map, out(i) = vol(c + M^T (i - c)): scale and axis order are folded into
M here (``kernel_matrices``), so they cost nothing in the hot loop; the tests' CPU restatement of the reference implements
them explicitly (not folded into the matrix) and the two are compared.
"""
import json

import torch

CLIP_MODES = ("output", "input", "none")
DEFAULT_SPLAT = {"sigma": 1.0, "window": 2, "voxel_offset": 0.0, "norm": 1.0}


def rotation_scale(rule, L):
    if rule is None:
        return 1.0
    if isinstance(rule, str):
        if rule == "(L-1)/L":
            return (float(L) - 1.0) / float(L)
        if rule == "L/(L-1)":
            return float(L) / (float(L) - 1.0)
        raise Exception("Unknown rotation_scale", rule)
    return float(rule)


def rotation_pivot(center, L, fine_L=None):
    """Pivot index on a grid of L voxels per edge; a numeric ``center`` is given on the grid of ``fine_L`` voxels."""
    if center is None:
        return float(L) / 2.0
    if isinstance(center, str):
        if center == "grid_sample":
            return (float(L) - 1.0) / 2.0
        if center == "L/2-1":
            return float(L) / 2.0 - 1.0
        raise Exception("Unknown rotation_center", center)
    return float(center) * float(L) / float(fine_L or L)


def kernel_matrices(R, scale=1.0, axis_order="xyz", transpose=False):
    """(n,3,3) rotation matrices -> the 3x3 maps M the kernels sample with, out(i) = vol(c + M^T (i - c)):
    M = scale * P R P, P the axis reversal for "zyx" (identity for "xyz"); R^T in place of R for ``transpose``."""
    if axis_order not in ("xyz", "zyx"):
        raise Exception("Unknown rotation_axis_order", axis_order)
    M = R.transpose(-1, -2) if transpose else R
    if axis_order == "zyx":
        M = M.flip(-1).flip(-2)
    if float(scale) != 1.0:
        M = M * float(scale)
    return M.contiguous()


class VolumeConventions(object):
    def __init__(self, rotation_center=None, rotation_scale=None, rotation_axis_order="xyz", clip_mode="output", splat=None,
                 rotation_transpose=False, atom_types=None):
        if clip_mode not in CLIP_MODES:
            raise Exception("Unknown clip_mode", clip_mode)
        if rotation_axis_order not in ("xyz", "zyx"):
            raise Exception("Unknown rotation_axis_order", rotation_axis_order)
        self.rotation_center = rotation_center
        self.rotation_scale = rotation_scale
        self.rotation_axis_order = rotation_axis_order
        self.rotation_transpose = bool(rotation_transpose)
        self.clip_mode = clip_mode
        self.splat = dict(DEFAULT_SPLAT, **(splat or {}))
        self.atom_types = dict(atom_types or {})

    def is_default(self):
        return (self.rotation_center is None and self.rotation_scale in (None, 1.0) and self.rotation_axis_order == "xyz"
                and not self.rotation_transpose and self.clip_mode == "output" and self.splat == DEFAULT_SPLAT
                and not self.atom_types)

    def pivot(self, L, fine_L=None):
        return rotation_pivot(self.rotation_center, L, fine_L)

    def scale(self, L):
        return rotation_scale(self.rotation_scale, L)

    def matrices(self, R, L):
        return kernel_matrices(R, self.scale(L), self.rotation_axis_order, self.rotation_transpose)

    def to_dict(self):
        return {"rotation_center": self.rotation_center, "rotation_scale": self.rotation_scale,
                "rotation_axis_order": self.rotation_axis_order, "rotation_transpose": self.rotation_transpose,
                "clip_mode": self.clip_mode, "splat": dict(self.splat), "atom_types": dict(self.atom_types)}

    KEYS = ("rotation_center", "rotation_scale", "rotation_axis_order", "rotation_transpose", "clip_mode", "splat", "atom_types")

    def copy(self):
        return VolumeConventions.from_dict(self.to_dict())

    @classmethod
    def from_dict(cls, d):
        """A key this class cannot represent is an ERROR, never dropped: a calibration that found something else in the
        library (e.g. correlation arguments in the opposite roles) must not silently run with the defaults."""
        d = d.get("conventions", d)
        unknown = sorted(set(d) - set(cls.KEYS))
        if unknown:
            raise Exception("Unknown convention keys", unknown)
        return cls(rotation_center=d.get("rotation_center"), rotation_scale=d.get("rotation_scale"),
                   rotation_axis_order=d.get("rotation_axis_order", "xyz"), clip_mode=d.get("clip_mode", "output"),
                   splat=d.get("splat"), rotation_transpose=d.get("rotation_transpose", False), atom_types=d.get("atom_types"))

    @classmethod
    def load(cls, path):
        with open(path) as fin:
            return cls.from_dict(json.load(fin))

    def __repr__(self):
        return "VolumeConventions(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.to_dict().items()))
