"""Atom-level front end: the pieces of TorchProteinLibrary the reference's ``Docker`` calls around
the hot loop, as one ``CoordsBackend`` object for ``Docker(...,
coords_backend=...)``:

    PDB2CoordsUnordered, Coords2TypedCoords, getBBox      /root/reference/src/Docker/Docker.py:37-38,51-54
    CoordsTranslate, CoordsRotate                          Docker.py:29-30,39,59,197-201,221-222
    TypedCoords2Volume(box_size, resolution)               Docker.py:31,204,208,223

TorchProteinLibrary's source is not available, so everything here is BUILD-DEFINED and documented
(parity unpinned): the PDB reader keeps heavy ATOM records in file order; the 11 atom types follow
Derevyanko et al. 2018 (Bioinformatics 34:4046, Table 1); the density is the Gaussian splat of
``csrc/dlpd_atoms.hip``.  Coordinates are padded (B, 3*Nmax) float64 on the CPU like TPL's; the
projection runs on the GPU and can apply a per-batch rotation on the fly, which replaces the
reference's per-iteration CPU rotate + H2D copy (Docker.py:221-223).
"""
import numpy as np
import torch

from deeplocalproteindocking_amd._lib import get_lib
from deeplocalproteindocking_amd.engine import _ptr, _stream

NUM_ATOM_TYPES = 11

_AROMATIC_C = {
    "HIS": {"CG", "CD2", "CE1"}, "PHE": {"CG", "CD1", "CD2", "CE1", "CE2", "CZ"},
    "TRP": {"CG", "CD1", "CD2", "CE2", "CE3", "CZ2", "CZ3", "CH2"},
    "TYR": {"CG", "CD1", "CD2", "CE1", "CE2", "CZ"},
}
_SP2_C = {"ARG": {"CZ"}, "ASN": {"CG"}, "ASP": {"CG"}, "GLN": {"CD"}, "GLU": {"CD"}}


def atom_type(resname, atomname):
    """0-based type index (0..10) or -1 for atoms that are skipped (hydrogens, unknown)."""
    r, a = resname.strip().upper(), atomname.strip().upper()
    if not a or a[0] == "H" or (a[0].isdigit() and len(a) > 1 and a[1] == "H"):
        return -1
    if a == "OXT":
        return 7
    e = a[0]
    if (r == "CYS" and a == "SG") or (r == "MET" and a == "SD") or (r == "MSE" and a == "SE"):
        return 0                                   # sulfur / selenium
    if e == "N":
        if a == "N" or (r == "ASN" and a == "ND2") or (r == "GLN" and a == "NE2"):
            return 1                               # amide N
        if (r == "HIS" and a in ("ND1", "NE2")) or (r == "TRP" and a == "NE1"):
            return 2                               # aromatic N
        if r == "ARG" and a in ("NE", "NH1", "NH2"):
            return 3                               # guanidinium N
        if r == "LYS" and a == "NZ":
            return 4                               # ammonium N
        return 1
    if e == "O":
        if a == "O" or (r == "ASN" and a == "OD1") or (r == "GLN" and a == "OE1"):
            return 5                               # carbonyl O
        if (r == "SER" and a == "OG") or (r == "THR" and a == "OG1") or (r == "TYR" and a == "OH"):
            return 6                               # hydroxyl O
        if (r == "ASP" and a in ("OD1", "OD2")) or (r == "GLU" and a in ("OE1", "OE2")):
            return 7                               # carboxyl O
        return 5
    if e == "C":
        if a == "C" or a in _SP2_C.get(r, ()):
            return 8                               # sp2 C
        if a in _AROMATIC_C.get(r, ()):
            return 9                               # aromatic C
        return 10                                  # sp3 C
    if e == "S":
        return 0
    return -1


_HEAVY_ELEMENTS = {"C", "N", "O", "S", "SE", "P"}


def _residue_number(field):
    """Columns 23-26: a decimal number; files with more than 9,999 residues carry hybrid-36 there ("A000" = 10000):
    decoded, and anything unreadable becomes 0 (the number only labels the atom, Docker never computes with it)."""
    f = field.strip()
    try:
        return int(f)
    except ValueError:
        pass
    digits = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"
    if len(f) == 4 and f[0].isalpha() and all(ch.upper() in digits for ch in f):
        up = f[0].isupper()
        v = 0
        for ch in f.upper():
            v = v * 36 + digits.index(ch)
        return v - 10 * 36 ** 3 + 10 ** 4 + (0 if up else 26 * 36 ** 3)
    return 0


def read_pdb_atoms(filename):
    """ATOM records -> (xyz float64 (n,3), chains, resnames, resnums, atomnames); first model only,
    alternate locations other than ' '/'A' dropped.

    What real PDB entries have and the synthetic test files may not: CRLF line ends and lines without trailing
    blanks (records are padded to 80 columns before slicing), TER / END / CONECT / ANISOU / HETATM records (skipped: the
    reference types protein atoms only), hybrid-36 serial and residue numbers, blank chain identifiers, an element
    column (77-78) that is present, blank or absent -- when it names an element the record is kept only for C, N, O, S,
    Se, P (hydrogen and deuterium records are dropped here, whatever their atom name looks like; metals written as
    ATOM too), otherwise the atom name decides (``atom_type``).  A file without a usable ATOM record, a record with a
    non-numeric or non-finite coordinate: ValueError naming the file and line."""
    xyz, chains, resnames, resnums, atomnames = [], [], [], [], []
    with open(filename, newline=None, errors="replace") as fin:
        for lineno, raw in enumerate(fin, 1):
            line = raw.rstrip("\r\n")
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if not rec.startswith("ATOM"):
                continue
            line = line.ljust(80)
            if line[16] not in (" ", "A"):
                continue
            elem = line[76:78].strip().upper()
            if elem.isalpha() and elem not in _HEAVY_ELEMENTS:
                continue                                        # H, D, metals
            try:
                pos = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
            except ValueError:
                raise ValueError("%s:%d: ATOM record without numeric coordinates in columns 31-54" % (filename, lineno))
            if not all(np.isfinite(pos)):
                raise ValueError("%s:%d: non-finite coordinate" % (filename, lineno))
            atomnames.append(line[12:16].strip())
            resnames.append(line[17:20].strip())
            chains.append(line[21])
            resnums.append(_residue_number(line[22:26]))
            xyz.append(pos)
    if not xyz:
        raise ValueError("%s: no ATOM records" % filename)
    return np.asarray(xyz, dtype=np.float64).reshape(-1, 3), chains, resnames, resnums, atomnames


class CoordsBackend:
    """Plug-in for ``Docker(coords_backend=...)``; method names follow Docker's calls."""

    def __init__(self, lib=None, splat=None, atom_types=None):
        self.lib = lib                 # None -> the product library (GPU); tests pass the emulated one
        # density shape of the projection (build-defined; Utils/Conventions.py): sigma, window, voxel_offset, norm
        self.splat = dict({"sigma": 1.0, "window": 2, "voxel_offset": 0.0, "norm": 1.0}, **(splat or {}))
        # optional typing table {"RES:ATOM": type 0..10 | -1} that overrides ``atom_type`` entry by entry
        # (scripts/calibrate_tpl.py reads TorchProteinLibrary's own assignment off a probe structure)
        self.atom_types = dict(atom_types or {})

    def type_of(self, resname, atomname):
        key = "%s:%s" % (resname.strip().upper(), atomname.strip().upper())
        if key in self.atom_types:
            return int(self.atom_types[key])
        return atom_type(resname, atomname)

    # ---- PDB2CoordsUnordered (Docker.py:51)
    def pdb2coords(self, filenames):
        recs = [read_pdb_atoms(f) for f in filenames]
        nmax = max(len(r[0]) for r in recs)
        coords = torch.zeros(len(recs), 3 * nmax, dtype=torch.double)
        num_atoms = torch.zeros(len(recs), dtype=torch.int32)
        for b, r in enumerate(recs):
            n = len(r[0])
            coords[b, :3 * n] = torch.from_numpy(r[0].reshape(-1))
            num_atoms[b] = n
        return (coords, [r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs], [r[4] for r in recs],
                num_atoms)

    # ---- Coords2TypedCoords (Docker.py:52): reorder by type, drop untyped atoms
    def assign_types(self, coords, resnames, atomnames, num_atoms):
        B = coords.shape[0]
        typed, counts = [], torch.zeros(B, NUM_ATOM_TYPES, dtype=torch.int32)
        for b in range(B):
            n = int(num_atoms[b])
            xyz = coords[b, :3 * n].reshape(n, 3)
            ty = np.array([self.type_of(resnames[b][i], atomnames[b][i]) for i in range(n)], dtype=np.int64)
            order = [np.nonzero(ty == t)[0] for t in range(NUM_ATOM_TYPES)]
            for t in range(NUM_ATOM_TYPES):
                counts[b, t] = len(order[t])
            idx = np.concatenate(order) if n else np.zeros(0, dtype=np.int64)
            typed.append(xyz[torch.from_numpy(idx)])
        nmax = max(t.shape[0] for t in typed)
        out = torch.zeros(B, 3 * nmax, dtype=torch.double)
        for b, t in enumerate(typed):
            out[b, :3 * t.shape[0]] = t.reshape(-1)
        offsets = torch.cumsum(counts, dim=1).to(torch.int32) - counts
        self.last_num_typed = counts.sum(dim=1).to(torch.int32)
        return out, counts, offsets

    # ---- getBBox (Docker.py:54)
    def get_bbox(self, coords, num_atoms):
        B = coords.shape[0]
        a = torch.zeros(B, 3, dtype=torch.double)
        b_ = torch.zeros(B, 3, dtype=torch.double)
        for b in range(B):
            n = int(num_atoms[b])
            xyz = coords[b, :3 * n].reshape(n, 3)
            a[b], b_[b] = xyz.min(dim=0).values, xyz.max(dim=0).values
        return a, b_

    # ---- CoordsTranslate / CoordsRotate (Docker.py:59,197-201,221-222); padded slots stay 0
    def translate(self, coords, T, num_atoms):
        out = coords.clone()
        for b in range(coords.shape[0]):
            n = int(num_atoms[b])
            out[b, :3 * n] = (coords[b, :3 * n].reshape(n, 3) + T[b if T.shape[0] > 1 else 0]).reshape(-1)
        return out

    def rotate(self, coords, R, num_atoms):
        out = coords.clone()
        for b in range(coords.shape[0]):
            n = int(num_atoms[b])
            Rb = R[b if R.shape[0] > 1 else 0].to(torch.double)
            out[b, :3 * n] = (coords[b, :3 * n].reshape(n, 3) @ Rb.t()).reshape(-1)
        return out

    # ---- TypedCoords2Volume (Docker.py:204,208,223), optionally p' = R p + shift on the fly
    def to_device(self, coords, num_atoms_of_type, offsets, device):
        return (coords.to(device=device, dtype=torch.float32).contiguous(),
                num_atoms_of_type.to(device=device, dtype=torch.int32).contiguous(),
                offsets.to(device=device, dtype=torch.int32).contiguous())

    def project(self, coords, num_atoms_of_type, offsets, box_size, resolution, device, R=None, shift=None,
                sum_types=False, lib=None, cells=False):
        """coords (B or 1, 3*Nmax); R (nb,3,3) f32 device -> one volume set per rotation of the
        SAME atoms (B must be 1 then).  Returns (nb, 11 or 1, L, L, L) float32 on ``device``.
        cells (all types only): the CELL-WISE projection -- only the 4 x 4 x 4 cells the atoms' windows reach are cleared,
        accumulated and converted; the result carries their map (``.dlpd_occupancy``, uint8 (nb, ceil(L/4)^3)) and is NOT
        WRITTEN elsewhere (``.dlpd_unwritten``): for consumers that go by the map (the E3 plugin inside outputs_with_maps)."""
        device = torch.device(device)
        lib = lib or self.lib or get_lib()
        ready = (coords.device == device and coords.dtype == torch.float32 and
                 num_atoms_of_type.dtype == torch.int32 and offsets.dtype == torch.int32)
        c, nt, of = (coords, num_atoms_of_type, offsets) if ready else \
            self.to_device(coords, num_atoms_of_type, offsets, device)
        nb = c.shape[0]
        stride = c.shape[1] // 3
        if R is not None:
            nb = R.shape[0]
            if c.shape[0] == 1 and nb > 1:
                c, nt, of = c.expand(nb, -1).contiguous(), nt.expand(nb, -1).contiguous(), of.expand(nb, -1).contiguous()
            R = R.to(device=device, dtype=torch.float32).contiguous()
        sx, sy, sz = (0.0, 0.0, 0.0) if shift is None else [float(v) for v in torch.as_tensor(shift).reshape(-1)[:3]]
        nch = 1 if sum_types else NUM_ATOM_TYPES
        out = torch.empty(nb, nch, box_size, box_size, box_size, dtype=torch.float32, device=device)
        if cells:
            if sum_types:
                raise RuntimeError("dlpd: the cell-wise projection keeps the atom types apart (sum_types=False)")
            nc = (box_size + 3) // 4
            occ = torch.empty(nb, nc, nc, nc, dtype=torch.uint8, device=device)
            lib.call("dlpd_project_atoms_cells", _ptr(c), _ptr(nt), _ptr(of), _ptr(R), sx, sy, sz, _ptr(out), _ptr(occ), nb, stride,
                     NUM_ATOM_TYPES, box_size, float(resolution), float(self.splat["sigma"]), int(self.splat["window"]),
                     float(self.splat["voxel_offset"]), float(self.splat["norm"]), _stream(device))
            out.dlpd_occupancy, out.dlpd_unwritten = occ, True
            return out
        lib.call("dlpd_project_atoms_ext", _ptr(c), _ptr(nt), _ptr(of), _ptr(R), sx, sy, sz, _ptr(out), nb, stride,
                 NUM_ATOM_TYPES, box_size, float(resolution), int(sum_types), float(self.splat["sigma"]),
                 int(self.splat["window"]), float(self.splat["voxel_offset"]), float(self.splat["norm"]), _stream(device))
        return out
