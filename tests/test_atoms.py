"""Atom-level front end (SURVEY.md 8(f) rows 1,3): PDB reader, 11-type typing, projection kernel and
Docker.dockSE3 / dockE3 end to end.  CPU tests run the kernel sources under the emulator."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import docking_oracle as orc
from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend, NUM_ATOM_TYPES, atom_type, read_pdb_atoms

RES = {"GLY": ["N", "CA", "C", "O"], "ALA": ["N", "CA", "C", "O", "CB"],
       "SER": ["N", "CA", "C", "O", "CB", "OG"], "CYS": ["N", "CA", "C", "O", "CB", "SG"],
       "LYS": ["N", "CA", "C", "O", "CB", "CG", "CD", "CE", "NZ"],
       "ASP": ["N", "CA", "C", "O", "CB", "CG", "OD1", "OD2"],
       "ARG": ["N", "CA", "C", "O", "CB", "CG", "CD", "NE", "CZ", "NH1", "NH2"],
       "PHE": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ"],
       "HIS": ["N", "CA", "C", "O", "CB", "CG", "ND1", "CD2", "CE1", "NE2"],
       "ASN": ["N", "CA", "C", "O", "CB", "CG", "OD1", "ND2"]}


def write_fake_pdb(path, nres, seed, radius=10.0):
    """A random-coil 'protein': residues of mixed types scattered in a ball (geometry is irrelevant
    to the code under test), plus a hydrogen and a HETATM line that must be ignored."""
    rng = np.random.RandomState(seed)
    names = list(RES)
    lines, serial = [], 1
    for r in range(nres):
        rn = names[rng.randint(len(names))]
        centre = rng.normal(size=3) * radius / 2.0
        for an in RES[rn]:
            x, y, z = centre + rng.normal(size=3) * 1.5
            lines.append("ATOM  %5d %-4s %3s A%4d    %8.3f%8.3f%8.3f  1.00  0.00" % (
                serial, (" " + an) if len(an) < 4 else an, rn, r + 1, x, y, z))
            serial += 1
    lines.insert(3, "ATOM  %5d  H   GLY A   1    %8.3f%8.3f%8.3f  1.00  0.00" % (9999, 0.0, 0.0, 0.0))
    lines.append("HETATM%5d  O   HOH A 999    %8.3f%8.3f%8.3f  1.00  0.00" % (serial, 1.0, 1.0, 1.0))
    lines.append("END")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return serial - 1


def test_atom_typing_table():
    assert atom_type("CYS", "SG") == 0 and atom_type("MET", "SD") == 0
    assert atom_type("ALA", "N") == 1 and atom_type("ASN", "ND2") == 1
    assert atom_type("HIS", "NE2") == 2 and atom_type("TRP", "NE1") == 2
    assert atom_type("ARG", "NH1") == 3 and atom_type("LYS", "NZ") == 4
    assert atom_type("GLY", "O") == 5 and atom_type("GLN", "OE1") == 5
    assert atom_type("SER", "OG") == 6 and atom_type("TYR", "OH") == 6
    assert atom_type("ASP", "OD2") == 7 and atom_type("GLU", "OE1") == 7 and atom_type("LEU", "OXT") == 7
    assert atom_type("ALA", "C") == 8 and atom_type("ARG", "CZ") == 8
    assert atom_type("PHE", "CZ") == 9 and atom_type("TRP", "CH2") == 9
    assert atom_type("ALA", "CA") == 10 and atom_type("LEU", "CD1") == 10
    assert atom_type("GLY", "H") == -1 and atom_type("ALA", "1HB") == -1


def test_backend_parsing_typing_and_transforms(tmp_path):
    f = str(tmp_path / "a.pdb")
    nheavy = write_fake_pdb(f, 12, seed=1)
    xyz, chains, resn, resi, atn = read_pdb_atoms(f)
    assert len(xyz) == nheavy + 1                                  # the H is read, HETATM is not
    be = CoordsBackend()
    coords, ch, rn, ri, an, nat = be.pdb2coords([f, f])
    typed, counts, offs = be.assign_types(coords, rn, an, nat)
    assert counts.shape == (2, NUM_ATOM_TYPES) and int(counts[0].sum()) == nheavy
    assert be.last_num_typed.tolist() == [nheavy, nheavy]
    assert offs[0].tolist() == (np.cumsum(counts[0].numpy()) - counts[0].numpy()).tolist()
    # atoms of type t sit in slots [offs[t], offs[t]+counts[t]) and keep file order
    ty = np.array([atom_type(rn[0][i], an[0][i]) for i in range(len(an[0]))])
    for t in range(NUM_ATOM_TYPES):
        want = xyz[ty == t]
        got = typed[0, 3 * int(offs[0, t]):3 * int(offs[0, t] + counts[0, t])].reshape(-1, 3).numpy()
        np.testing.assert_allclose(got, want)
    a, b = be.get_bbox(typed, be.last_num_typed)
    np.testing.assert_allclose(a[0].numpy(), xyz[ty >= 0].min(0))
    T = -(a + b) * 0.5
    centred = be.translate(typed, T, be.last_num_typed)
    a2, b2 = be.get_bbox(centred, be.last_num_typed)
    assert (a2 + b2).abs().max() < 1e-9
    R = torch.from_numpy(orc.euler_to_matrix([0.4], [1.0], [-0.7]))
    rot = be.rotate(centred, R, be.last_num_typed)
    n = nheavy
    np.testing.assert_allclose(rot[0, :3 * n].reshape(n, 3).numpy(),
                               centred[0, :3 * n].reshape(n, 3).numpy() @ R[0].numpy().T, atol=1e-12)
    assert rot[0, 3 * n:].abs().max() == 0 if rot.shape[1] > 3 * n else True


def _typed(tmp_path, nres, seed):
    f = str(tmp_path / ("p%d.pdb" % seed))
    write_fake_pdb(f, nres, seed)
    be = CoordsBackend()
    coords, ch, rn, ri, an, nat = be.pdb2coords([f])
    typed, counts, offs = be.assign_types(coords, rn, an, nat)
    a, b = be.get_bbox(typed, be.last_num_typed)
    return f, be.translate(typed, -(a + b) * 0.5, be.last_num_typed), counts, offs


def test_projection_kernel_matches_oracle(emu, tmp_path):
    L, res = 24, 1.25
    _, coords, counts, offs = _typed(tmp_path, 10, seed=3)
    be = CoordsBackend(lib=emu)
    centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)
    R = torch.from_numpy(orc.euler_to_matrix([0.4, -2.0], [1.0, 0.3], [-0.7, 1.9])).float()
    vol = be.project(coords, counts, offs, L, res, "cpu", R=R, shift=centre)
    assert vol.shape == (2, NUM_ATOM_TYPES, L, L, L)
    for b in range(2):
        want = orc.project_atoms(coords[0].numpy(), counts[0].numpy(), offs[0].numpy(), L, res,
                                 R=R[b].double().numpy(), shift=centre[0].numpy())
        assert np.abs(vol[b].numpy() - want).max() < 1e-4
        assert want.sum() > 10
    s = be.project(coords, counts, offs, L, res, "cpu", R=R, shift=centre, sum_types=True)
    assert (s[:, 0] - vol.sum(dim=1)).abs().max() < 1e-4
    plain = be.project(be.translate(coords, centre, torch.tensor([int(counts.sum())])), counts, offs, L, res, "cpu")
    want = orc.project_atoms(coords[0].numpy(), counts[0].numpy(), offs[0].numpy(), L, res, shift=centre[0].numpy())
    assert np.abs(plain[0].numpy() - want).max() < 1e-4


def _protein_typed(tmp_path, nres, seed, nchains=2):
    from synth_pdb import write_protein_like_pdb
    f = str(tmp_path / ("prot%d.pdb" % seed))
    kept = write_protein_like_pdb(f, nres, seed, nchains=nchains)
    be = CoordsBackend()
    coords, ch, rn, ri, an, nat = be.pdb2coords([f])
    typed, counts, offs = be.assign_types(coords, rn, an, nat)
    assert int(counts.sum()) == kept and len(set(ch[0])) == nchains      # H / HETATM / altloc B / MODEL 2 dropped
    a, b = be.get_bbox(typed, be.last_num_typed)
    return f, be.translate(typed, -(a + b) * 0.5, be.last_num_typed), counts, offs


def test_splat_accumulator_holds_hundreds_of_stacked_atoms(emu):
    """Pathological input (duplicated records / several models summed into one channel): 300 atoms on one site.  The
    fixed-point accumulator (2^-20 units in 32 unsigned bits) must hold the sum -- a 2^-24 signed one wrapped to a
    negative density at 128, which inverts the clash mask."""
    from deeplocalproteindocking_amd.Utils.FullAtom import NUM_ATOM_TYPES
    n, L, res = 300, 8, 1.0
    coords = torch.full((1, 3 * n), 3.0, dtype=torch.float32)        # all on the voxel centre (3, 3, 3)
    counts = torch.zeros(1, NUM_ATOM_TYPES, dtype=torch.int32)
    offs = torch.zeros(1, NUM_ATOM_TYPES, dtype=torch.int32)
    counts[0, 2] = n
    be = CoordsBackend(lib=emu)
    vol = be.project(coords, counts, offs, L, res, "cpu", sum_types=True)
    assert abs(float(vol[0, 0, 3, 3, 3]) - n) < 1e-3
    assert float(vol.min()) >= 0.0
    assert abs(float(vol[0, 0, 3, 3, 4]) - n * np.exp(-0.5)) < 1e-2


def test_reader_on_a_protein_sized_file(tmp_path):
    """Thousands of atoms, two chains, negative residue numbers, insertion codes, ANISOU, altloc,
    HETATM, a second MODEL: only first-model heavy ATOM records with altloc ' '/'A' are typed."""
    f, coords, counts, offs = _protein_typed(tmp_path, 260, seed=11)
    assert int(counts.sum()) > 2000 and (counts > 0).all()
    xyz, chains, resn, resi, atn = read_pdb_atoms(f)
    assert min(resi) < 0 and "HOH" not in resn and "ZN" not in [r.strip() for r in resn]
    assert np.abs(xyz).max() < 45.0                                  # MODEL 2 (at +50 A) not read


@pytest.mark.gpu
def test_projection_kernel_matches_oracle_on_gpu_protein_sized(tmp_path):
    """dlpd_project_atoms on the device, directly against the oracle (not through ranked lists): a
    2,200-atom two-chain globule at the reference's box (80 / 1.25 A), rotated on the fly, per type and
    summed; plus bit-identical repeat runs (fixed-point accumulation)."""
    import __graft_entry__ as entry
    entry.build()
    dev = torch.device("cuda:0")
    L, res = 80, 1.25
    _, coords, counts, offs = _protein_typed(tmp_path, 260, seed=12)
    be = CoordsBackend()
    centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)
    R = torch.from_numpy(orc.euler_to_matrix([0.0, 0.4, -2.0], [0.0, 1.0, 0.3], [0.0, -0.7, 1.9])).float()
    vol = be.project(coords, counts, offs, L, res, dev, R=R.to(dev), shift=centre)
    assert vol.shape == (3, NUM_ATOM_TYPES, L, L, L)
    n = int(counts.sum())
    xyz = coords[0, :3 * n].reshape(n, 3).numpy()
    for b in range(3):
        want = orc.project_atoms_fast(coords[0].numpy(), counts[0].numpy(), offs[0].numpy(), L, res,
                                      R=R[b].double().numpy(), shift=centre[0].numpy())
        diff = np.abs(vol[b].cpu().numpy() - want).max(axis=0)       # (L,L,L), worst over types
        # The 5^3 window makes the density a discontinuous function of the position: an atom whose p'/res sits
        # within float32 round-off of an integer may be binned one cell over by the f32 kernel (up to
        # exp(-(2*1.25)^2/2) = 0.044 on the window's faces).  Such atoms are identified from the f64 positions;
        # everywhere else the kernel must agree to f32 expf + 2^-20 fixed-point accuracy.
        p = (xyz @ R[b].double().numpy().T + centre[0].numpy()) / res
        near = np.abs(p - np.round(p)).min(axis=1) < 1e-4
        allowed = np.zeros((L, L, L), dtype=bool)
        for c in np.round(p[near]).astype(int):
            lo, hi = np.maximum(c - 4, 0), np.minimum(c + 5, L)
            allowed[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
        assert near.sum() <= 5, near.sum()
        err = diff[~allowed].max()
        assert err < 2e-5 * max(1.0, want.max()), (err, int(near.sum()))
        assert diff.max() < 0.05
        assert want.sum() > 1000 and want.max() < 50.0                   # (the 32-bit fixed-point accumulator holds 4095)
    s = be.project(coords, counts, offs, L, res, dev, R=R.to(dev), shift=centre, sum_types=True)
    assert (s[:, 0] - vol.sum(dim=1)).abs().max() < 1e-5
    for _ in range(3):                                               # run-to-run: bit-identical
        again = be.project(coords, counts, offs, L, res, dev, R=R.to(dev), shift=centre)
        assert torch.equal(again, vol)


class _TinyRepr(torch.nn.Module):
    """single-resolution stand-in for a representation plugin: fixed random 11 -> C 3x3x3 conv"""

    def __init__(self, C=4):
        super().__init__()
        torch.manual_seed(77)
        self.conv = torch.nn.Conv3d(NUM_ATOM_TYPES, C, 3, padding=1, bias=False)
        self.C = C

    def get_num_outputs(self):
        return [self.C]

    def forward(self, volume):
        return [self.conv(volume) * 0.05]


class _Model(torch.nn.Module):
    def __init__(self, repr_, filt, thr):
        super().__init__()
        self.representation, self.filter, self.threshold_clash, self.clip = repr_, filt, thr, 5.0


def _tiny_model(thr=3.0):
    """_TinyRepr(4) + a SimpleFilter whose seed gives NEGATIVE scores on these synthetic pairs: with an all-positive
    filter output every pick of update_top is a masked 0.0 and any list of zeros would compare equal."""
    from deeplocalproteindocking_amd.Models import SimpleFilter
    repr_ = _TinyRepr(4)
    torch.manual_seed(80)
    return _Model(repr_, SimpleFilter([4]), thr=thr)


def _assert_scores_are_informative(top_list):
    scores = [t[4] for t in top_list]
    assert min(scores) < -0.05 and max(scores) < 0.0 and len(set(scores)) > len(scores) // 4


def _dock_reference_shape(be, model, frec, flig, R, L, res, K, randR=None):
    """The reference loop of Docker.dockSE3 restated with the oracle pieces (randR: the random rotation the
    reference applies to the receptor's atoms, Docker.py:193-194)."""
    centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)

    def load(f):
        c, ch, rn, ri, an, nat = be.pdb2coords([f])
        t, cnt, off = be.assign_types(c, rn, an, nat)
        a, b = be.get_bbox(t, be.last_num_typed)
        return be.translate(t, -(a + b) * 0.5, be.last_num_typed), cnt, off
    rc, rn_, ro = load(frec)
    lc, ln_, lo = load(flig)
    rec = torch.from_numpy(orc.project_atoms(rc[0].numpy(), rn_[0].numpy(), ro[0].numpy(), L, res,
                                             R=None if randR is None else np.asarray(randR, dtype=np.float64),
                                             shift=centre[0].numpy())).float()[None]
    lig = torch.from_numpy(orc.project_atoms(lc[0].numpy(), ln_[0].numpy(), lo[0].numpy(), L, res,
                                             shift=centre[0].numpy())).float()[None]
    with torch.no_grad():
        rv, lv = model.representation(rec), model.representation(lig)
    W = [w.cpu() for w in model.filter.parameters_tuple()]
    top, scale = [], 0.0
    for ri in range(R.shape[0]):
        Rb = torch.from_numpy(R[ri:ri + 1]).float()
        lrot = [orc.rotate_volume(v, Rb) for v in lv]
        lforb = torch.from_numpy(orc.project_atoms(lc[0].numpy(), ln_[0].numpy(), lo[0].numpy(), L, res, R=R[ri],
                                                   shift=centre[0].numpy(), sum_types=True)).float()[None]
        mask, _ = orc.clash_mask(rec.sum(dim=1, keepdim=True), lforb, model.threshold_clash)
        V = (mask * orc.score_volumes(rv, lrot, *W, clip=5.0))[0].contiguous()
        scale = max(scale, float(V.abs().max()))
        idx, sc = orc.rotation_picks_fast(V.numpy(), K)
        x, y, z = orc.flat_to_xyz(idx, 2 * L)
        top += [(ri, int(x[i]), int(y[i]), int(z[i]), float(sc[i])) for i in range(K)]
        top.sort(key=lambda t: t[4])
        top = top[:K]
    return top, scale


def _dock_reference_shape_e3(be, model, frec, flig, R, L, res, K):
    """The reference loop of Docker.dockE3 (Docker.py:135-182) restated with the oracle pieces: rotate the
    ATOMS, project, represent, correlate -- no volume rotation."""
    centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)

    def load(f):
        c, ch, rn, ri, an, nat = be.pdb2coords([f])
        t, cnt, off = be.assign_types(c, rn, an, nat)
        a, b = be.get_bbox(t, be.last_num_typed)
        return be.translate(t, -(a + b) * 0.5, be.last_num_typed), cnt, off
    rc, rn_, ro = load(frec)
    lc, ln_, lo = load(flig)
    rec = torch.from_numpy(orc.project_atoms(rc[0].numpy(), rn_[0].numpy(), ro[0].numpy(), L, res,
                                             shift=centre[0].numpy())).float()[None]
    W = [w.cpu() for w in model.filter.parameters_tuple()]
    top, scale = [], 0.0
    with torch.no_grad():
        rv = model.representation(rec)
        for ri in range(R.shape[0]):
            lig = torch.from_numpy(orc.project_atoms(lc[0].numpy(), ln_[0].numpy(), lo[0].numpy(), L, res, R=R[ri],
                                                     shift=centre[0].numpy())).float()[None]
            lv = model.representation(lig)
            mask, _ = orc.clash_mask(rec.sum(dim=1, keepdim=True), lig.sum(dim=1, keepdim=True), model.threshold_clash)
            V = (mask * orc.score_volumes(rv, lv, *W, clip=5.0))[0].contiguous()
            scale = max(scale, float(V.abs().max()))
            idx, sc = orc.rotation_picks_fast(V.numpy(), K)
            x, y, z = orc.flat_to_xyz(idx, 2 * L)
            top += [(ri, int(x[i]), int(y[i]), int(z[i]), float(sc[i])) for i in range(K)]
            top.sort(key=lambda t: t[4])
            top = top[:K]
    return top, scale


def test_dockE3_end_to_end_emulated(emu, tmp_path):
    """PDB files -> Docker.dockE3 (per-batch re-projection + representation, engine fed with the batch's
    volumes) -> list, all kernels emulated, vs the oracle restatement of Docker.py:135-182."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import SimpleFilter
    L, res, K = 32, 1.25, 20
    frec, _, _, _ = _typed(tmp_path, 14, seed=5)
    flig, _, _, _ = _typed(tmp_path, 9, seed=6)
    model = _tiny_model()
    R = orc.euler_to_matrix([0.3, -1.0, 2.0], [1.1, 0.4, 2.2], [-2.0, 2.5, 0.1])
    be = CoordsBackend(lib=emu)
    dk = Docker(model, box_size=L, resolution=res, max_conf=K, rotations=R, device="cpu", coords_backend=be, lib=emu)
    with torch.no_grad():
        dk.dockE3(frec, flig, batch_size=2)
    want, scale = _dock_reference_shape_e3(be, model, frec, flig, R, L, res, K)
    assert len(dk.top_list) == K
    _assert_scores_are_informative(want)
    assert max(abs(a[4] - b[4]) for a, b in zip(dk.top_list, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(dk.top_list, want)) >= K - 2


def test_dockSE3_end_to_end_emulated(emu, tmp_path):
    """PDB files -> Docker.dockSE3 -> .dat, all kernels emulated, vs the oracle restatement."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import SimpleFilter
    L, res, K = 32, 1.25, 20
    frec, _, _, _ = _typed(tmp_path, 14, seed=5)
    flig, _, _, _ = _typed(tmp_path, 9, seed=6)
    model = _tiny_model()
    R = orc.euler_to_matrix([0.3, -1.0, 2.0], [1.1, 0.4, 2.2], [-2.0, 2.5, 0.1])
    be = CoordsBackend(lib=emu)
    # no coords_backend argument, as in the reference's constructor: Docker creates its own
    dk = Docker(model, box_size=L, resolution=res, max_conf=K, rotations=R, device="cpu", lib=emu)
    log = str(tmp_path / "out.dat")
    assert dk.new_log(log)
    with torch.no_grad():
        dk.dockSE3(frec, flig, batch_size=2)
    dk.cleanup()
    want, scale = _dock_reference_shape(be, model, frec, flig, R, L, res, K)
    assert len(dk.top_list) == K
    _assert_scores_are_informative(want)
    assert max(abs(a[4] - b[4]) for a, b in zip(dk.top_list, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(dk.top_list, want)) >= K - 2
    lines = open(log).read().strip().split("\n")
    assert len(lines) == K and all(len(l.split("\t")) == 13 for l in lines)
    assert dk.new_log(log, rewrite=False) is False              # finished target is skipped on resume
    # the next target reuses the engine (workspaces, top-list buffers): nothing of the first pair may survive
    eng = dk.engine
    with torch.no_grad():
        dk.dockSE3(flig, frec, batch_size=2)
    assert dk.engine is eng
    fresh = Docker(model, box_size=L, resolution=res, max_conf=K, rotations=R, device="cpu", lib=emu)
    with torch.no_grad():
        fresh.dockSE3(flig, frec, batch_size=2)
    assert dk.top_list == fresh.top_list and fresh.engine is not eng


class _GatedFilter(torch.nn.Module):
    """A user-defined filter (NOT the reference's Linear-ReLU-Linear): any module mapping
    (voxels, channels) -> (voxels, 1) must work as ``docking_model.filter`` (DockingModels.py:82)."""

    def __init__(self, C):
        super().__init__()
        self.a, self.b = torch.nn.Linear(C, 3), torch.nn.Linear(C, 3)

    def forward(self, x):
        return (torch.tanh(self.a(x)) * self.b(x)).sum(dim=-1, keepdim=True)


def _oracle_list_with_filter(rv, lv, recf, ligf, R, filt, thr, K, clip=5.0):
    """Reference loop (Docker.py:211-236) with GlobalDockingModel.forward's data movement
    (DockingModels.py:70-83) and an arbitrary filter module."""
    top, scale = [], 0.0
    L = rv[0].shape[-1]
    N = 2 * L
    for ri in range(R.shape[0]):
        Rb = torch.from_numpy(R[ri:ri + 1]).float()
        conv = []
        for r, l in zip(rv, lv):
            c = orc.correlate_fft(r, orc.rotate_volume(l, Rb), clip=clip)
            if c.shape[2] < N:
                s = N // c.shape[2]
                c = c.repeat_interleave(s, 2).repeat_interleave(s, 3).repeat_interleave(s, 4)
            conv.append(c)
        feat = torch.cat(conv, dim=1).permute(0, 2, 3, 4, 1).reshape(N * N * N, -1)
        with torch.no_grad():
            V = filt(feat).reshape(1, N, N, N)
        mask, _ = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
        V = (mask * V)[0].contiguous()
        scale = max(scale, float(V.abs().max()))
        idx, sc = orc.rotation_picks_fast(V.numpy(), K)
        x, y, z = orc.flat_to_xyz(idx, N)
        top += [(ri, int(x[i]), int(y[i]), int(z[i]), float(sc[i])) for i in range(K)]
        top.sort(key=lambda t: t[4])
        top = top[:K]
    return top, scale


def _volume_case(seed, C=4, L=32):
    g = torch.Generator().manual_seed(seed)
    rec, lig = torch.randn(1, C, L, L, L, generator=g) * 0.1, torch.randn(1, C, L, L, L, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    R = orc.euler_to_matrix([0.3, -1.0, 2.0], [1.1, 0.4, 2.2], [-2.0, 2.5, 0.1])
    return rec, lig, recf, ligf, R


def _check_lists(got, want, scale, K):
    assert len(got) == len(want) == K
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(got, want)) >= K - 2


def _check_lists_band(got, want, scale, K, tol=1e-4):
    """Same ranked list up to the rounding band: scores rank by rank within tol * scale, every pose of the list is a pose
    of the oracle's list unless it sits within the band of the K-th score, and a pose may stand at another rank only if
    the oracle's scores at the two ranks are closer than the band."""
    assert len(got) == len(want) == K
    band = tol * scale
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= band
    where = {tuple(b[:4]): i for i, b in enumerate(want)}
    moved = 0
    for i, a in enumerate(got):
        j = where.get(tuple(a[:4]))
        if j is None:
            assert abs(a[4] - want[-1][4]) <= band
        else:
            assert abs(want[i][4] - want[j][4]) <= band and abs(a[4] - want[j][4]) <= band
            moved += int(i != j)
    return moved


def _run_user_filter(lib, device):
    """Docker.py:229: ``V = self.docking_model(rec, lig_rot)`` really calls the user's module."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SyntheticRepr
    L, C, K, thr = 32, 4, 25, 4000.0
    rec, lig, recf, ligf, R = _volume_case(31, C, L)
    torch.manual_seed(5)
    filt = _GatedFilter(C)
    model = GlobalDockingModel(SyntheticRepr((C,)), filt, threshold_clash=thr, lib=lib)
    want, scale = _oracle_list_with_filter([rec], [lig], recf, ligf, R, filt, thr, K)
    dk = Docker(model.to(device), box_size=L, max_conf=K, rotations=R, device=device, lib=lib)
    got = dk.dock_volumes([rec], [lig], recf, ligf, write=False, model_batch=2)
    assert dk.path == "call"
    _check_lists(got, want, scale, K)

    class OwnForward(torch.nn.Module):                     # a docking model with a forward of its own
        def __init__(self, inner):
            super().__init__()
            self.inner, self.representation, self.filter = inner, inner.representation, inner.filter
            self.threshold_clash, self.calls = inner.threshold_clash, 0

        def forward(self, receptor_volumes, ligand_volumes):
            self.calls += 1
            return 2.0 * self.inner(receptor_volumes, ligand_volumes)
    own = OwnForward(model)
    dk2 = Docker(own, box_size=L, max_conf=K, rotations=R, device=device, lib=lib)
    got2 = dk2.dock_volumes([rec], [lig], recf, ligf, write=False, model_batch=2)
    assert dk2.path == "call" and own.calls == 2            # 3 rotations, 2 per call
    assert [t[:4] for t in got2] == [t[:4] for t in got]
    assert max(abs(a[4] - 2.0 * b[4]) for a, b in zip(got2, got)) <= 1e-5 * scale


def test_user_defined_filter_and_model_are_called_emulated(emu):
    _run_user_filter(emu, "cpu")


@pytest.mark.gpu
def test_user_defined_filter_and_model_are_called_on_gpu():
    import __graft_entry__ as entry
    entry.build()
    _run_user_filter(None, torch.device("cuda:0"))


def _run_wide_hidden(lib, device):
    """Hidden width 40 > 32 at box 32, where the fused pipeline has no wide-hidden kernel (it has at box 64 / 80:
    dlpd_fused_hidden_pad): the search must take the stand-alone ops + HIP filter kernel instead of raising."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
    L, C, K, thr = 32, 4, 25, 4000.0
    rec, lig, recf, ligf, R = _volume_case(32, C, L)
    torch.manual_seed(6)
    filt = SimpleFilter([C])
    filt.fc[0], filt.fc[2] = torch.nn.Linear(C, 40), torch.nn.Linear(40, 1)
    model = GlobalDockingModel(SyntheticRepr((C,)), filt, threshold_clash=thr, lib=lib)
    W = [w.cpu() for w in filt.parameters_tuple()]
    want = orc.dock_volumes([rec], [lig], recf[None, None], ligf[None, None], R, *W, thr, K, clip=5.0, return_V=True)
    scale = max(float(v.abs().max()) for v in want[1])
    dk = Docker(model.to(device), box_size=L, max_conf=K, rotations=R, device=device, lib=lib)
    got = dk.dock_volumes([rec], [lig], recf, ligf, write=False)
    assert dk.path == "ops"
    _check_lists(got, want[0], scale, K)


def _oracle_list_with_pivot(rec, lig, recf, ligf, R, W, thr, K, pivot):
    top, scale, N = [], 0.0, 2 * rec.shape[-1]
    for ri in range(R.shape[0]):
        Rb = torch.from_numpy(R[ri:ri + 1]).float()
        lrot = [orc.rotate_volume(lig, Rb, center=pivot)]
        lf = orc.rotate_volume(ligf[None, None], Rb, center=pivot)
        mask, _ = orc.clash_mask(recf[None, None], lf, thr)
        V = (mask * orc.score_volumes([rec], lrot, *W, clip=5.0))[0].contiguous()
        scale = max(scale, float(V.abs().max()))
        idx, sc = orc.rotation_picks_fast(V.numpy(), K)
        x, y, z = orc.flat_to_xyz(idx, N)
        top += [(ri, int(x[i]), int(y[i]), int(z[i]), float(sc[i])) for i in range(K)]
        top.sort(key=lambda t: t[4])
        top = top[:K]
    return top, scale


def _run_rotation_pivot(lib, device, L):
    """``Docker(rotation_center=...)`` moves the pivot of the trilinear volume rotation (the one TorchProteinLibrary
    convention most likely to differ from this build's, unpinnable here): "grid_sample" = index (L-1)/2.  The fused
    engine (compiled box) and the stand-alone ops (any other box) must both follow it -- oracle with the same pivot --
    and the default (L/2) must give a different list."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
    C, K = 4, 20
    thr = 0.12 * L ** 3
    rec, lig, recf, ligf, R = _volume_case(35, C, L)
    torch.manual_seed(6)
    filt = SimpleFilter([C])
    model = GlobalDockingModel(SyntheticRepr((C,)), filt, threshold_clash=thr, lib=lib).to(device)
    W = [w.cpu() for w in filt.parameters_tuple()]
    want, scale = _oracle_list_with_pivot(rec, lig, recf, ligf, R, W, thr, K, (L - 1) / 2.0)
    dk = Docker(model, box_size=L, max_conf=K, rotations=R, device=device, lib=lib, rotation_center="grid_sample")
    got = dk.dock_volumes([rec], [lig], recf, ligf, write=False)
    _check_lists(got, want, scale, K)
    default = Docker(model, box_size=L, max_conf=K, rotations=R, device=device, lib=lib).dock_volumes([rec], [lig], recf, ligf,
                                                                                               write=False)
    assert [t[:4] for t in default] != [t[:4] for t in got]
    return dk.path


def test_rotation_pivot_is_switchable_from_docker_emulated(emu):
    assert _run_rotation_pivot(emu, "cpu", 32) == "fused"
    assert _run_rotation_pivot(emu, "cpu", 12) == "embedded"      # (rotated at its own size about its own pivot)


def _run_uncompiled_box(lib, device, L, K, C=4):
    """A box size without a compiled FFT plan (the reference's box_size is a free argument, Docker.py:18,22-24,31):
    Docker must run the search through the stand-alone ops -- generic plan-free correlation, HIP filter kernel, device
    top-K -- instead of raising; ranked list against the oracle."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
    thr = 0.12 * L ** 3
    rec, lig, recf, ligf, R = _volume_case(33, C, L)
    torch.manual_seed(6)
    filt = SimpleFilter([C])
    model = GlobalDockingModel(SyntheticRepr((C,)), filt, threshold_clash=thr, lib=lib)
    W = [w.cpu() for w in filt.parameters_tuple()]
    want = orc.dock_volumes([rec], [lig], recf[None, None], ligf[None, None], R, *W, thr, K, clip=5.0, return_V=True)
    scale = max(float(v.abs().max()) for v in want[1])
    dk = Docker(model.to(device), box_size=L, max_conf=K, rotations=R, device=device, lib=lib)
    dk.embed_uncompiled_boxes = False
    got = dk.dock_volumes([rec], [lig], recf, ligf, write=False)
    assert dk.path == "ops" and dk.engine is None
    _check_lists(got, want[0], scale, K)
    # the default for such a box: the fused kernels on the next compiled box, the reference's grid gathered out of it
    dk2 = Docker(model.to(device), box_size=L, max_conf=K, rotations=R, device=device, lib=lib)
    got2 = dk2.dock_volumes([rec], [lig], recf, ligf, write=False)
    assert dk2.path == "embedded" and dk2.engine is not None and dk2.engine.L == dk2.engine_box > L
    assert dk2.engine.extent == L and dk2.engine.use_cl == (C >= 8)
    _check_lists(got2, want[0], scale, K)


def test_uncompiled_box_size_takes_the_generic_path_emulated(emu):
    assert not emu.call("dlpd_grid_supported", 12) and emu.call("dlpd_generic_box_supported", 12)
    assert not emu.call("dlpd_generic_box_supported", 129)
    _run_uncompiled_box(emu, "cpu", 12, 20)
    _run_uncompiled_box(emu, "cpu", 20, 20, C=8)          # eight channels: the channels-last K1 with its crop


@pytest.mark.gpu
def test_uncompiled_box_sizes_take_the_generic_path_on_gpu():
    """box_size 48 (grid 96^3) and 50 (grid 100^3: not a multiple of 8, radix 5) on the device."""
    import __graft_entry__ as entry
    entry.build()
    _run_uncompiled_box(None, torch.device("cuda:0"), 48, 200)
    _run_uncompiled_box(None, torch.device("cuda:0"), 50, 200)


def _run_embedded_two_resolutions(lib, device, L, C0, C1, K):
    """The reference's two-resolution layout [C0 @ L^3, C1 @ (L/2)^3] at a box size without a compiled plan: fused kernels
    on the next compiled pair (L -> 80 / 40 or 64 / 32), ranked list against the oracle at the box's own size."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
    thr = 0.12 * L ** 3
    rec, lig, recf, ligf, R = _volume_case(41, C0, L)
    g = torch.Generator().manual_seed(42)
    rec1, lig1 = torch.randn(1, C1, L // 2, L // 2, L // 2, generator=g) * 0.1, torch.randn(1, C1, L // 2, L // 2, L // 2, generator=g) * 0.1
    R = R[:2]
    torch.manual_seed(8)
    filt = SimpleFilter([C0, C1])
    model = GlobalDockingModel(SyntheticRepr((C0, C1)), filt, threshold_clash=thr, lib=lib)
    W = [w.cpu() for w in filt.parameters_tuple()]
    want = orc.dock_volumes([rec, rec1], [lig, lig1], recf[None, None], ligf[None, None], R, *W, thr, K, clip=5.0, return_V=True)
    scale = max(float(v.abs().max()) for v in want[1])
    dk = Docker(model.to(device), box_size=L, max_conf=K, rotations=R, device=device, lib=lib)
    got = dk.dock_volumes([rec, rec1], [lig, lig1], recf, ligf, write=False)
    assert dk.path == "embedded" and dk.engine.C1 == C1 and dk.engine.L > L
    # (a transform of another length rounds differently: poses whose oracle scores are ~1e-6 apart may swap)
    assert _check_lists_band(got, want[0], scale, K) <= K // 20
    return dk.engine_box


@pytest.mark.gpu
def test_uncompiled_two_resolution_boxes_run_on_the_fused_kernels():
    """box 72 -> [4 @ 72^3, 6 @ 36^3] inside the 80 / 40 plans; box 56 -> inside 64 / 32."""
    import __graft_entry__ as entry
    entry.build()
    assert _run_embedded_two_resolutions(None, torch.device("cuda:0"), 72, 4, 6, 200) == 80
    assert _run_embedded_two_resolutions(None, torch.device("cuda:0"), 56, 4, 6, 200) == 64


def test_hidden_width_above_32_takes_the_ops_path_emulated(emu):
    _run_wide_hidden(emu, "cpu")


@pytest.mark.gpu
def test_hidden_width_above_32_takes_the_ops_path_on_gpu():
    import __graft_entry__ as entry
    entry.build()
    _run_wide_hidden(None, torch.device("cuda:0"))


@pytest.mark.gpu
def test_dockSE3_and_dockE3_on_gpu(tmp_path):
    """Same on the real device, plus dockE3 (re-projection + representation per batch) with the
    reference-shaped two-resolution CNN plugin; both entry points must agree on a single-resolution
    model up to the trilinear-vs-reprojection difference being absent (E3 == SE3 only for R = I)."""
    import __graft_entry__ as entry
    entry.build()
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4, GlobalDockingModel, SimpleFilter
    dev = torch.device("cuda:0")
    L, res, K = 32, 1.25, 30
    frec, _, _, _ = _typed(tmp_path, 14, seed=5)
    flig, _, _, _ = _typed(tmp_path, 9, seed=6)
    model = _tiny_model().to(dev)
    R = orc.euler_to_matrix([0.3, -1.0, 2.0], [1.1, 0.4, 2.2], [-2.0, 2.5, 0.1])
    be = CoordsBackend()
    dk = Docker(model, box_size=L, resolution=res, max_conf=K, rotations=R, device=dev, coords_backend=be)
    with torch.no_grad():
        dk.dockSE3(frec, flig, batch_size=2)
    cpu_model = _Model(_TinyRepr(4), model.filter.cpu(), thr=3.0)
    want, scale = _dock_reference_shape(be, cpu_model, frec, flig, R, L, res, K)
    assert max(abs(a[4] - b[4]) for a, b in zip(dk.top_list, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(dk.top_list, want)) >= K - 2
    # two-resolution reference-shaped plugin through dockE3 and dockSE3 with the identity rotation:
    # both must give the same list (no rotation -> re-projection == the unrotated volumes)
    torch.manual_seed(79)
    repr2 = E3MultiResRepr4x4(multiplier=8)          # 16 / 32 channels: every layer on the HIP kernels
    gm = GlobalDockingModel(repr2, SimpleFilter(repr2.get_num_outputs()), threshold_clash=3.0).to(dev)
    I = np.eye(3)[None]
    L2 = 64                                                   # second resolution 32^3 (16^3 is not compiled)
    d1 = Docker(gm, box_size=L2, resolution=res, max_conf=K, rotations=I, device=dev, coords_backend=be)
    d2 = Docker(gm, box_size=L2, resolution=res, max_conf=K, rotations=I, device=dev, coords_backend=be)
    with torch.no_grad():
        d1.dockSE3(frec, flig, batch_size=1)
        d2.dockE3(frec, flig, batch_size=1)
    s = max(abs(t[4]) for t in d1.top_list) + 1e-6
    assert max(abs(a[4] - b[4]) for a, b in zip(d1.top_list, d2.top_list)) <= 1e-4 * s
    assert sum(a[:4] == b[:4] for a, b in zip(d1.top_list, d2.top_list)) >= K - 2
    # ... and both equal the oracle restatement of the loop (not only each other)
    import copy
    want2, scale2 = _dock_reference_shape(be, copy.deepcopy(gm).cpu(), frec, flig, I, L2, res, K)
    for got2 in (d1.top_list, d2.top_list):
        assert max(abs(a[4] - b[4]) for a, b in zip(got2, want2)) <= 1e-4 * scale2
        assert sum(a[:4] == b[:4] for a, b in zip(got2, want2)) >= K - 2


@pytest.mark.gpu
def test_dockSE3_reference_configuration_on_gpu(tmp_path):
    """BASELINE config 4 geometry end to end: PDB files -> 11-type densities at box 80 / 1.25 A ->
    SE3MultiResReprScalar(multiplier=8) = [16 @ 80^3, 32 @ 40^3] -> 160^3 search with per-rotation
    clash re-projection (local_test.py:53-69), against the oracle restatement of the same loop."""
    import __graft_entry__ as entry
    entry.build()
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
    dev = torch.device("cuda:0")
    L, res, K = 80, 1.25, 40
    frec, _, _, _ = _typed(tmp_path, 40, seed=15)
    flig, _, _, _ = _typed(tmp_path, 25, seed=16)
    torch.manual_seed(80)
    repr_ = SE3MultiResReprScalar(multiplier=8)
    model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=3.0)
    R = orc.euler_to_matrix([0.3, -1.0], [1.1, 0.4], [-2.0, 2.5])
    be = CoordsBackend()
    want, scale = _dock_reference_shape(be, model, frec, flig, R, L, res, K)
    dk = Docker(model.to(dev), box_size=L, resolution=res, max_conf=K, rotations=R, device=dev, coords_backend=be)
    assert dk.new_log(str(tmp_path / "pair.dat"))
    with torch.no_grad():
        dk.dockSE3(frec, flig, batch_size=2)
    assert max(abs(a[4] - b[4]) for a, b in zip(dk.top_list, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(dk.top_list, want)) >= K - 2
    dk.cleanup()
    assert len(open(tmp_path / "pair.dat").read().strip().splitlines()) == K


@pytest.mark.gpu
def test_dockSE3_at_a_box_without_a_compiled_plan(tmp_path):
    """The same call sequence with ``box_size=72`` (a free argument of the reference, Docker.py:18): PDB files ->
    densities at 72^3 -> [16 @ 72^3, 32 @ 36^3], clash volume re-projected from the rotated atoms per batch, the search
    on the fused kernels inside the 80 / 40 plans; ranked list against the oracle at box 72."""
    import __graft_entry__ as entry
    entry.build()
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
    dev = torch.device("cuda:0")
    L, res, K = 72, 1.25, 40
    frec, _, _, _ = _typed(tmp_path, 40, seed=15)
    flig, _, _, _ = _typed(tmp_path, 25, seed=16)
    torch.manual_seed(80)
    repr_ = SE3MultiResReprScalar(multiplier=8)
    model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=3.0)
    R = orc.euler_to_matrix([0.3, -1.0], [1.1, 0.4], [-2.0, 2.5])
    be = CoordsBackend()
    want, scale = _dock_reference_shape(be, model, frec, flig, R, L, res, K)
    dk = Docker(model.to(dev), box_size=L, resolution=res, max_conf=K, rotations=R, device=dev, coords_backend=be)
    assert dk.new_log(str(tmp_path / "pair72.dat"))
    with torch.no_grad():
        dk.dockSE3(frec, flig, batch_size=2)
    assert dk.path == "embedded" and dk.engine_box == 80
    assert _check_lists_band(dk.top_list, want, scale, K) <= 2
    dk.cleanup()
    assert len(open(tmp_path / "pair72.dat").read().strip().splitlines()) == K


@pytest.mark.gpu
def test_dockE3_reference_configuration_on_gpu(tmp_path):
    """BASELINE config 5 geometry: E3MultiResRepr4x4(multiplier=8) = [16 @ 80^3, 32 @ 40^3], the ligand
    re-projected and re-represented for every rotation (local_test.py:67, Docker.py:135-182), scored by the
    fused engine from the batch's own volumes."""
    import __graft_entry__ as entry
    entry.build()
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4, GlobalDockingModel, SimpleFilter
    dev = torch.device("cuda:0")
    L, res, K = 80, 1.25, 40
    frec, _, _, _ = _typed(tmp_path, 40, seed=15)
    flig, _, _, _ = _typed(tmp_path, 25, seed=16)
    torch.manual_seed(81)
    repr_ = E3MultiResRepr4x4(multiplier=8)
    model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=3.0)
    R = orc.euler_to_matrix([0.3, -1.0, 0.7], [1.1, 0.4, 2.0], [-2.0, 2.5, 0.9])
    be = CoordsBackend()
    want, scale = _dock_reference_shape_e3(be, model, frec, flig, R, L, res, K)
    dk = Docker(model.to(dev), box_size=L, resolution=res, max_conf=K, rotations=R, device=dev, coords_backend=be)
    with torch.no_grad():
        dk.dockE3(frec, flig, batch_size=2)
    assert max(abs(a[4] - b[4]) for a, b in zip(dk.top_list, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(dk.top_list, want)) >= K - 2
    assert dk.path == "fused"
    # the plugin skipped the tiles that are zero away from the ligand (the default); computing everywhere gives the same list
    assert repr_.use_tile_occupancy
    skipping = list(dk.top_list)
    repr_.use_tile_occupancy = False
    try:
        with torch.no_grad():
            dk.dockE3(frec, flig, batch_size=2)
    finally:
        repr_.use_tile_occupancy = True
    assert dk.top_list == skipping
    # round 6: by default the skipped tiles are not WRITTEN either and the engine's K1 goes by the maps; with every
    # fresh buffer full of NaNs (a read of an unwritten cell would surface) the list is the same as with every voxel
    # written (unwritten_activations = False) -- entry for entry, bit for bit
    from guard_alloc import GuardedAllocations
    assert dk.unwritten_activations
    with GuardedAllocations(empty_byte=0xFF) as guard:
        with torch.no_grad():
            dk.dockE3(frec, flig, batch_size=2)
        assert guard.check() == []
    assert dk.top_list == skipping
    dk.unwritten_activations = False
    with torch.no_grad():
        dk.dockE3(frec, flig, batch_size=2)
    assert dk.top_list == skipping
    # the same at box 72 (no compiled plan): every batch's volumes inside the 80 / 40 engine
    L2 = 72
    want2, scale2 = _dock_reference_shape_e3(be, model.cpu(), frec, flig, R, L2, res, K)
    dk2 = Docker(model.to(dev), box_size=L2, resolution=res, max_conf=K, rotations=R, device=dev, coords_backend=be)
    with torch.no_grad():
        dk2.dockE3(frec, flig, batch_size=2)
    assert dk2.path == "embedded" and dk2.engine_box == 80
    assert _check_lists_band(dk2.top_list, want2, scale2, K) <= 2


@pytest.mark.gpu
def test_e3_plugin_on_the_hip_convolutions_reproduces_the_reference_class(golden):
    """Fixture G7 on hardware: the reference's E3MultiResRepr4x4 weights and input (multiplier 8: 16 / 32 channels, every
    layer on dlpd_conv3d / dlpd_maxpool3d_5s2) reproduce the reference's outputs to 1e-5 of the largest output value,
    in the default split-bf16 arithmetic and in exact f32."""
    import __graft_entry__ as entry
    entry.build()
    from test_host_logic import _g7_net
    from deeplocalproteindocking_amd import ops
    g = golden("g7_e3_plugin.npz")
    dev = torch.device("cuda:0")
    net = _g7_net(g, 8).to(dev)
    x = torch.from_numpy(g["m8_input"]).to(dev)
    saved = ops.CONV_PRECISION
    try:
        for precision in ("split_bf16", "f32"):
            ops.CONV_PRECISION = precision
            with torch.no_grad():
                v = net(x)
            for got, want in zip(v, (g["m8_out0"], g["m8_out1"])):
                err = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
                assert err <= 1e-5, (precision, err)
    finally:
        ops.CONV_PRECISION = saved


def test_a_prepared_pair_overwritten_in_its_engine_slot_refuses_to_dock(tmp_path):
    """Docker.prepare(slot=s) fills engine s with the pair's receptor spectrum and ligand; a second prepare into the SAME slot
    replaces that content, and docking the first pair afterwards would silently score the second pair's receptor -- it raises."""
    from emu_lib import emu_lib
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    lib = emu_lib()
    pa, pb, pc = _typed(tmp_path, 12, seed=61)[0], _typed(tmp_path, 9, seed=62)[0], _typed(tmp_path, 10, seed=63)[0]
    R = orc.euler_to_matrix(np.array([0.2, 1.1]), np.array([0.7, 2.0]), np.array([-0.4, 0.9]))
    dk = Docker(_tiny_model(), box_size=32, resolution=1.25, max_conf=20, rotations=R, device="cpu", lib=lib,
                coords_backend=CoordsBackend(lib=lib))
    first = dk.prepare(pa, pb, "SE3", slot=1)
    second = dk.prepare(pc, pb, "SE3", slot=1)
    with pytest.raises(Exception, match="overwritten"):
        dk.dockSE3(pa, pb, 2, prepared=first)
    dk.log = None
    dk.dockSE3(pc, pb, 2, prepared=second)                     # the pair the slot holds docks
    assert dk.path == "fused" and len(dk.top_list) == 20
    third = dk.prepare(pa, pb, "SE3", slot=0)                  # and the other slot is independent
    dk.dockSE3(pa, pb, 2, prepared=third)
    assert len(dk.top_list) == 20


def _cellwise_projection_checks(be, device, L=24, res=1.25):
    """CoordsBackend.project(cells=True) (dlpd_project_atoms_cells: only the cells the atoms' windows reach are cleared,
    accumulated and converted) against the dense projection: the same values wherever the map is set, zeros in the dense
    volume everywhere else, a map that covers every non-zero cell, and nothing written outside it (NaN-filled buffer)."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dlpd_cells_")
    _, coords, counts, offs = _typed(__import__("pathlib").Path(tmp), 10, seed=3)
    centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)
    R = torch.from_numpy(orc.euler_to_matrix([0.4, -2.0, 1.3], [1.0, 0.3, 2.2], [-0.7, 1.9, 0.2])).float()
    dense = be.project(coords, counts, offs, L, res, device, R=R, shift=centre)
    orig = torch.empty

    def nan_empty(*a, **kw):
        t = orig(*a, **kw)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    torch.empty = nan_empty
    try:
        sparse = be.project(coords, counts, offs, L, res, device, R=R, shift=centre, cells=True)
    finally:
        torch.empty = orig
    occ = sparse.dlpd_occupancy
    nc = (L + 3) // 4
    assert sparse.dlpd_unwritten and occ.shape == (3, nc, nc, nc) and 0 < int(occ.sum()) < occ.numel()
    live = occ.bool().repeat_interleave(4, 1).repeat_interleave(4, 2).repeat_interleave(4, 3)[:, None, :L, :L, :L].expand_as(dense)
    assert torch.equal(sparse[live], dense[live]) and bool((dense[~live] == 0).all()) and bool(torch.isnan(sparse[~live]).all())
    with pytest.raises(RuntimeError, match="sum_types"):
        be.project(coords, counts, offs, L, res, device, R=R, shift=centre, cells=True, sum_types=True)


def test_cellwise_projection_equals_the_dense_one_where_its_map_is_set_emulated(emu):
    _cellwise_projection_checks(CoordsBackend(lib=emu), "cpu")


@pytest.mark.gpu
def test_cellwise_projection_equals_the_dense_one_where_its_map_is_set_on_gpu():
    import __graft_entry__ as entry
    entry.build()
    _cellwise_projection_checks(CoordsBackend(), torch.device("cuda:0"), L=80)
