"""The conventions TorchProteinLibrary may define differently from this build (rotation pivot / scale / axis order /
direction, what VolumeConvolution(clip) clamps, the density splat, the atom typing) as PARAMETERS of the product path,
and the tool that pins them on a machine that has the library (scripts/calibrate_tpl.py).

TorchProteinLibrary is absent here (reference README.md:5; call sites src/Docker/Docker.py:29-40,218,221-225 and
src/Models/DockingModels.py:48,71), so the script is exercised against a STAND-IN package whose operators have known,
non-default conventions (built from the oracle): it must recover them, and the fixture it writes must make the product
kernels (emulated here; ``-m gpu``: the HIP library) reproduce the stand-in's outputs.  A real fixture committed as
tests/golden/tpl_conventions.json is replayed the same way on the GPU; without one that test reports itself as skipped."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd.ops import VolumeConvolution, VolumeRotation
from deeplocalproteindocking_amd.Utils.Conventions import VolumeConventions, kernel_matrices, rotation_scale
from deeplocalproteindocking_amd.Utils.FullAtom import NUM_ATOM_TYPES, CoordsBackend, atom_type
from oracle import docking_oracle as orc

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REAL_FIXTURE = os.path.join(ROOT, "tests", "golden", "tpl_conventions.json")


def _load_script():
    spec = importlib.util.spec_from_file_location("calibrate_tpl", os.path.join(ROOT, "scripts", "calibrate_tpl.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_stand_in(name, center, scale_rule, axis_order, transpose, clip_mode, splat, retype):
    """A package with TorchProteinLibrary's operator classes and call signatures whose conventions are the arguments."""
    pkg, vol, fam = types.ModuleType(name), types.ModuleType(name + ".Volume"), types.ModuleType(name + ".FullAtomModel")

    class VolumeRotationS(object):
        def __call__(self, volume, R):
            L = volume.shape[-1]
            c = {"L/2": L / 2.0, "grid_sample": (L - 1) / 2.0, "L/2-1": L / 2.0 - 1.0}[center]
            return orc.rotate_volume(volume.cpu(), R.cpu(), center=c, scale=orc.rotation_scale(scale_rule, L),
                                     axis_order=axis_order, transpose=transpose, dtype=torch.float64).float()

    class VolumeConvolutionS(object):
        def __init__(self, clip=None):
            self.clip = clip

        def __call__(self, v1, v2):
            return orc.correlate_fft(v1.cpu(), v2.cpu(), clip=self.clip, clip_mode=clip_mode, dtype=torch.float64).float()

    class TypedCoords2VolumeS(object):
        def __init__(self, box_size, resolution):
            self.L, self.res = box_size, resolution

        def __call__(self, coords, num_atoms_of_type, offsets):
            out = [orc.project_atoms(coords[b].numpy(), num_atoms_of_type[b].numpy(), offsets[b].numpy(), self.L, self.res, **splat)
                   for b in range(coords.shape[0])]
            return torch.from_numpy(np.stack(out)).float()

    be = CoordsBackend(lib=object(), atom_types=retype)

    class PDB2CoordsUnorderedS(object):
        def __call__(self, filenames):
            return be.pdb2coords(filenames)

    class Coords2TypedCoordsS(object):
        def __call__(self, coords, resnames, atomnames, num_atoms):
            return be.assign_types(coords, resnames, atomnames, num_atoms)

    vol.VolumeRotation, vol.VolumeConvolution, vol.TypedCoords2Volume = VolumeRotationS, VolumeConvolutionS, TypedCoords2VolumeS
    fam.PDB2CoordsUnordered, fam.Coords2TypedCoords = PDB2CoordsUnorderedS, Coords2TypedCoordsS
    pkg.Volume, pkg.FullAtomModel = vol, fam
    sys.modules[name], sys.modules[name + ".Volume"], sys.modules[name + ".FullAtomModel"] = pkg, vol, fam
    return pkg


STAND_IN = dict(center="grid_sample", scale_rule="(L-1)/L", axis_order="zyx", transpose=False, clip_mode="input",
                splat={"sigma": 0.8, "window": 3, "voxel_offset": 0.5, "norm": 1.5},
                retype={"LYS:NZ": 3, "SER:OG": 5})


@pytest.fixture(scope="module")
def fixture_path(tmp_path_factory):
    make_stand_in("dlpd_fake_tpl", **STAND_IN)
    out = str(tmp_path_factory.mktemp("tpl") / "tpl_conventions.json")
    rc = _load_script().main(["--module", "dlpd_fake_tpl", "--out", out, "--device", "cpu"])
    assert rc == 0
    return out


def test_calibration_recovers_the_stand_ins_conventions(fixture_path):
    d = json.load(open(fixture_path))
    c = d["conventions"]
    assert d["all_identified"]
    assert c["rotation_center"] == "grid_sample" and c["rotation_scale"] == "(L-1)/L"
    assert c["rotation_axis_order"] == "zyx" and c["rotation_transpose"] is False and c["clip_mode"] == "input"
    s = c["splat"]
    assert abs(s["sigma"] - 0.8) < 1e-3 and s["window"] == 3 and s["voxel_offset"] == 0.5 and abs(s["norm"] - 1.5) < 2e-3
    assert c["atom_types"]["LYS:NZ"] == 3 and c["atom_types"]["SER:OG"] == 5
    assert c["atom_types"]["ALA:CB"] == atom_type("ALA", "CB") and c["atom_types"]["TRP:NE1"] == atom_type("TRP", "NE1")
    conv = VolumeConventions.load(fixture_path)
    assert not conv.is_default() and conv.scale(80) == 79.0 / 80.0 and conv.pivot(40, 80) == 19.5


def test_calibration_of_the_build_defaults_and_an_unidentifiable_library(tmp_path):
    """The build's own conventions come back as the defaults; a library that does something outside the candidate family
    (here: a rotation about a pivot no rule describes) is reported as NOT identified (exit code 1), not silently fitted."""
    make_stand_in("dlpd_fake_tpl_default", "L/2", None, "xyz", False, "output",
                  {"sigma": 1.0, "window": 2, "voxel_offset": 0.0, "norm": 1.0}, {})
    out = str(tmp_path / "default.json")
    assert _load_script().main(["--module", "dlpd_fake_tpl_default", "--out", out, "--device", "cpu"]) == 0
    conv = VolumeConventions.load(out)
    conv.atom_types = {}
    assert conv.is_default()
    pkg = make_stand_in("dlpd_fake_tpl_odd", "L/2", None, "xyz", False, "output",
                        {"sigma": 1.0, "window": 2, "voxel_offset": 0.0, "norm": 1.0}, {})

    class Odd(object):
        def __call__(self, volume, R):
            return orc.rotate_volume(volume.cpu(), R.cpu(), center=volume.shape[-1] / 2.0 + 0.3, dtype=torch.float64).float()
    pkg.Volume.VolumeRotation = Odd
    out2 = str(tmp_path / "odd.json")
    assert _load_script().main(["--module", "dlpd_fake_tpl_odd", "--out", out2, "--device", "cpu"]) == 1
    d = json.load(open(out2))
    assert not d["all_identified"] and not d["evidence"]["rotation"]["identified"] and d["evidence"]["convolution"]["identified"]


def test_calibration_never_writes_a_file_that_runs_with_the_wrong_conventions(tmp_path, emu):
    """Two things a library could do that used to be written as free text beside DEFAULT values:
    (i) a pivot one voxel below the centre -- now a rule of its own ("L/2-1"), written and honoured;
    (ii) correlation arguments in the opposite roles -- no such convention exists in this build (MultiplyVolumes.py:13-47
    pins the roles), so the operator is reported as NOT identified (exit code 1) and nothing about it is written.
    A conventions file with a key the class does not know is refused, not read with the key dropped."""
    splat = {"sigma": 1.0, "window": 2, "voxel_offset": 0.0, "norm": 1.0}
    make_stand_in("dlpd_fake_tpl_low", "L/2-1", None, "xyz", False, "output", splat, {})
    out = str(tmp_path / "low.json")
    assert _load_script().main(["--module", "dlpd_fake_tpl_low", "--out", out, "--device", "cpu"]) == 0
    d = json.load(open(out))
    assert d["conventions"]["rotation_center"] == "L/2-1" and set(d["conventions"]) <= set(VolumeConventions.KEYS)
    conv = VolumeConventions.load(out)
    assert conv.pivot(80) == 39.0 and conv.pivot(40, 80) == 19.0
    assert max(replay_fixture(out, emu, "cpu").values()) < 2e-5          # the product kernels reproduce that library
    pkg = make_stand_in("dlpd_fake_tpl_swapped", "L/2", None, "xyz", False, "output", splat, {})
    plain = pkg.Volume.VolumeConvolution

    class Swapped(plain):
        def __call__(self, v1, v2):
            return plain.__call__(self, v2, v1)
    pkg.Volume.VolumeConvolution = Swapped
    out2 = str(tmp_path / "swapped.json")
    assert _load_script().main(["--module", "dlpd_fake_tpl_swapped", "--out", out2, "--device", "cpu"]) == 1
    d2 = json.load(open(out2))
    assert not d2["all_identified"] and not d2["evidence"]["convolution"]["identified"]
    assert "clip_mode" not in d2["conventions"] and set(d2["conventions"]) <= set(VolumeConventions.KEYS)
    with pytest.raises(Exception, match="Unknown convention keys"):
        VolumeConventions.from_dict({"clip_mode": "output", "correlation_arguments_swapped": True})


def test_docker_keeps_its_own_copy_of_the_conventions(emu):
    shared = VolumeConventions()
    dk = Docker(None, box_size=32, rotations=np.eye(3)[None], device="cpu", lib=emu, conventions=shared)
    dk.rotation_center = "grid_sample"
    assert shared.rotation_center is None and dk.rotation_pivot(32) == 15.5


def replay_fixture(path, lib, device):
    """The product operators, configured from the fixture, against the library outputs stored in it."""
    d = json.load(open(path))
    conv, probes = VolumeConventions.from_dict(d), d["probes"]
    worst = {}
    for pr in probes["rotation"]:
        L = pr["L"]
        vol = torch.tensor(pr["volume"], dtype=torch.float32, device=device).reshape(1, 1, L, L, L)
        R = torch.tensor(pr["R"], dtype=torch.float32, device=device).reshape(1, 3, 3)
        op = VolumeRotation(center=conv.pivot(L), lib=lib, scale=conv.rotation_scale, axis_order=conv.rotation_axis_order,
                            transpose=conv.rotation_transpose)
        got = op(vol, R).cpu().numpy()[0, 0]
        want = np.array(pr["out"])
        worst["rotation L=%d" % L] = np.abs(got - want).max() / np.abs(want).max()
    pr = probes["convolution"]
    L = pr["L"]
    v1 = torch.tensor(pr["v1"], dtype=torch.float32, device=device).reshape(1, 1, L, L, L)
    v2 = torch.tensor(pr["v2"], dtype=torch.float32, device=device).reshape(1, 1, L, L, L)
    got = VolumeConvolution(clip=pr["clip"], lib=lib, clip_mode=conv.clip_mode)(v1, v2).cpu().numpy()[0, 0]
    want = np.array(pr["out_clip"])
    worst["convolution(clip)"] = np.abs(got - want).max() / np.abs(want).max()
    be = CoordsBackend(lib=lib, splat=conv.splat)
    for pr in probes["splat"]:
        coords = torch.tensor([pr["position"]], dtype=torch.double)
        counts = torch.zeros(1, NUM_ATOM_TYPES, dtype=torch.int32)
        counts[0, pr["type"]] = 1
        offs = torch.zeros(1, NUM_ATOM_TYPES, dtype=torch.int32)
        got = be.project(coords, counts, offs, pr["L"], pr["resolution"], device).cpu().numpy()[0, pr["type"]]
        want = np.array(pr["volume"])
        worst["splat %s" % (pr["position"],)] = np.abs(got - want).max() / np.abs(want).max()
    return worst


def test_fixture_replays_through_the_emulated_kernels(fixture_path, emu):
    worst = replay_fixture(fixture_path, emu, "cpu")
    assert max(worst.values()) < 2e-3, worst           # (the splat parameters are fitted to ~1e-3)
    assert max(v for k, v in worst.items() if not k.startswith("splat")) < 1e-5, worst


@pytest.mark.gpu
def test_committed_tpl_fixture_replays_on_the_gpu():
    """Runs once a maintainer has committed the output of scripts/calibrate_tpl.py (a machine with TorchProteinLibrary)."""
    if not os.path.exists(REAL_FIXTURE):
        pytest.skip("no tests/golden/tpl_conventions.json: TorchProteinLibrary's conventions are still build-defined "
                    "(scripts/calibrate_tpl.py writes the file on a machine that has the library)")
    worst = replay_fixture(REAL_FIXTURE, None, "cuda")
    assert max(worst.values()) < 2e-3, worst


def _conv_case(L, C, seed):
    g = torch.Generator().manual_seed(seed)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.1, torch.randn(C, L, L, L, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    H = max(C // 2, 1)
    W = (torch.randn(H, C, generator=g) * 0.5, torch.randn(H, generator=g) * 0.1, torch.randn(1, H, generator=g), torch.randn(1, generator=g))
    return rec, lig, recf, ligf, W


def _engine_with_conventions(lib, device, L, C, conv, clip):
    """DockingEngine under non-default conventions: V of three rotations against the oracle with the same conventions
    written out explicitly (scale / axis order / direction NOT folded into R there)."""
    rec, lig, recf, ligf, W = _conv_case(L, C, 5)
    thr = 0.13 * L ** 3
    eng = DockingEngine(L, C, *W, clip=clip, threshold_clash=thr, max_conf=20, batch=3, device=device, lib=lib,
                        center=conv.pivot(L), rotation_scale=conv.scale(L), rotation_axis_order=conv.rotation_axis_order,
                        rotation_transpose=conv.rotation_transpose, clip_mode=conv.clip_mode)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    R = orc.euler_to_matrix([0.5, -2.1, 1.0], [1.1, 0.3, 2.0], [-0.4, 1.7, 0.2])
    V = eng.score_batch(torch.from_numpy(R).float().to(device).contiguous()).cpu().clone()
    sw = eng.switches()
    assert sw["clip_mode"] == conv.clip_mode and sw["rotation"]["axis_order"] == conv.rotation_axis_order
    for i in range(3):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        rot = lambda v: orc.rotate_volume(v, Rb, center=conv.pivot(L), scale=conv.scale(L), axis_order=conv.rotation_axis_order,
                                          transpose=conv.rotation_transpose)
        S = orc.score_volumes([rec[None]], [rot(lig[None])], *W, clip=clip, clip_mode=conv.clip_mode)[0]
        mask, norm = orc.clash_mask(recf[None, None], rot(ligf[None, None]), thr)
        sure = (norm[0] - thr).abs() > 1e-3 * thr
        assert ((V[i] - mask[0] * S).abs()[sure]).max() <= 1e-4 * S.abs().max()
    return V


@pytest.mark.parametrize("conv", [
    VolumeConventions(rotation_center="grid_sample", rotation_scale="(L-1)/L", rotation_axis_order="zyx"),
    VolumeConventions(rotation_transpose=True, clip_mode="input"),
    VolumeConventions(rotation_scale="L/(L-1)", clip_mode="none"),
], ids=["pivot+scale+zyx", "transposed+input-clamp", "scale+no-clamp"])
def test_engine_under_other_conventions_emulated(emu, conv):
    _engine_with_conventions(emu, "cpu", 32, 3, conv, clip=0.3)


@pytest.mark.gpu
def test_engine_under_other_conventions_on_gpu():
    for conv in (VolumeConventions(rotation_center="grid_sample", rotation_scale="(L-1)/L", rotation_axis_order="zyx"),
                 VolumeConventions(rotation_transpose=True, clip_mode="input")):
        _engine_with_conventions(None, "cuda", 64, 9, conv, clip=0.6)      # nine channels: the channels-last K1


def test_folded_matrices_equal_the_explicit_conventions():
    R = torch.from_numpy(orc.euler_to_matrix([0.3], [1.2], [-0.9])).float()
    v = torch.randn(1, 2, 10, 10, 10, generator=torch.Generator().manual_seed(1))
    for scale, ax, tr in ((0.9, "zyx", False), (1.1, "xyz", True), (1.0, "zyx", True)):
        a = orc.rotate_volume(v, R, center=4.5, scale=scale, axis_order=ax, transpose=tr)
        b = orc.rotate_volume(v, kernel_matrices(R, scale, ax, tr), center=4.5)
        assert (a - b).abs().max() < 1e-5
    assert rotation_scale("(L-1)/L", 80) == 79 / 80 and rotation_scale(None, 3) == 1.0 and rotation_scale(0.97, 3) == 0.97


def test_docker_takes_the_fixture_and_docks_under_it(fixture_path, emu):
    """Docker(conventions=<file>): the search on volumes under the fixture's conventions against the oracle's."""
    L, C, K = 32, 3, 30
    rec, lig, recf, ligf, W = _conv_case(L, C, 9)
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter
    filt = SimpleFilter([C])
    with torch.no_grad():
        filt.fc[0].weight.copy_(W[0]); filt.fc[0].bias.copy_(W[1]); filt.fc[2].weight.copy_(W[2]); filt.fc[2].bias.copy_(W[3])
    model = GlobalDockingModel(None, filt, threshold_clash=0.13 * L ** 3, clip=0.3, lib=emu)
    R = orc.euler_to_matrix([0.5, -2.1, 1.0, 2.4], [1.1, 0.3, 2.0, 0.7], [-0.4, 1.7, 0.2, -1.3])
    dk = Docker(model, box_size=L, resolution=1.0, max_conf=K, rotations=R, device="cpu", lib=emu, conventions=fixture_path)
    assert dk.conventions.clip_mode == "input" and dk.rotation_pivot(L) == (L - 1) / 2.0
    got = dk.dock_volumes([rec[None]], [lig[None]], recf, ligf, write=False)
    assert dk.path == "fused"
    want, Vs = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], R, *W, 0.13 * L ** 3, K,
                                clip=0.3, faithful_topk=False, return_V=True, clip_mode="input", rotation_center_offset=-0.5,
                                rotation_scale_rule="(L-1)/L", axis_order="zyx")
    scale = max(float(v.abs().max()) for v in Vs)
    assert len(got) == K and max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(got, want)) >= K - 4


def test_input_clamp_on_a_box_without_a_compiled_plan_takes_the_ops_path(emu):
    """clip_mode "input" is not combined with the embedded-box engine (the clamp acts on the ROTATED volumes, which the
    embedded path never materialises): Docker must route such a pair to the stand-alone ops -- before any engine buffer is
    allocated or kernel launched -- and the engine itself refuses the combination at construction."""
    L, C, K = 12, 3, 20
    rec, lig, recf, ligf, W = _conv_case(L, C, 11)
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter
    filt = SimpleFilter([C])
    with torch.no_grad():
        filt.fc[0].weight.copy_(W[0]); filt.fc[0].bias.copy_(W[1]); filt.fc[2].weight.copy_(W[2]); filt.fc[2].bias.copy_(W[3])
    thr = 0.13 * L ** 3
    model = GlobalDockingModel(None, filt, threshold_clash=thr, clip=0.3, lib=emu)
    R = orc.euler_to_matrix([0.5, -2.1, 1.0], [1.1, 0.3, 2.0], [-0.4, 1.7, 0.2])
    dk = Docker(model, box_size=L, resolution=1.0, max_conf=K, rotations=R, device="cpu", lib=emu,
                conventions=VolumeConventions(clip_mode="input"))
    got = dk.dock_volumes([rec[None]], [lig[None]], recf, ligf, write=False)
    assert dk.path == "ops" and dk.engine is None
    want, Vs = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], R, *W, thr, K,
                                clip=0.3, faithful_topk=False, return_V=True, clip_mode="input")
    scale = max(float(v.abs().max()) for v in Vs)
    assert len(got) == K and max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= 1e-4 * scale
    with pytest.raises(RuntimeError, match="not combined with embedded boxes"):
        DockingEngine(32, C, *W, clip=0.3, threshold_clash=thr, max_conf=K, batch=2, device="cpu", lib=emu, extent=12, clip_mode="input")
