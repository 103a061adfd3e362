"""CPU-side checks of the host mirror of the reference interface and of the C-ABI library."""
import io
import os
import re

import numpy as np
import pytest
import torch

import __graft_entry__ as entry
from deeplocalproteindocking_amd.Utils.Rotations import Rotations, generate_angles, generated_set_size

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built_lib():
    entry.build()
    from deeplocalproteindocking_amd._lib import DlpdLib
    return DlpdLib(entry.LIB)


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "dlpd.h")).read()
    assert not re.search(r"\bdlpd_debug_[a-z0-9_]+\s*\(", header)   # test hooks are declared in their own header ...
    debug = open(os.path.join(ROOT, "include", "dlpd_debug.h")).read()
    names = set(re.findall(r"\b(dlpd_[a-z0-9_]+)\s*\(", header)) | set(re.findall(r"\b(dlpd_debug_[a-z0-9_]+)\s*\(", debug))
    assert len(names) >= 15
    # ... and every call of the minimal set the header's first comment names is declared below it
    for n in ("dlpd_rfft3d_padded", "dlpd_receptor_pack", "dlpd_make_channels_last", "dlpd_zfft_channels_last", "dlpd_project_atoms",
              "dlpd_zfft_into", "dlpd_xy_correlate_packed", "dlpd_xy_correlate", "dlpd_zifft_filter_cand", "dlpd_topk_select_cand",
              "dlpd_topk_merge_tau", "dlpd_zifft_preact"):
        assert header.count(n) >= 2 and n in names, n
    from deeplocalproteindocking_amd._lib import SIGNATURES
    assert names == set(SIGNATURES), names ^ set(SIGNATURES)
    import ctypes
    dll = ctypes.CDLL(entry.LIB)
    for n in names:
        assert hasattr(dll, n), n
    assert built_lib.call("dlpd_version") >= 200
    # the library carries the hash of the sources it was built from; build() rebuilds on mismatch
    assert built_lib.source_hash() == entry.source_hash() == entry.library_hash() and len(entry.source_hash()) == 64
    assert built_lib.call("dlpd_grid_supported", 64) == 1 and built_lib.call("dlpd_grid_supported", 50) == 0
    assert built_lib.call("dlpd_hidden_pad", 24) == 24 and built_lib.call("dlpd_hidden_pad", 3) == 4
    assert built_lib.call("dlpd_topk_glist_bytes", 2000) == (2 + 4000) * 8


def test_product_library_ships_one_formulation_per_box(built_lib):
    """libdlpd.so holds only kernels a Docker path can reach: the channel-owning K3 at N = 128 / 160 and the
    transposed-slab K2 at N = 160 are TEST VARIANTS (tests/variants/libdlpd_variants.so, -DDLPD_TEST_VARIANTS)."""
    import subprocess
    nm = subprocess.run(["nm", "-C", entry.LIB], capture_output=True, text=True).stdout
    assert "k_zifft_filter_rs<128" in nm and "k_zifft_filter_rs<160" in nm and "k_xy_corr_q4<160, true>" in nm and "k_xy_corr_s4<80, true>" in nm
    assert "k_zifft_filter<64," in nm and "k_zifft_filter<80," in nm                      # boxes 32 / 40: the one formulation there
    for absent in ("k_rotate_zfft_cl_rs<", "k_xy_corr_quad<", "k_zifft_filter<128, 24, 1>", "k_zifft_filter_tiles<160", "k_zifft_filter<160, 24, 2>",
                   "k_zifft_filter<80, 24, 2>"):
        assert absent not in nm, absent
    assert built_lib.call("dlpd_orientation_supported", 64) == 1 and built_lib.call("dlpd_orientation_supported", 80) == 0
    src = open(os.path.join(ROOT, "deeplocalproteindocking_amd", "engine.py")).read()
    assert "environ" not in src and "getenv" not in src                              # kernel choices are constructor arguments only


def test_product_path_fails_loudly_without_gpu_or_library(tmp_path):
    from deeplocalproteindocking_amd._lib import DlpdLib
    from deeplocalproteindocking_amd.engine import DockingEngine
    with pytest.raises(RuntimeError, match="not found"):
        DlpdLib(str(tmp_path / "missing.so"))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            DockingEngine(32, 4, torch.zeros(2, 4), torch.zeros(2), torch.zeros(1, 2), torch.zeros(1), device="cpu")
        from deeplocalproteindocking_amd.ops import VolumeConvolution
        with pytest.raises(RuntimeError, match="no CPU path"):
            VolumeConvolution()(torch.zeros(1, 1, 32, 32, 32), torch.zeros(1, 1, 32, 32, 32))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deeplocalproteindocking_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(d, f)


def test_rotations_generated_sets_match_soi_sizes():
    sizes = {20: 1854, 15: 4392, 12: 8580, 10: 14868, 8: 29025, 6: 68760}
    for inc, n in sizes.items():
        ns, nphi = generated_set_size(inc)
        assert ns * nphi == n
    ang = generate_angles(20)
    assert ang.shape == (1854, 3)
    os.environ["DLPD_ROTATIONS_DIR"] = "/nonexistent-dir"
    try:
        # the reference's behaviour (Rotations.py:41,55) unless the caller opts in
        with pytest.raises(Exception, match="Can't find rotation angles"):
            Rotations(4, verbose=False)
        with pytest.warns(UserWarning, match="GENERATED"):
            r = Rotations(4, allow_generated=True, verbose=False)
        os.environ["DLPD_ALLOW_GENERATED_ROTATIONS"] = "1"
        with pytest.warns(UserWarning, match="GENERATED"):
            assert Rotations(20, verbose=False).source == "generated"
    finally:
        del os.environ["DLPD_ROTATIONS_DIR"]
        os.environ.pop("DLPD_ALLOW_GENERATED_ROTATIONS", None)
    assert r.source == "generated" and r.R.shape == (232020, 3, 3) and r.R.dtype == torch.float64
    RtR = torch.matmul(r.R[::997].transpose(1, 2), r.R[::997])
    assert (RtR - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-14
    assert (torch.linalg.det(r.R[::997]) - 1).abs().max() < 1e-14


@pytest.mark.skipif(not os.path.exists("/root/reference/data/oim06.eul"), reason="reference data not present")
def test_rotations_loader_reads_reference_files_incl_zero_padded_names(golden):
    os.environ["DLPD_ROTATIONS_DIR"] = "/root/reference/data"
    try:
        g = golden("g2_rotations.npz")
        for inc in (20, 15, 12, 10):
            r = Rotations(inc, verbose=False)
            R = r.R.numpy()
            assert R.shape[0] == int(g["n_%d" % inc])
            np.testing.assert_allclose(R[:8], g["first8_%d" % inc], atol=1e-15)
            np.testing.assert_allclose(R[-8:], g["last8_%d" % inc], atol=1e-15)
            np.testing.assert_allclose(R.sum(0), g["sum_%d" % inc], atol=1e-9)
            w = np.arange(1, R.shape[0] + 1, dtype=np.float64)[:, None, None]
            np.testing.assert_allclose((w * R).sum(0), g["wsum_%d" % inc], rtol=1e-12, atol=1e-6)
        assert Rotations(6, verbose=False).R.shape[0] == 68760      # reference loader cannot open these
        assert Rotations(8, verbose=False).R.shape[0] == 29025
    finally:
        del os.environ["DLPD_ROTATIONS_DIR"]


def _docker(max_conf=6, box_size=4, rotations=None):
    from deeplocalproteindocking_amd.Docker import Docker
    if rotations is None:
        rotations = np.tile(np.eye(3), (4, 1, 1))
    return Docker(docking_model=None, angle_inc=20, box_size=box_size, resolution=1.25, max_conf=max_conf,
                  rotations=rotations, device="cpu")


def test_docker_write_conformations_matches_reference_text(golden):
    g = golden("g4_write_conformations.npz")
    nrot = int(g["rot_ids"].max()) + 1
    R = np.zeros((nrot, 3, 3))
    for k, i in enumerate(g["rot_ids"]):
        R[int(i)] = g["R_used"][k]
    dk = _docker(rotations=R)
    dk.top_list = [(int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])) for t in g["top_list"]]
    dk.log = io.StringIO()
    dk.write_conformations()
    assert dk.log.getvalue() == bytes(g["text"]).decode()
    dk.randomize_rot = True
    dk.randR = torch.from_numpy(g["randR"]).unsqueeze(0)
    dk.log = io.StringIO()
    dk.write_conformations()
    assert dk.log.getvalue() == bytes(g["text_rand"]).decode()
    dk.log = None


def test_docker_new_log_resume_rule(tmp_path):
    dk = _docker()
    f = str(tmp_path / "t.dat")
    assert dk.new_log(f, rewrite=False) is True                 # new file
    dk.log.write("1\n"); dk.cleanup()
    assert dk.new_log(f, rewrite=False) is True                 # one line: not finished -> redo
    dk.log.write("1\n\n2\n"); dk.cleanup()
    assert dk.new_log(f, rewrite=False) is False                # >1 non-blank lines: skip (Docker.py:73-74)
    assert dk.new_log(f, rewrite=True) is True
    dk.cleanup()


def test_docker_interface_surface():
    import inspect
    from deeplocalproteindocking_amd.Docker import Docker
    sig = inspect.signature(Docker.__init__)
    assert list(sig.parameters)[:7] == ["self", "docking_model", "angle_inc", "box_size", "resolution", "max_conf",
                                        "randomize_rot"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["angle_inc"], d["box_size"], d["resolution"], d["max_conf"], d["randomize_rot"]) == (15.0, 80, 1.25, 1000, False)
    for m in ("load_batch", "new_log", "cleanup", "update_top", "write_conformations", "dockE3", "dockSE3",
              "dock_volumes"):
        assert callable(getattr(Docker, m))
    # the reference's positional arguments first (Docker.py:135,184); `prepared` is an optional superset keyword
    for m in (Docker.dockSE3, Docker.dockE3):
        p = inspect.signature(m).parameters
        assert list(p) == ["self", "ureceptor", "uligand", "batch_size", "prepared"] and p["prepared"].default is None
    dk = _docker()
    # no coords_backend argument (the reference's constructor has none): the build's own is created on first use
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    assert dk.coords_backend is None and isinstance(dk._need_backend(), CoordsBackend)
    assert dk.launch_batch == 32                                # launches are not sized by the caller's batch_size
    assert dk.box_length == 5.0 and dk.shard(10).tolist() == list(range(10))
    dk.rank, dk.world_size = 1, 4
    assert dk.shard(10).tolist() == [1, 5, 9]
    # the reference's operator attributes (Docker.py:29-40) exist for subclasses that touch them
    for name in ("rotate", "translate", "translation", "project", "convolve", "vol_rotate", "pdb2coords", "assignTypes"):
        assert callable(getattr(dk, name)), name
    moved = dk.translate(torch.zeros(1, 6, dtype=torch.double), torch.ones(1, 3, dtype=torch.double), torch.tensor([2], dtype=torch.int32))
    assert moved.tolist() == [[1.0] * 6]
    # pivot of the volume rotation: index L/2 by default, (L-1)/2 for torch's sampling-grid centre, scaled to coarse grids
    assert dk.rotation_pivot(4) == 2.0 and dk.rotation_pivot(2) == 1.0
    dk.rotation_center = "grid_sample"
    assert dk.rotation_pivot(4) == 1.5 and dk.rotation_pivot(2) == 0.5
    dk.rotation_center = 1.75
    assert dk.rotation_pivot(4) == 1.75 and dk.rotation_pivot(2) == 0.875


def test_simple_filter_matches_reference_structure(golden):
    from deeplocalproteindocking_amd.Models import SimpleFilter
    torch.manual_seed(303)                      # same seed as the fixture generator
    f = SimpleFilter([16, 32])
    g = golden("g5_global_forward.npz")
    np.testing.assert_array_equal(f.fc[0].weight.detach().numpy(), g["multires_W1"])   # same init stream
    assert f.fc[0].weight.shape == (24, 48) and f.fc[2].weight.shape == (1, 24)


def test_merge_entries_is_deterministic_and_stable():
    from deeplocalproteindocking_amd.engine import DeviceTopList
    a = (np.array([0, 2]), np.array([5, 6]), np.array([-3.0, -1.0], np.float32), np.array([0, 1]))
    b = (np.array([1, 1, 3]), np.array([7, 8, 9]), np.array([-3.0, -1.0, -1.0], np.float32), np.array([0, 1, 0]))
    rot, idx, score, pick = DeviceTopList.merge_entries([b, a], 4)
    assert rot.tolist() == [0, 1, 1, 2] and idx.tolist() == [5, 7, 8, 6]


def test_se3_scalar_representation_is_equivariant_and_shaped_like_the_reference():
    """ProteinRepresentationModels.py:23-76: outputs [2m @ L^3, 4m @ (L/2)^3]; isotropic kernels commute
    with the grid's exact (90 degree) rotations, the property the volume-rotation search relies on."""
    from deeplocalproteindocking_amd.Models import SE3MultiResReprScalar
    torch.manual_seed(3)
    model = SE3MultiResReprScalar(multiplier=2).eval()
    assert model.get_num_outputs() == [4, 8]
    L = 12
    x = torch.zeros(1, 11, L, L, L)
    x[:, :, 3:9, 3:9, 3:9] = torch.rand(1, 11, 6, 6, 6)          # support away from the faces
    with torch.no_grad():
        v0, v1 = model(x)
        r0, r1 = model(torch.rot90(x, 1, dims=(2, 3)))
    assert v0.shape == (1, 4, L, L, L) and v1.shape == (1, 8, L // 2, L // 2, L // 2)
    assert torch.allclose(torch.rot90(v0, 1, dims=(2, 3)), r0, atol=1e-5)
    k = model.sequence_res0[0].kernel()
    assert torch.allclose(k, k.flip(2)) and torch.allclose(k, k.transpose(2, 3))      # radial kernels
    # even box + stride 2 samples a rotated lattice, so only the full-resolution branch is exactly equivariant
    assert r1.shape == v1.shape


def test_bench_reads_the_dominant_kernels_traffic_from_a_counter_pass(tmp_path, monkeypatch):
    """bench.py's ``roofline.traffic``: the parser of the two rocprofv3 --pmc child passes (FETCH_SIZE x 2 + WRITE_SIZE,
    KB per dispatch summed over a launch's rows, averaged over the launches of the stage's main kernel); the profiler and
    the child are replaced by a canned counter_collection.csv."""
    import subprocess
    import bench

    def fake_run(cmd, cwd=None, env=None, timeout=None):
        out = cmd[cmd.index("-d") + 1]
        counter = cmd[cmd.index("--pmc") + 1]
        os.makedirs(os.path.join(out, "host"), exist_ok=True)
        rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
        val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 3000.0}[counter]
        for d in (1, 2):                                  # two launches, the counter split over two rows each
            rows += ['%d,"void k_xy_corr<128, 1>(float2 const*, int)",%s,%f' % (d, counter, val / 2)] * 2
        rows.append('3,"void k_xy_corr<128, 0>(float2 const*, int)",%s,77777' % counter)     # receptor prep: not the stage
        rows.append('4,"void k_zifft_filter_rs<128, 24, 1>(void const*)",%s,55555' % counter)
        open(os.path.join(out, "host", "1_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
        return 0

    monkeypatch.setattr(bench, "run_in_own_group", fake_run)
    monkeypatch.setattr(os.path, "exists", lambda p: True)
    import sys
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    args = bench.parse_args()
    # K2<128>: the raw read count is taken apart by load shape (scripts/micro/fetch_size_shapes.hip: A rows tallied at 0.500 of
    # their bytes, the receptor's 64-byte runs at 0.727) -- a canned 4 GB raw count at config 2's sizes
    A, rec = 16 * 49 * 65 * 64 * 64 * 8.0, 49 * 65 * 128 * 128 * 8.0
    reads, refetch = bench.k2_natural_read_bytes(0.5 * A + 0.727 * 3.0 * rec, 16, 49, 64)
    assert abs(reads - (A + 3.0 * rec)) < 1.0 and abs(refetch - 3.0) < 1e-9
    traffic, src = bench.live_pmc_traffic(args, "k2_xy_corr", 128)
    A_launch = A * args.batch / 16.0                              # (the default launch holds args.batch rotations)
    assert traffic == A_launch + 3000.0 * 1024.0 and "measured in this run" in src and "0.727" in src     # (1000 KB raw < the A rows' share)
    # every other stage: x 2 (whole 128-byte lines tallied at 64 bytes)
    traffic, src = bench.live_pmc_traffic(args, "k3_zifft_filter", 128)
    assert traffic == (2.0 * 55555.0 + 55555.0) * 1024.0 and "measured in this run" in src
    assert "raw FETCH_SIZE %d bytes x 2" % (55555 * 1024) in src and "raw WRITE_SIZE %d bytes" % (55555 * 1024) in src      # auditable
    assert bench.live_pmc_traffic(args, "topk_select", 128) == (None, None)
    # K3's wide configurations read 64-byte runs (8-row tiles): their FETCH_SIZE is taken at face value
    assert bench.fetch_scale("k3_zifft_filter", 160, 24) == 2.0 and bench.fetch_scale("k3_zifft_filter", 160, 32) == 1.0
    assert bench.fetch_scale("k3_zifft_filter", 128, 32) == 2.0 and bench.fetch_scale("k3_zifft_filter", 128, 48) == 1.0
    assert bench.fetch_scale("k2_xy_corr", 160, 48) == 2.0


def _g7_net(g, m):
    """The build's E3MultiResRepr4x4 holding the reference class's seeded weights (fixture G7), strictly loaded."""
    import json
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4
    keys = json.loads(bytes(g["m%d_keys" % m]).decode())
    net = E3MultiResRepr4x4(multiplier=m).eval()
    assert list(net.state_dict().keys()) == keys                  # the reference's module tree, key for key
    net.load_state_dict({k: torch.from_numpy(g["m%d_sd_%s" % (m, k)]) for k in keys}, strict=True)
    return net


def test_e3_plugin_reproduces_the_reference_class(golden):
    """Fixture G7: the reference's own E3MultiResRepr4x4 (ProteinRepresentationModels.py:78-128, plain torch, imported
    with se3cnn mocked) with seeded weights on a seeded (1, 11, 12^3) input.  The build's class loads that state dict
    with strict=True and reproduces both outputs to 2e-6 of the largest output value (the bias-free stack shrinks the
    signal to 1e-2 / 1e-3, so the tolerance is relative; torch's own CPU kernels differ by up to 4e-7 with the thread
    count)."""
    g = golden("g7_e3_plugin.npz")
    for m in (1, 8):
        net = _g7_net(g, m)
        assert net.get_num_outputs() == g["m%d_num_outputs" % m].tolist() == [2 * m, 4 * m]
        with torch.no_grad():
            v = net(torch.from_numpy(g["m%d_input" % m]))
        for got, want in zip(v, (g["m%d_out0" % m], g["m%d_out1" % m])):
            assert tuple(got.shape) == want.shape
            assert np.abs(got.numpy() - want).max() <= 2e-6 * np.abs(want).max()
            assert np.abs(want).max() > 1e-4                       # (not a comparison of zeros)


def test_e3_plugin_on_the_emulated_hip_convolutions_reproduces_the_reference_class(golden, emu):
    """The same fixture through ops.conv3d / ops.maxpool3d_5s2 (the kernels' sources, emulated): 16 / 32 channels."""
    g = golden("g7_e3_plugin.npz")
    net = _g7_net(g, 8)
    net.hip_lib = emu
    with torch.no_grad():
        v = net(torch.from_numpy(g["m8_input"]))
    for got, want in zip(v, (g["m8_out0"], g["m8_out1"])):
        assert np.abs(got.numpy() - want).max() <= 1e-5 * np.abs(want).max()
    # the network on an input that is zero away from a small blob (what a protein's density splat looks like in its box),
    # with the all-zero tiles skipped (the default) and without: the same bits in both outputs
    x = torch.zeros(1, 11, 16, 16, 16)
    x[:, :, 0:2, 0:3, 1:3] = torch.randn(1, 11, 2, 3, 2, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        skipping = net._run(net.conv1, x)                          # (the five full-resolution layers)
        net.use_tile_occupancy = False
        dense = net._run(net.conv1, x)
    assert torch.equal(skipping, dense) and float(dense.abs().max()) > 0
    assert float((dense == 0).float().mean()) > 0.3                # (there WAS something to skip)


def test_select_model_follows_the_reference(golden, monkeypatch):
    """Fixture G7, second half: what the reference's select_model (local_train.py:19-43) returns for the E3 branch --
    classes, multiplier, filter sizing -- and how it fails for unknown names (Exception with the reference's args)."""
    import json
    import os
    import sys
    import types
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeplocalproteindocking_amd")
    monkeypatch.syspath_prepend(pkg)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, device=None: self)     # no GPU in this test
    import local_train
    want = json.loads(bytes(golden("g7_e3_plugin.npz")["select_model_json"]).decode())
    pm, cf = local_train.select_model(types.SimpleNamespace(group="E3", model="E3MultiResRepr4x4", filter="SimpleFilter"))
    assert (type(pm).__name__, type(cf).__name__) == (want["repr_class"], want["filter_class"])
    assert pm.get_num_outputs() == want["num_outputs"]
    assert {k: list(v.shape) for k, v in pm.state_dict().items()} == want["repr_shapes"]
    assert {k: list(v.shape) for k, v in cf.state_dict().items()} == want["filter_shapes"]
    for tag, bad in (("group", dict(group="XX", model="E3MultiResRepr4x4", filter="SimpleFilter")),
                     ("model", dict(group="E3", model="Nope", filter="SimpleFilter")),
                     ("filter", dict(group="E3", model="E3MultiResRepr4x4", filter="Nope"))):
        try:
            local_train.select_model(types.SimpleNamespace(**bad))
            got = None
        except Exception as exc:
            got = [type(exc).__name__] + [str(a) for a in exc.args]
        assert got == want["errors"][tag], (tag, got, want["errors"][tag])


def test_engine_gating_on_the_product_dispatch_tables():
    """The emulated library of the other CPU tests is built WITH the test-variant kernels (it must hold both sides of
    every bit-identity test), so it answers dlpd_orientation_supported(80) = 1 and accepts every K3 formulation.  Here
    the same sources are compiled as libdlpd.so compiles them (no -DDLPD_TEST_VARIANTS): the engine must find box 80
    without a transposed-slab K2 and visit it in one orientation, and a K3 formulation the product does not ship at a
    box must be refused (DLPD_ERR_UNSUPPORTED), not silently replaced."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from deeplocalproteindocking_amd._lib import DlpdLib
    from deeplocalproteindocking_amd.engine import DockingEngine
    lib = DlpdLib(build_emu.build(variants=False))
    assert lib.call("dlpd_orientation_supported", 64) == 1 and lib.call("dlpd_orientation_supported", 80) == 0
    z = lambda *s: torch.zeros(*s)
    eng80 = DockingEngine(80, 2, z(1, 2), z(1), z(1, 1), z(1), max_conf=4, batch=1, device="cpu", lib=lib, channels_last=False)
    assert eng80.orient is False and eng80.switches()["k1_slab_orientation"] is False
    eng64 = DockingEngine(64, 2, z(1, 2), z(1), z(1, 1), z(1), max_conf=4, batch=1, device="cpu", lib=lib, channels_last=False)
    assert eng64.orient is True
    # K3 at box 64: the product ships the role-split formulation only (the channel-owning one is a test variant)
    bad = DockingEngine(64, 1, z(1, 1), z(1), z(1, 1), z(1), max_conf=4, batch=1, device="cpu", lib=lib, k3_form=1, has_clash=False)
    bad.set_receptor(z(1, 64, 64, 64))
    bad.set_ligand(z(1, 64, 64, 64))
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        bad.score_batch(torch.eye(3).reshape(1, 3, 3).contiguous())
    # K1 at boxes 64 / 80: the role-split formulation is a test variant as well
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        lib.call("dlpd_zfft_channels_last_form", 1, 1, 1, 1, 4, 5, 0, 64, 32.0, 0, 2, 0)


def test_bench_secondary_roofline_from_the_sq_counters():
    """bench.py's ``roofline.secondary``: the counter averages of K2<128> as rocprofv3 reported them on an MI355X
    (profiles/r05_pmc_units.txt) give the figures the calibration was checked against -- 2.0 GHz from GRBM_GUI_ACTIVE / 8
    XCDs over the launch duration, 1.9 resident waves per SIMD for a kernel compiled for two, LDS active 45 % of the
    cycles, vector issue >= 21 % -- and a kernel without a usable cycle counter falls back to an ASSUMED clock, flagged."""
    import bench
    c = {"GRBM_GUI_ACTIVE": 4.178e7, "SQ_ACTIVE_INST_VALU": 5.654e8, "SQ_INSTS_VALU": 5.651e8, "SQ_LDS_IDX_ACTIVE": 5.985e8,
         "SQ_LDS_BANK_CONFLICT": 1.076e8, "SQ_WAVE_CYCLES": 2.583e9}
    s = bench.secondary_from_counters(c, 2596.1e3)
    assert abs(s["clock_GHz"] - 2.012) < 0.01 and not s["clock_assumed"]
    assert abs(s["waves_per_simd"] - 1.93) < 0.02
    assert abs(s["lds_active_frac"] - 0.448) < 0.005 and abs(s["valu_issue_frac"] - 0.211) < 0.005
    assert abs(s["valu_active_frac"] - 0.423) < 0.005
    assert abs(s["valu_plus_lds_frac"] - (s["valu_issue_frac"] + s["lds_active_frac"])) < 1e-12
    t = bench.secondary_from_counters(dict(c, GRBM_GUI_ACTIVE=0.0), 2596.1e3)
    assert t["clock_assumed"] and t["clock_GHz"] == 2.0
    assert bench.secondary_from_counters({}, None) is None


def test_no_product_path_runs_the_plugin_beside_the_pipeline():
    """EXPERIMENTS.md R5: the plugin's bf16 x 3 matrix-instruction convolution co-resident with the pipeline's LDS kernels
    changes low bits of the pipeline's results on this hardware.  The two places that could put it on a second stream -- an
    overlapped dockE3 loop and a preparing stream in the sweep -- were measured (no gain) and removed."""
    import inspect
    from deeplocalproteindocking_amd import local_test
    from deeplocalproteindocking_amd.Docker import Docker
    assert not hasattr(Docker, "E3_OVERLAP")
    assert "prepare_stream" not in inspect.signature(local_test.sweep).parameters
    assert "torch.cuda.Stream(" not in inspect.getsource(local_test) and "cuda.Stream(" not in inspect.getsource(Docker)
    assert inspect.signature(Docker.prepare).parameters["stream"].default is None


def test_product_library_has_no_lds_atomic_instruction():
    """Round 5 traced wrong low mantissa bits in OTHER kernels' results to the LDS atomics of k_topk_hist running beside a
    bf16 matrix-instruction kernel on the same CU (EXPERIMENTS.md R5).  The product library therefore carries no LDS atomic at
    all (counts go through ballots, butterflies and plain LDS words): every gfx950 code object inside csrc/libdlpd.so is
    disassembled and searched for the ds_* read-modify-write instructions."""
    import re
    import struct
    import subprocess
    import tempfile
    import __graft_entry__ as entry
    entry.build()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    assert os.path.exists(objdump)
    data = open(entry.LIB, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    lines = kernels = 0
    found = set()
    for m in re.finditer(re.escape(magic), data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        q = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" not in triple or not size:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[p + off:p + off + size])
                f.flush()
                text = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True, check=True).stdout
            lines += text.count("\n")
            kernels += len(re.findall(r"^[0-9a-f]+ <_Z\w+>:", text, flags=re.M))
            found |= set(re.findall(r"\bds_(?:add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|cmpswap|wrxchg|pk_add|append|consume)\w*", text))
    assert lines > 100000 and kernels > 50, (lines, kernels)          # the scan really saw the library's kernels
    assert not found, sorted(found)


def test_history_holds_no_compiled_object():
    """Built artefacts travel to the GPU box with the snapshot but stay out of the history (.gitignore): no tracked file
    may be an ELF image or an offload bundle (a bundler extraction once committed 18 code objects)."""
    import subprocess
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout (the GPU box's snapshot has no .git)")
    files = subprocess.run(["git", "ls-files", "-z"], cwd=ROOT, capture_output=True, check=True).stdout.split(b"\0")
    bad = []
    for f in files:
        p = os.path.join(ROOT.encode(), f)
        if not f or not os.path.isfile(p):
            continue
        with open(p, "rb") as fh:
            head = fh.read(24)
        if head[:4] == b"\x7fELF" or head.startswith(b"__CLANG_OFFLOAD_BUNDLE__") or b".hipv4-" in f or b".host-x86_64" in f:
            bad.append(f.decode())
    assert not bad, bad
