"""The reference's unchanged caller on this build (SURVEY.md 8(b) seam 1, BASELINE north_star: "drops in
under local_test.py unchanged"): scripts/replay_local_test.py issues local_test.py:5-14,44-71 call for
call -- reference constructor kwargs only, ``select_model``, ``GlobalDockingModel(...).cuda()``,
``.load``, ``new_log``, ``dockSE3(rec, lig, batch_size=2)`` -- on a synthetic DockingBenchmark directory
with protein-sized two-chain structures.  Runs in fresh processes (the package reads its directories
from the environment at import)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def make_benchmark(root, targets=(("1SYN", 230, 120, 21),)):
    """DATA_DIR/DebugDockingBenchmark in the layout SplitComplexBenchmark.py:20-46 reads with
    struct_folder='Matched' (local_test.py:46): Table.csv + Matched/<pdb>_{r_u,l_u,r_b,l_b}.pdb, <pdb>_b.pdb."""
    from synth_pdb import write_protein_like_pdb
    bench = os.path.join(root, "data", "DebugDockingBenchmark")
    os.makedirs(os.path.join(bench, "Matched"))
    rows = ["Complex\tCat.\tPDB ID 1\tProtein 1\tPDB ID 2\tProtein 2\tI-RMSD\tdASA", "Rigid-body (1)"]
    for name, nrec, nlig, seed in targets:
        rows.append("%s_A:B\tOX\t%s_A\tsynthetic receptor\t%s_B\tsynthetic ligand\t1.0\t1500" % (name, name, name))
        for suffix, nres, sd, off in (("_r_u", nrec, seed, (30.0, -12.0, 5.0)), ("_l_u", nlig, seed + 1, (-40.0, 8.0, 60.0)),
                                      ("_r_b", nrec, seed, (0.0, 0.0, 0.0)), ("_l_b", nlig, seed + 1, (25.0, 0.0, 0.0))):
            write_protein_like_pdb(os.path.join(bench, "Matched", name + suffix + ".pdb"), nres, sd, offset=off)
        write_protein_like_pdb(os.path.join(bench, "Matched", name + "_b.pdb"), nrec + nlig, seed + 2)
    rows += ["Medium Difficulty (0)", "Difficult (0)"]
    with open(os.path.join(bench, "Table.csv"), "w") as f:
        f.write("\n".join(rows[:2] + rows[2:]) + "\n")
    return bench


def _replay(root, log_name, extra_env=None, extra_args=()):
    env = dict(os.environ)
    env.update({"DLPD_DATA_DIR": os.path.join(root, "data"), "DLPD_MODELS_DIR": os.path.join(root, "models"),
                "DLPD_LOG_DIR": os.path.join(root, log_name), "DLPD_ALLOW_GENERATED_ROTATIONS": "1",
                "PYTHONDONTWRITEBYTECODE": "1"})
    env.pop("DLPD_LAUNCH_BATCH", None)
    env.update(extra_env or {})
    os.makedirs(env["DLPD_LOG_DIR"], exist_ok=True)
    os.makedirs(os.path.join(env["DLPD_LOG_DIR"], "LocalDebugSE3"), exist_ok=True)
    os.makedirs(os.path.join(env["DLPD_LOG_DIR"], "LocalDebugE3"), exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "replay_local_test.py"), "-angle_inc", "20", "-seed", "7",
           "-init_weights", "1", "-report", "1", "-threshold_clash", "40.0"] + list(extra_args)
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rep = [l for l in out.stdout.splitlines() if l.startswith("REPLAY ")]
    return json.loads(rep[-1][len("REPLAY "):]), out.stdout


@pytest.mark.gpu
def test_reference_driver_replay_dockSE3_and_dockE3(tmp_path):
    import __graft_entry__ as entry
    entry.build()
    root = str(tmp_path)
    make_benchmark(root)
    rep, stdout = _replay(root, "logA", extra_args=["-rewrite", "1"])
    t = rep["targets"][0]
    assert "Processing 1SYN" in stdout and t["path"] == "fused" and t["launch_batch"] == 32 and t["rotations"] == 1854
    dat = os.path.join(rep["test_dir"], "1SYN.dat")
    lines = open(dat).read().strip().split("\n")
    assert len(lines) == 2000 and all(len(l.split("\t")) == 13 for l in lines)
    scores = [float(l.split("\t")[12]) for l in lines]
    assert scores == sorted(scores)                               # ranked, ascending (Docker.py:104)
    # the same search with launches literally sized by the caller's batch_size=2: byte-identical .dat
    # (deterministic splat + batch-independent merge), and what decoupling the launch batch buys
    rep2, _ = _replay(root, "logB", extra_env={"DLPD_LAUNCH_BATCH": "2"}, extra_args=["-rewrite", "1"])
    t2 = rep2["targets"][0]
    assert t2["launch_batch"] == 2
    assert open(os.path.join(rep2["test_dir"], "1SYN.dat")).read() == open(dat).read()
    print("dockSE3 via the reference's calls (batch_size=2): %.0f rot/s at launch batch 32, %.0f rot/s if the "
          "launches were sized by the caller (2)" % (t["rot_per_s"], t2["rot_per_s"]))
    # resume rule (local_test.py:65 with rewrite=0): the finished target is skipped
    rep3, stdout3 = _replay(root, "logA", extra_args=["-rewrite", "0"])
    assert "Skipping 1SYN" in stdout3 and rep3["targets"] == []
    # the E3 branch of the driver (local_test.py:67)
    rep4, _ = _replay(root, "logA", extra_args=["-rewrite", "1", "-group", "E3", "-model", "E3MultiResRepr4x4",
                                                "-experiment", "LocalDebugE3"])
    t4 = rep4["targets"][0]
    assert t4["path"] == "fused" and t4["poses"] == 2000
    assert len(open(os.path.join(rep4["test_dir"], "1SYN.dat")).read().strip().split("\n")) == 2000


@pytest.mark.gpu
def test_reference_driver_replay_scores_equal_the_oracle(tmp_path):
    """The poses the unchanged call sequence writes -- not only their count and order: a 10-row ``oim20.eul`` in
    DLPD_ROTATIONS_DIR makes local_test.py's own calls (``Docker(..., angle_inc=20, randomize_rot=True)``,
    ``dockSE3(rec, lig, batch_size=2)``) search ten rotations; the .dat is compared line by line with the oracle
    restatement of Docker.py:184-238 on the same files, checkpoint and random receptor rotation: scores within
    1e-4 of max|V|, pose columns identical except for swaps inside that band."""
    import numpy as np
    import torch
    import __graft_entry__ as entry
    entry.build()
    from test_atoms import _dock_reference_shape
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    from deeplocalproteindocking_amd.Utils.Rotations import euler_to_matrices
    root = str(tmp_path)
    make_benchmark(root, targets=(("1SYN", 60, 40, 21),))
    rotdir = os.path.join(root, "rotations")
    os.makedirs(rotdir)
    ang = np.random.RandomState(17).uniform(-np.pi, np.pi, size=(10, 3))
    ang[:, 1] = np.abs(ang[:, 1])
    np.savetxt(os.path.join(rotdir, "oim20.eul"), ang, fmt="%.9f")
    ang = np.loadtxt(os.path.join(rotdir, "oim20.eul")).reshape(-1, 3)     # what the loader parses
    rep, _ = _replay(root, "logC", extra_env={"DLPD_ROTATIONS_DIR": rotdir, "DLPD_ALLOW_GENERATED_ROTATIONS": ""},
                     extra_args=["-rewrite", "1", "-threshold_clash", "3.0"])
    t = rep["targets"][0]
    assert t["rotations"] == 10 and t["path"] == "fused" and t["poses"] == 2000
    got = [[float(v) for v in l.split("\t")] for l in open(os.path.join(rep["test_dir"], "1SYN.dat")).read().strip().split("\n")]
    # the same model from the checkpoint the replay wrote, on the CPU
    repr_ = SE3MultiResReprScalar(multiplier=8)
    model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=3.0)
    model.load(os.path.join(root, "models", "LocalDebugSE3"), epoch=299)
    R = euler_to_matrices(ang[:, 0], ang[:, 1], ang[:, 2])
    randR = np.asarray(t["randR"], dtype=np.float64)
    want, scale = _dock_reference_shape(CoordsBackend(), model, t["receptor"], t["ligand"], R, 80, 1.25, 2000, randR=randR)
    # the oracle's list in the .dat's own format (Docker.write_conformations: index wrap, resolution, randR undone)
    ref = Docker(model, angle_inc=20, box_size=80, resolution=1.25, max_conf=2000, rotations=R, device="cpu")
    ref.randomize_rot, ref.randR = True, torch.from_numpy(randR).reshape(1, 3, 3)
    ref.top_list = want
    assert ref.new_log(os.path.join(root, "oracle.dat"))
    ref.write_conformations()
    ref.cleanup()
    exp = [[float(v) for v in l.split("\t")] for l in open(os.path.join(root, "oracle.dat")).read().strip().split("\n")]
    assert len(got) == len(exp) == 2000
    tol = 1e-4 * scale
    assert max(abs(a[12] - b[12]) for a, b in zip(got, exp)) <= tol          # rank by rank
    # pose by pose: every written pose is a pose of the oracle's list with the same score (to the tolerance) -- rows
    # may be swapped only among scores closer than the band, and only poses within the band of the K-th score may
    # differ between the two lists (they are cut off at different sides of the boundary)
    key = lambda row: tuple(int(round(v * 1e4)) for v in row[:12])
    exp_score = {key(r): r[12] for r in exp}
    missing = [r for r in got if key(r) not in exp_score]
    assert all(abs(r[12] - exp[-1][12]) <= tol for r in missing), len(missing)
    assert len(missing) <= 20
    assert max(abs(r[12] - exp_score[key(r)]) for r in got if key(r) in exp_score) <= tol
    same_row = sum(1 for a, b in zip(got, exp) if key(a) == key(b))
    assert same_row >= 1800, same_row                                        # most rows are not even swapped


def test_synthetic_benchmark_directory_is_what_the_loader_reads(tmp_path):
    from deeplocalproteindocking_amd.Dataset import get_benchmark_stream
    bench = make_benchmark(str(tmp_path), targets=(("1SYN", 30, 20, 3), ("2SYN", 25, 15, 9)))
    items = list(get_benchmark_stream(bench, struct_folder="Matched", subset="Table.csv", debug=False))
    assert [i[0][0] for i in items] == ["1SYN", "2SYN"]
    assert all(os.path.exists(i[k][0]) for i in items for k in (1, 2, 3, 4, 5)) and int(items[0][6][0]) == 1


def _same_dat_up_to_the_parity_band(a, b, n=2000):
    """Two .dat files of one target: byte-identical, or -- for runs in which two PROCESSES shared one GPU -- the same poses
    with scores inside the 1e-4 parity band and at most a handful of rows displaced.  (On one GPU the other rank's plugin
    convolution runs beside this rank's search, and that co-residency now and then changes low bits of a rotation's scores on
    this hardware -- EXPERIMENTS.md R5; a single process never schedules the two together.  The comparison is tolerant of that
    effect for the shared-GPU layout only; one process per GPU -- the production layout, and the RCCL test below -- is
    byte-identical.)"""
    if a == b:
        return True
    ra = [[float(v) for v in l.split(b"\t")] for l in a.strip().split(b"\n")]
    rb = [[float(v) for v in l.split(b"\t")] for l in b.strip().split(b"\n")]
    if len(ra) != n or len(rb) != n:
        return False
    scale = max(abs(r[12]) for r in ra)
    if max(abs(x[12] - y[12]) for x, y in zip(ra, rb)) > 1e-4 * scale:
        return False
    key = lambda r: tuple(int(round(v * 1e4)) for v in r[:12])
    same_pose = sum(1 for x, y in zip(ra, rb) if key(x) == key(y))
    return same_pose >= n - 40 and len({key(r) for r in ra} ^ {key(r) for r in rb}) <= 8


def _sweep(root, log_name, nproc, extra_args=(), extra_env=None, port=29671, rccl=False):
    """deeplocalproteindocking_amd/local_test.py (the rank-aware driver) in fresh processes -> (report, stdout)."""
    env = dict(os.environ)
    env.update({"DLPD_DATA_DIR": os.path.join(root, "data"), "DLPD_MODELS_DIR": os.path.join(root, "models"),
                "DLPD_LOG_DIR": os.path.join(root, log_name), "DLPD_ALLOW_GENERATED_ROTATIONS": "1",
                "PYTHONDONTWRITEBYTECODE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "DLPD_LAUNCH_BATCH"):
        env.pop(k, None)
    env.update(extra_env or {})
    os.makedirs(os.path.join(env["DLPD_LOG_DIR"], "LocalDebugSE3"), exist_ok=True)
    script = os.path.join(ROOT, "deeplocalproteindocking_amd", "local_test.py")
    launch = [sys.executable, script] if nproc == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
         "127.0.0.1", "--master-port", str(port), script] + ([] if rccl else ["-backend", "gloo", "-same_device", "1"])
    cmd = launch + ["-angle_inc", "20", "-seed", "7", "-init_weights", "1", "-report", "1", "-threshold_clash", "40.0",
                    "-start", "0", "-end", "3"] + list(extra_args)
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rep = [l for l in out.stdout.splitlines() if l.startswith("SWEEP ")]
    assert len(rep) == 1
    return json.loads(rep[0][len("SWEEP "):]), out.stdout


@pytest.mark.gpu
def test_rank_aware_sweep_two_ranks_on_this_gpu_equals_one_rank(tmp_path):
    """BASELINE config 5's driver on hardware, as far as a one-GPU box goes: local_test.py's target loop over a
    three-target synthetic benchmark by TWO ranks (both on this GPU, gloo transport), rotations of every target sharded,
    rank 0 writing, the next target prepared by a second host thread -- against the single-rank sweep
    without preparation ahead: the same .dat bytes; then the resume rule across ranks."""
    import __graft_entry__ as entry
    entry.build()
    root = str(tmp_path)
    make_benchmark(root, targets=(("1SYN", 150, 90, 21), ("2SYN", 120, 100, 41), ("3SYN", 100, 60, 61)))
    two, stdout2 = _sweep(root, "logW2", 2, ["-rewrite", "1"])
    one, _ = _sweep(root, "logW1", 1, ["-rewrite", "1", "-prefetch", "0"])
    assert two["world_size"] == 2 and two["processed"] == 3 and two["prepared_ahead"] and not one["prepared_ahead"]
    assert [t["prepared_ahead"] for t in two["targets"]] == [False, True, True]
    assert all(t["path"] == "fused" and t["poses"] == 2000 and t["rotations"] == 1854 for t in two["targets"])
    assert [t["randR"] for t in two["targets"]] == [t["randR"] for t in one["targets"]]      # -seed 7 on both
    for name in ("1SYN", "2SYN", "3SYN"):
        a = open(os.path.join(two["test_dir"], name + ".dat"), "rb").read()
        b = open(os.path.join(one["test_dir"], name + ".dat"), "rb").read()
        assert len(a.splitlines()) == 2000 and _same_dat_up_to_the_parity_band(a, b), name
    print("sweep: %.2f targets/s on two ranks sharing the GPU (prepared ahead), %.2f on one rank (not prepared ahead); "
          "waited for a prepared target: %s s" % (two["targets_per_s"], one["targets_per_s"],
                                                  [round(t["waited_for_preparation_s"], 3) for t in two["targets"]]))
    # resume (local_test.py:65 with -rewrite 0): an interrupted 2SYN is redone, the others are skipped by both ranks
    dat = os.path.join(two["test_dir"], "2SYN.dat")
    whole = open(dat, "rb").read()
    with open(dat, "wb") as f:
        f.write(whole.splitlines(True)[0])
    again, stdout = _sweep(root, "logW2", 2, ["-rewrite", "0"], port=29673)
    assert again["processed"] == 1 and again["skipped"] == 2 and [t["target"] for t in again["targets"]] == ["2SYN"]
    assert "Skipping 1SYN" in stdout and "Processing 2SYN" in stdout and "Skipping 3SYN" in stdout
    assert _same_dat_up_to_the_parity_band(open(dat, "rb").read(), whole)
    # one rank, targets prepared ahead: the same files again (the second engine and the preparing host thread in one process)
    pre, _ = _sweep(root, "logW1p", 1, ["-rewrite", "1", "-prefetch", "1"])
    for name in ("1SYN", "2SYN", "3SYN"):
        assert open(os.path.join(pre["test_dir"], name + ".dat"), "rb").read() == \
            open(os.path.join(one["test_dir"], name + ".dat"), "rb").read()
    print("sweep, one rank: %.2f targets/s prepared ahead vs %.2f" % (pre["targets_per_s"], one["targets_per_s"]))


@pytest.mark.gpu
def test_rank_aware_sweep_on_a_one_rank_rccl_group_equals_the_ungrouped_sweep(tmp_path):
    """What a one-GPU box can run of the RCCL side of the sweep: `-force_group 1 -backend nccl` makes a communicator of one
    rank, and the plan / decision broadcasts, the random-rotation broadcast and the per-target all-gather all go through
    RCCL on device tensors (the code path of the 8-GPU sweep, with one participant).  Same .dat bytes as without a group."""
    import __graft_entry__ as entry
    entry.build()
    root = str(tmp_path)
    make_benchmark(root, targets=(("1SYN", 150, 90, 21), ("2SYN", 120, 100, 41)))
    grouped, _ = _sweep(root, "logR1", 1, ["-rewrite", "1", "-end", "2", "-force_group", "1", "-backend", "nccl"],
                        extra_env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29679"})
    plain, _ = _sweep(root, "logP1", 1, ["-rewrite", "1", "-end", "2"])
    assert grouped["collective_backend"] == "nccl" and plain["collective_backend"] is None
    assert grouped["world_size"] == 1 and grouped["processed"] == 2
    assert [t["randR"] for t in grouped["targets"]] == [t["randR"] for t in plain["targets"]]
    for name in ("1SYN", "2SYN"):
        assert open(os.path.join(grouped["test_dir"], name + ".dat"), "rb").read() == \
            open(os.path.join(plain["test_dir"], name + ".dat"), "rb").read()


@pytest.mark.gpu
def test_rank_aware_sweep_over_rccl_equals_one_rank(tmp_path):
    """The same sweep with one GPU per rank and the `nccl` (= RCCL) backend -- the decision broadcast, the random-rotation
    broadcast and the per-target all-gather on device tensors.  Needs two GPUs; skipped on the one-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL)")
    import __graft_entry__ as entry
    entry.build()
    root = str(tmp_path)
    make_benchmark(root, targets=(("1SYN", 150, 90, 21), ("2SYN", 120, 100, 41)))
    two, _ = _sweep(root, "logN2", 2, ["-rewrite", "1", "-end", "2"], port=29677, rccl=True)
    one, _ = _sweep(root, "logN1", 1, ["-rewrite", "1", "-end", "2", "-prefetch", "0"])
    assert two["world_size"] == 2 and two["processed"] == 2
    for name in ("1SYN", "2SYN"):
        assert open(os.path.join(two["test_dir"], name + ".dat"), "rb").read() == \
            open(os.path.join(one["test_dir"], name + ".dat"), "rb").read()
