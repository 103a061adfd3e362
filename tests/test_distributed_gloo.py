"""world_size-2 rotation sharding on CPU (gloo): each rank scores its interleaved shard with the
emulated kernels, one all-gather + deterministic merge must reproduce the single-process list."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class _Model:
    threshold_clash = 4000.0
    clip = 5.0

    def __init__(self, filt):
        self.filter = filt

    def eval(self):
        return self


def _inputs():
    from deeplocalproteindocking_amd.Models import SimpleFilter
    from oracle import docking_oracle as orc
    g = torch.Generator().manual_seed(21)
    L, C = 32, 4
    rec = torch.randn(1, C, L, L, L, generator=g) * 0.1
    lig = torch.randn(1, C, L, L, L, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    torch.manual_seed(22)
    filt = SimpleFilter([C])
    R = orc.euler_to_matrix([0.3, -1.0, 2.0, 0.7, -2.2], [1.1, 0.4, 2.2, 1.9, 0.2], [-2.0, 2.5, 0.1, 1.0, -0.4])
    return L, rec, lig, recf, ligf, filt, R


def _run(rank, world, port, out, nrot=5):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from emu_lib import emu_lib
    from deeplocalproteindocking_amd.Docker import Docker
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    L, rec, lig, recf, ligf, filt, R = _inputs()
    R = R[:nrot]
    dk = Docker(_Model(filt), angle_inc=20, box_size=L, resolution=1.25, max_conf=30, rotations=R, device="cpu",
                rank=rank, world_size=world, lib=emu_lib())
    top = dk.dock_volumes([rec], [lig], recf, ligf, batch_size=2, write=False)
    out[rank] = top
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_sharded_search_equals_single_process():
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out2 = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_run, args=(r, 2, port, out2)) for r in range(2)]
    for p in procs:
        p.start()
    single = {}
    _run(0, 1, 0, single)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert list(out2[0]) == list(out2[1])                    # every rank ends with the same list
    assert list(out2[0]) == single[0]
    rots = {t[0] for t in single[0]}
    assert len(rots) > 1                                     # entries from both shards


def test_rank_with_empty_shard_still_joins_the_gather():
    """Fewer rotations than ranks: rank 1 scores nothing, contributes an empty list to the all-gather,
    and still ends with the global list."""
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out2 = mgr.dict()
    port = 31600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_run, args=(r, 2, port, out2, 1)) for r in range(2)]
    for p in procs:
        p.start()
    single = {}
    _run(0, 1, 0, single, 1)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert list(out2[0]) == list(out2[1]) == single[0]
    assert len(single[0]) == 30 and {t[0] for t in single[0]} == {0}


def _run_bench_gather(rank, world, port, out):
    """bench.py's `gather_check` / `strong` leg (bench.sharded_search + bench.list_sha256) on the emulated kernels."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    import bench
    from emu_lib import emu_lib
    from deeplocalproteindocking_amd.engine import DockingEngine
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    L, rec, lig, recf, ligf, filt, R = _inputs()
    K = 25
    eng = DockingEngine(L, 4, *filt.parameters_tuple(), clip=5.0, threshold_clash=4000.0, max_conf=K, batch=2, device="cpu",
                        lib=emu_lib())
    eng.set_receptor(rec[0], recf)
    eng.set_ligand(lig[0], ligf)
    R_all = torch.from_numpy(R)
    ids = np.array([4, 0, 3, 1, 2])[:5]                       # a visiting order that is not the index order
    ent, dt = bench.sharded_search(eng, R_all, ids, rank, world, K, dist if world > 1 else None, torch.device("cpu"))
    out[rank] = (bench.list_sha256(ent), len(ent[0]), sorted(set(int(r) for r in ent[0])))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_bench_gather_check_hash_does_not_depend_on_the_world_size():
    """What the driver's N = 1 and N = 8 bench lines are compared by: `gather_check.list_sha256` -- the same subset of
    rotations sharded r::W, one all-gather, the deterministic merge -- must be one value for every W (here 1, 2, 3; three
    ranks share five rotations 2 / 2 / 1)."""
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    results = {}
    for world, base in ((2, 33100), (3, 34200)):
        out = mgr.dict()
        port = base + (os.getpid() % 900)
        procs = [ctx.Process(target=_run_bench_gather, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
            assert p.exitcode == 0
        assert len({out[r][0] for r in range(world)}) == 1    # every rank holds the same merged list
        results[world] = out[0]
    single = {}
    _run_bench_gather(0, 1, 0, single)
    assert single[0][1] == 25 and len(single[0][2]) > 1
    assert results[2] == single[0] and results[3] == single[0]


def test_bench_spawns_its_own_ranks_when_started_as_plain_python():
    """The driver runs `python bench.py --gpus N` for its scaling sweep: the process must start the N ranks itself
    (child torch.distributed.run, 127.0.0.1 rendezvous, free port), relay rank 0's single JSON line and its exit
    code.  --dry_run keeps the GPU out of it (gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry_run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "n_gpus": 2, "steps": 3, "warmup": 1}
    # a rank that fails makes the launcher's exit code non-zero
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry_run"],
                         env=dict(env, WORLD_SIZE="3"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


# ---------------------------------------------------------------------------------------------------------------
# world size 8 (the node the scaling sweep runs on): uneven shards, empty shards, list sizes of the real search
# ---------------------------------------------------------------------------------------------------------------
class _CalledModel:
    """A docking model with a forward of its own (not the reference's MLP): Docker calls it (path "call") between the
    emulated rotation kernel and the emulated device top-K -- cheap enough on the emulator for dozens of rotations."""
    threshold_clash = 1e9

    def eval(self):
        return self

    def __call__(self, rec, lig):
        r, l = rec[0], lig[0]
        N = 2 * r.shape[-1]
        f = torch.fft.irfftn(torch.fft.rfftn(r, s=(N, N, N)) * torch.conj(torch.fft.rfftn(l, s=(N, N, N))), s=(N, N, N))
        return -f.abs().sum(dim=1)


def _run8(rank, world, port, out, nrot):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from emu_lib import emu_lib
    from oracle import docking_oracle as orc
    from deeplocalproteindocking_amd.Docker import Docker
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    L, C, K = 8, 2, 25
    g = torch.Generator().manual_seed(3)
    rec, lig = torch.randn(1, C, L, L, L, generator=g), torch.randn(1, C, L, L, L, generator=g)
    ang = np.random.RandomState(5).uniform(-np.pi, np.pi, size=(nrot, 3))
    R = orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])
    dk = Docker(_CalledModel(), angle_inc=20, box_size=L, resolution=1.25, max_conf=K, rotations=R, device="cpu",
                rank=rank, world_size=world, lib=emu_lib())
    top = dk.dock_volumes([rec], [lig], None, None, batch_size=4, write=False)
    assert dk.path == "call"
    out[rank] = (top, len(dk.shard(nrot)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(target, world, port, *args):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=target, args=(r, world, port, out) + args) for r in range(world)]
    for p in procs:
        p.start()
    return out, procs


def _join(procs):
    for p in procs:
        p.join(900)
        assert p.exitcode == 0


def test_eight_rank_uneven_shards_equal_single_process():
    """54 rotations over 8 ranks: six ranks score 7, two score 6 (the remainder pattern of 1,854 % 8 = 6: six ranks get
    one more); every rank must end with the single-process list."""
    out, procs = _spawn(_run8, 8, 33100 + (os.getpid() % 2000), 54)
    single = {}
    _run8(0, 1, 0, single, 54)
    _join(procs)
    assert sorted(out[r][1] for r in range(8)) == [6, 6, 7, 7, 7, 7, 7, 7]
    for r in range(8):
        assert list(out[r][0]) == single[0][0]
    assert len({t[0] % 8 for t in single[0][0]}) > 2            # entries from several shards


def test_eight_ranks_with_three_empty_shards():
    """5 rotations over 8 ranks: ranks 5, 6, 7 score nothing and still join the one all-gather."""
    out, procs = _spawn(_run8, 8, 35200 + (os.getpid() % 2000), 5)
    single = {}
    _run8(0, 1, 0, single, 5)
    _join(procs)
    assert [out[r][1] for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
    for r in range(8):
        assert list(out[r][0]) == single[0][0]


def _gather_only(rank, world, port, out, nrot, K):
    """The collective + merge alone at the real search's sizes: every rank keeps the top K of its interleaved shard of a
    synthetic table of picks (heavy ties, signed zeros), one all-gather, the merged list must be the global sort."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from deeplocalproteindocking_amd.Docker.Docker import all_gather_top_entries
    from deeplocalproteindocking_amd.engine import DeviceTopList
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rot, idx, score, pick = _synthetic_picks(nrot)
    mine = (rot % world) == rank
    local = DeviceTopList.merge_entries([(rot[mine], idx[mine], score[mine], pick[mine])], K)
    merged = all_gather_top_entries(local, K, world, None, "cpu")
    out[rank] = tuple(np.asarray(a).tolist() for a in merged)
    dist.barrier()
    dist.destroy_process_group()


def _synthetic_picks(nrot, per_rot=40):
    rs = np.random.RandomState(11)
    rot = np.repeat(np.arange(nrot), per_rot)
    pick = np.tile(np.arange(per_rot), nrot)
    idx = rs.randint(0, 128 ** 3, size=rot.size)
    score = -np.round(rs.rand(rot.size) * 50).astype(np.float32) / 8      # 400 distinct values: ties everywhere
    score[rs.rand(rot.size) < 0.01] = np.float32(-0.0)
    return rot, idx, score, pick


def test_eight_rank_gather_of_full_size_lists_equals_the_global_sort():
    """1,854 rotations (the 20-degree set: 1,854 % 8 = 6), K = 2000, 40 picks per rotation with heavily tied scores: the
    list every rank holds after the single all-gather equals the sort of ALL picks by (score, rotation, pick) -- the
    reference's stable insertion order (Docker.py:100-105) -- entry for entry."""
    nrot, K = 1854, 2000
    out, procs = _spawn(_gather_only, 8, 37300 + (os.getpid() % 2000), nrot, K)
    _join(procs)
    rot, idx, score, pick = _synthetic_picks(nrot)
    order = np.lexsort((pick, rot, score + np.float32(0.0)))[:K]
    want = (rot[order].tolist(), idx[order].tolist(), score[order].tolist(), pick[order].tolist())
    for r in range(8):
        assert out[r][0] == want[0] and out[r][1] == want[1] and out[r][3] == want[3]
        assert np.array_equal(np.asarray(out[r][2], dtype=np.float32), np.asarray(want[2], dtype=np.float32))


def test_bench_eight_rank_launch_plumbing():
    """`python bench.py --gpus 8 --dry_run`: the launcher the driver's scaling sweep uses, at the node's width."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
                        "--dry_run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_run": True, "n_gpus": 8, "steps": 4, "warmup": 1}


# ---------------------------------------------------------------------------------------------------------------
# the reference's production call (local_test.py:55: randomize_rot=True) on a rotation-sharded search
# ---------------------------------------------------------------------------------------------------------------
def _random_rot_inputs(tmp):
    """The receptor / ligand PDB files of the random-rotation cases, written ONCE by the test process: a rank only ever
    reads them (two ranks writing the same path raced: one truncated the file the other was parsing)."""
    from test_atoms import write_fake_pdb
    frec, flig = os.path.join(str(tmp), "p5.pdb"), os.path.join(str(tmp), "p6.pdb")
    write_fake_pdb(frec, 14, 5)
    write_fake_pdb(flig, 9, 6)
    return frec, flig


def _run_random_rot(rank, world, port, out, tmp, frec, flig, method="dockSE3", late_group=False, randR=None):
    """One rank of Docker(randomize_rot=True).dockSE3 / dockE3 from PDB files on the emulated kernels.  Every rank has
    its OWN RNG state (global seed and rotation_seed differ per rank), as any rank-dependent seeding would give it; rank
    0 alone opens the .dat.  late_group: the process group is created AFTER the Docker (the matrix is then shared at the
    start of dock*).  randR: force this matrix (the single-process comparison run)."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from emu_lib import emu_lib
    from oracle import docking_oracle as orc
    from test_atoms import _assert_scores_are_informative, _tiny_model
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Docker.Docker import random_rotation
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    import pathlib
    tmp = pathlib.Path(tmp)
    init = lambda: dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    if world > 1 and not late_group:
        init()
    L, res, K = 32, 1.25, 40
    model = _tiny_model()
    # five neighbouring rotations: their best scores are close, so the merged list draws on both shards
    R = orc.euler_to_matrix(0.3 + 0.03 * np.arange(5), 1.1 - 0.02 * np.arange(5), -2.0 + 0.025 * np.arange(5))
    torch.manual_seed(1000 + 17 * rank)                            # rank-dependent global RNG state
    lib = emu_lib()
    dk = Docker(model, box_size=L, resolution=res, max_conf=K, rotations=R, device="cpu", lib=lib,
                coords_backend=CoordsBackend(lib=lib), randomize_rot=True, rotation_seed=500 + rank,
                rank=rank, world_size=world)
    own = random_rotation(seed=500 + rank)
    if world > 1 and late_group:
        assert torch.equal(dk.randR, own)                          # nothing to share it through yet
        init()
    if randR is not None:
        dk.randR = torch.as_tensor(randR, dtype=torch.float64).reshape(1, 3, 3)
    log = str(tmp / ("%s_w%d.dat" % (method, world)))
    if rank == 0:
        assert dk.new_log(log)
    with torch.no_grad():
        getattr(dk, method)(frec, flig, batch_size=2)
    dk.cleanup()
    _assert_scores_are_informative(dk.top_list)
    out[rank] = {"top": list(dk.top_list), "randR": dk.randR.reshape(9).tolist(), "own": own.reshape(9).tolist(),
                 "dat": open(log, "rb").read() if rank == 0 else None}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _random_rot_case(tmp_path, method, late_group, base_port, counter_check=True):
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(), d1.mkdir()
    frec, flig = _random_rot_inputs(tmp_path)
    out, procs = _spawn(_run_random_rot, 2, base_port + (os.getpid() % 1500), str(d2), frec, flig, method, late_group)
    _join(procs)
    r0, r1 = out[0], out[1]
    assert r0["own"] != r1["own"]                                  # the ranks drew different matrices ...
    assert r0["randR"] == r0["own"] and r1["randR"] == r0["own"]   # ... and both ended with rank 0's
    assert r0["top"] == r1["top"] and len(r0["top"]) == 40
    assert len({t[0] % 2 for t in r0["top"]}) == 2                 # entries from both shards
    single = {}
    _run_random_rot(0, 1, 0, single, str(d1), frec, flig, method, False, r0["randR"])
    assert single[0]["top"] == r0["top"]
    assert single[0]["dat"] == r0["dat"] and len(r0["dat"].splitlines()) == 40
    if not counter_check:
        return
    # a rank-1-only matrix would have given another list: the receptor really is rotated by randR
    other = {}
    d3 = tmp_path / "w1b"
    d3.mkdir()
    _run_random_rot(0, 1, 0, other, str(d3), frec, flig, method, False, r1["own"])
    assert other[0]["top"] != r0["top"]


def test_two_rank_dockSE3_with_random_receptor_rotation_equals_single_process(tmp_path):
    """local_test.py:55 constructs Docker(randomize_rot=True); Docker.py:44,193-197 rotates the receptor's atoms by that
    matrix.  Rotation-sharded, every rank must use rank 0's: the merged list and the .dat bytes equal a single-process
    run given the same matrix, although each rank's generator would have drawn another."""
    _random_rot_case(tmp_path, "dockSE3", False, 38400)


def test_two_rank_dockE3_with_random_receptor_rotation_and_a_late_process_group(tmp_path):
    """The same for dockE3 (Docker.py:135-182), with the process group created after the Docker: the matrix is then
    shared at the start of the dock call."""
    _random_rot_case(tmp_path, "dockE3", True, 40100, counter_check=False)


def test_sharded_docker_without_a_process_group_refuses_to_dock_with_a_private_rotation(tmp_path):
    from emu_lib import emu_lib
    from test_atoms import _tiny_model, _typed
    from deeplocalproteindocking_amd.Docker import Docker
    import pytest
    frec, _, _, _ = _typed(tmp_path, 6, seed=5)
    dk = Docker(_tiny_model(), box_size=32, max_conf=5, rotations=np.eye(3)[None],
                device="cpu", lib=emu_lib(), randomize_rot=True, rank=1, world_size=2)
    with pytest.raises(Exception, match="process group"):
        dk.dockSE3(frec, frec, batch_size=2)


def test_random_rotation_is_fresh_per_docker_and_reproducible_with_a_seed():
    from deeplocalproteindocking_amd.Docker.Docker import random_rotation
    a, b = random_rotation(), random_rotation()
    assert not torch.equal(a, b)
    assert torch.equal(random_rotation(seed=4), random_rotation(seed=4))
    state = torch.get_rng_state()
    random_rotation()
    assert torch.equal(torch.get_rng_state(), state)               # the process-global stream is left alone
    for r in (a, b):
        m = r[0].numpy()
        assert np.abs(m @ m.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(m) - 1) < 1e-12


# ---------------------------------------------------------------------------------------------------------------
# the rank-aware benchmark sweep (local_test.py:57-75 on W ranks; BASELINE config 5)
# ---------------------------------------------------------------------------------------------------------------
def _sweep_targets(root, write=False):
    """three synthetic targets (receptor / ligand PDB files) in ``root``; written by the test, only named by the ranks"""
    import pathlib
    from test_atoms import _typed
    root = pathlib.Path(root)
    out = []
    for k, (nrec, nlig) in enumerate(((14, 9), (11, 8), (9, 12))):
        if write:
            root.mkdir(exist_ok=True)
            _typed(root, nrec, seed=30 + 2 * k), _typed(root, nlig, seed=31 + 2 * k)
        out.append(("T%d" % k, str(root / ("p%d.pdb" % (30 + 2 * k))), str(root / ("p%d.pdb" % (31 + 2 * k)))))
    return out


def _run_sweep(rank, world, port, out, root, test_dir, rewrite, prefetch, group="SE3", ntargets=3, group_of_one=False,
               plan_misses=None):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from emu_lib import emu_lib
    from oracle import docking_oracle as orc
    from test_atoms import _tiny_model
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    from deeplocalproteindocking_amd import local_test
    if world > 1 or group_of_one:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    targets = _sweep_targets(root)[:ntargets]
    R = orc.euler_to_matrix(0.3 + 0.03 * np.arange(5), 1.1 - 0.02 * np.arange(5), -2.0 + 0.025 * np.arange(5))
    torch.manual_seed(2000 + rank)
    lib = emu_lib()
    dk = Docker(_tiny_model(), box_size=32, resolution=1.25, max_conf=40, rotations=R, device="cpu", lib=lib,
                coords_backend=CoordsBackend(lib=lib), randomize_rot=True, rotation_seed=900 + rank, rank=rank, world_size=world,
                collectives_with_one_rank=group_of_one)
    if plan_misses:                                  # the plan takes this target for finished; at its turn it is not
        honest, asked = dk.log_is_complete, []

        def first_answer_is_wrong(path):             # (the plan asks first; new_log asks again when the target's turn comes)
            if path.endswith(plan_misses + ".dat") and not asked:
                asked.append(path)
                return True
            return honest(path)
        dk.log_is_complete = first_answer_is_wrong
    said = []
    rep = local_test.sweep(dk, targets, test_dir, group=group, rewrite=rewrite, batch_size=2, prefetch=prefetch,
                           say=lambda *a: said.append(" ".join(str(x) for x in a)))
    assert dk.log is None                                          # closed by the sweep
    out[rank] = {"rep": rep, "said": said, "randR": dk.randR.reshape(9).tolist()}
    if world > 1 or group_of_one:
        dist.barrier()
        dist.destroy_process_group()


def _dat_files(test_dir):
    return {f: open(os.path.join(test_dir, f), "rb").read() for f in sorted(os.listdir(test_dir))}


def test_two_rank_benchmark_sweep_equals_the_single_process_sweep(tmp_path):
    """local_test.py's target loop on two ranks: every target's rotations sharded, rank 0 alone writes, the next
    target prepared on a second host thread while the current one is searched.  The .dat files must equal the
    single-process sweep's byte for byte; with -rewrite 0 finished targets are skipped BY BOTH RANKS (rank 0's decision
    is broadcast) and an unfinished one is redone."""
    root, d2, d1 = str(tmp_path / "pdb"), str(tmp_path / "w2"), str(tmp_path / "w1")
    os.makedirs(d2), os.makedirs(d1)
    _sweep_targets(root, write=True)
    out, procs = _spawn(_run_sweep, 2, 41800 + (os.getpid() % 1500), root, d2, True, True)
    _join(procs)
    r0, r1 = out[0]["rep"], out[1]["rep"]
    assert out[0]["randR"] == out[1]["randR"]
    assert r0["processed"] == r1["processed"] == 3 and r0["skipped"] == 0 and r0["world_size"] == 2
    assert r0["targets_per_s"] > 0 and r0["prepared_ahead"]
    assert [t["target"] for t in r0["targets"]] == ["T0", "T1", "T2"]
    assert [t["prepared_ahead"] for t in r0["targets"]] == [False, True, True]
    assert all(t["poses"] == 40 and t["path"] == "fused" for t in r0["targets"])
    assert out[0]["said"] == ["Processing T0", "Processing T1", "Processing T2"] == out[1]["said"]
    # single process, no preparation ahead, the same random rotation (rotation_seed of rank 0)
    single = {}
    _run_sweep(0, 1, 0, single, root, d1, True, False)
    assert single[0]["randR"] == out[0]["randR"] and not single[0]["rep"]["prepared_ahead"]
    files2, files1 = _dat_files(d2), _dat_files(d1)
    assert sorted(files2) == ["T0.dat", "T1.dat", "T2.dat"] and files2 == files1
    assert all(len(b.splitlines()) == 40 for b in files2.values()) and len(set(files2.values())) == 3
    # resume: T1 is cut down to one line (an interrupted run), T0 and T2 are complete
    with open(os.path.join(d2, "T1.dat"), "wb") as f:
        f.write(files2["T1.dat"].splitlines(True)[0])
    out, procs = _spawn(_run_sweep, 2, 43400 + (os.getpid() % 1500), root, d2, False, True)
    _join(procs)
    for r in (0, 1):
        assert out[r]["said"] == ["Skipping T0", "Processing T1", "Skipping T2"]
        assert out[r]["rep"]["processed"] == 1 and out[r]["rep"]["skipped"] == 2
    assert _dat_files(d2) == files1


def test_benchmark_sweep_E3_prepared_ahead_equals_unprepared(tmp_path):
    """The same loop through dockE3, single rank: targets prepared ahead (receptor side in the second engine) give the
    files of the plain loop."""
    root, da, db = str(tmp_path / "pdb"), str(tmp_path / "a"), str(tmp_path / "b")
    os.makedirs(da), os.makedirs(db)
    _sweep_targets(root, write=True)
    a, b = {}, {}
    _run_sweep(0, 1, 0, a, root, da, True, True, "E3", 2)
    _run_sweep(0, 1, 0, b, root, db, True, False, "E3", 2)
    assert [t["prepared_ahead"] for t in a[0]["rep"]["targets"]] == [False, True]
    assert _dat_files(da) == _dat_files(db) and len(_dat_files(da)) == 2


def test_benchmark_sweep_on_a_process_group_of_one_rank_equals_the_ungrouped_sweep(tmp_path):
    """`collectives_with_one_rank` (local_test.py -force_group 1: how a one-GPU box takes the sweep through RCCL): the plan
    and decision broadcasts, the random-rotation broadcast and the all-gather run for a group of one and change nothing."""
    root, da, db = str(tmp_path / "pdb"), str(tmp_path / "a"), str(tmp_path / "b")
    os.makedirs(da), os.makedirs(db)
    _sweep_targets(root, write=True)
    a, b = {}, {}
    _run_sweep(0, 1, 29741, a, root, da, True, True, "SE3", 2, True)
    _run_sweep(0, 1, 0, b, root, db, True, True, "SE3", 2)
    assert a[0]["rep"]["collective_backend"] == "gloo" and b[0]["rep"]["collective_backend"] is None
    assert a[0]["randR"] == b[0]["randR"]
    assert _dat_files(da) == _dat_files(db) and len(_dat_files(da)) == 2


def test_benchmark_sweep_target_the_plan_did_not_expect_keeps_clear_of_the_prepared_engine(tmp_path):
    """The plan (read before the sweep) takes T1 for finished, so T2 is prepared ahead -- into the second engine -- while T0 is
    searched; at T1's turn its log turns out NOT to be finished (someone truncated it): it is prepared on the spot and must
    take the engine T2's preparation does not hold.  Same files as the plain sweep."""
    root, da, db = str(tmp_path / "pdb"), str(tmp_path / "a"), str(tmp_path / "b")
    os.makedirs(da), os.makedirs(db)
    _sweep_targets(root, write=True)
    a, b = {}, {}
    _run_sweep(0, 1, 0, a, root, da, False, True, "SE3", 3, False, "T1")
    _run_sweep(0, 1, 0, b, root, db, True, False, "SE3", 3)
    assert [(t["target"], t["prepared_ahead"]) for t in a[0]["rep"]["targets"]] == [("T0", False), ("T1", False), ("T2", True)]
    assert _dat_files(da) == _dat_files(db) and len(_dat_files(da)) == 3


def test_no_rank_function_writes_an_input_file():
    """Source check: the functions that run INSIDE a spawned rank (every `_spawn` target in this file and what
    test_replay_local_test.py starts) never create a PDB file -- inputs are written once by the test process.  Two ranks
    writing the same path is a race (one truncates the file the other parses), and these tests are the multi-rank
    evidence of SURVEY 8(e)."""
    import ast
    import inspect
    src = open(__file__).read()
    tree = ast.parse(src)
    funcs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    targets = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "_spawn" and isinstance(n.args[0], ast.Name):
            targets.add(n.args[0].id)
        if isinstance(n, ast.Call) and getattr(n.func, "attr", "") == "Process":
            targets |= {k.value.id for k in n.keywords if k.arg == "target" and isinstance(k.value, ast.Name)}
    targets.discard("target")                                       # _spawn's own parameter
    assert {"_run", "_run8", "_gather_only", "_run_random_rot", "_run_sweep"} <= targets
    writers = ("write_fake_pdb", "write_protein_like_pdb", "_typed", "_random_rot_inputs")
    for name in sorted(targets):
        body = ast.get_source_segment(src, funcs[name])
        calls = {getattr(c.func, "id", getattr(c.func, "attr", "")) for c in ast.walk(funcs[name]) if isinstance(c, ast.Call)}
        assert not calls & set(writers), (name, calls & set(writers))
        if "_sweep_targets(" in body:                                # naming the files is fine, writing them is not
            assert "write=True" not in body
    assert inspect.signature(_sweep_targets).parameters["write"].default is False
