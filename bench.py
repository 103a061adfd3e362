#!/usr/bin/env python3
"""Benchmark of the rotation x translation correlation search (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment this process only SPAWNS the N ranks
(``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` as a
child process, before anything here has touched a GPU), passes their output through and exits with
their code; started by torch.distributed.run itself (RANK / WORLD_SIZE set) it is one rank.

A "step" is one pass of the hot path over one batch of ``--batch`` rotations of a synthetic pair:
trilinear rotation + z R2C (K1), per-slab 2-D FFT / conj-multiply / 2-D inverse (K2), z C2R + clip +
filter MLP + clash mask (K3), per-rotation top-K select and the running global merge.  Inputs are
resident in HBM before the timed region.  The K timed steps are batches SPREAD EVENLY over the rank's
whole visiting sequence of the rotation set (all four slab-orientation x gather-layout groups, in the
engine's order), because K1's gather cost depends on the rotation: ``value`` is the sustained
whole-set rate, the cheaper head of the set is reported beside it (``head_of_set``).  With N ranks
every rank scores its own interleaved shard (weak scaling: per-GPU work fixed) and the timed region
ends with the single all-gather of the per-rank top lists and the deterministic merge.

Workloads (``--workload``): ``config2`` (default; BASELINE config 2: 48 ch x 64^3, the configuration
the metric is quoted on), ``real`` (the reference model's shapes [16 @ 80^3, 32 @ 40^3] -> 160^3,
configs 4/5), ``c48l80`` (config 5's literal 48 ch x 80^3), ``config1`` (4 ch x 32^3).

Prints ONE JSON line (rank 0): the contract keys plus "roofline" (dominant kernel: algorithmic bytes
per launch / its average launch duration from HIP events on the launch stream inside the timed
region), "rooflines" (the same for every stage), "step" (whole step against the bytes the pipeline
really moves, SURVEY 8(d)'s compulsory floor and its stage-boundary model), "head_of_set",
"real_shapes" (a short measurement of the N = 160 pipeline, default workload only) and
"cpu_baseline" (the CPU oracle = restated reference path on this box's host cores, bounded sample,
N = 1 only, with the parity of the GPU scores on the same rotations).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# SURVEY.md section 8(d), MB per rotation: stage-boundary model and compulsory floor
SURVEY_MB_PER_ROT = {"config2": (3080.3, 468.9), "config1": (41.6, 6.1), "c48l80": (6003.7, 913.2),
                     "real": (3054.6, 392.2), "real_protein": (3054.6, 392.2)}
WORKLOADS = {   # name: (C fine, L, C coarse, rotation-set angle, description)
    "config2": (48, 64, 0, 6, "BASELINE config 2: single synthetic 48-channel 64^3 pair"),
    "config1": (4, 32, 0, 20, "BASELINE config 1: single synthetic 4-channel 32^3 pair"),
    "c48l80": (48, 80, 0, 6, "BASELINE config 5 shape: synthetic 48-channel 80^3 pair (grid 160^3)"),
    "real": (16, 80, 32, 6, "reference model shapes (configs 4/5): synthetic [16 @ 80^3, 32 @ 40^3] pair (grid 160^3)"),
    # the same shapes with PROTEIN-SHAPED contents: the E3 plugin's representation of a synthetic 160 / 110-residue pair
    # (what Docker.dockSE3 searches, Docker.py:204-209,218) -- zero away from the protein, unlike "real"'s dense random volumes
    "real_protein": (16, 80, 32, 6, "reference model shapes with protein-shaped contents: E3MultiResRepr4x4(8) representation "
                                    "[16 @ 80^3, 32 @ 40^3] of a synthetic 160 / 110-residue pair (grid 160^3)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="rotations per step = per launch (Docker.launch_batch: 32 since round 6, 16 before)")
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--channels", type=int, default=None, help="override the workload's channel count")
    ap.add_argument("--box", type=int, default=None, help="override the workload's box size")
    ap.add_argument("--angle_inc", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None,
                    help="hidden width of the filter (default: the reference's SimpleFilter, half the channel count)")
    ap.add_argument("--max_conf", type=int, default=2000)
    ap.add_argument("--cpu_rotations", type=int, default=16,
                    help="rotations of the CPU baseline sample, spread over the four search groups (0: skip)")
    ap.add_argument("--no_real_shapes", action="store_true", help="skip the short N = 160 measurement")
    ap.add_argument("--rotation_order", default="set", choices=("set", "random", "strided8"),
                    help="diagnostic: which rotations share a launch -- set order (16 consecutive phi steps of one direction), a random "
                         "permutation, or every 8th rotation (what one of 8 ranks sees under r::8 interleaving)")
    ap.add_argument("--k1_occupancy", default="auto", choices=("auto", "on", "off"),
                    help="per-rotation occupancy maps in the channels-last K1: auto = where the ligand leaves most cells empty")
    ap.add_argument("--no_pmc", action="store_true",
                    help="roofline.traffic from the committed profiles/ file instead of two rocprofv3 --pmc child passes of this run")
    ap.add_argument("--sustained_s", type=float, default=8.0,
                    help="untimed extra: seconds of CONSECUTIVE batches from the head of the visiting sequence (0: skip)")
    ap.add_argument("--strong_s", type=float, default=12.0,
                    help="untimed extra `strong`: the COMPLETE rotation set sharded r::W over the W ranks if the estimate fits these "
                         "seconds, else the prefix of the visiting sequence that does (0: skip)")
    ap.add_argument("--gather_rotations", type=int, default=1024,
                    help="untimed extra `gather_check`: the first n rotations of the visiting sequence, sharded r::W, one "
                         "all-gather + merge; the list hash must not depend on the world size (0: skip)")
    ap.add_argument("--no_extras", action="store_true", help="skip the c48l80 and e3 extra objects of the default line")
    ap.add_argument("--natural_receptor", action="store_true",
                    help="A/B switch: K2 of boxes 80 / 40 reads the receptor spectrum in its natural layout (not the packed copy)")
    ap.add_argument("--k3_form", type=int, default=0, choices=(0, 1, 2),
                    help="kernel formulation of the fused K3 (include/dlpd.h, dlpd_zifft_filter_form): 0 = library default")
    ap.add_argument("--force_group", action="store_true",
                    help="initialise the process group and run every collective even with ONE rank (a one-rank RCCL "
                         "communicator: how a one-GPU box exercises the nccl branches of this file)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend of the multi-rank run: nccl (= RCCL over xGMI, one GPU per rank) or gloo")
    ap.add_argument("--same_device", action="store_true",
                    help="every rank on cuda:0 (needs --backend gloo): the multi-rank code path with the HIP kernels on a one-GPU box")
    ap.add_argument("--k1_form", type=int, default=0, choices=(0, 1, 2),
                    help="kernel formulation of the channels-last K1 (include/dlpd.h, dlpd_zfft_channels_last_form): 0 = library default")
    ap.add_argument("--dry_run", action="store_true",
                    help="launch plumbing only (gloo, no GPU): every rank joins the group, rank 0 prints a JSON line")
    return ap.parse_args()


def spawn_ranks(args):
    """Plain ``python bench.py --gpus N``: start the N ranks as a CHILD process tree (never exec from
    a process that may have initialised the GPU; this one has not), relay the rank-0 JSON line."""
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def synthetic_pair(C, L, seed=0):
    """SURVEY.md 8(d) synthetic inputs: N(0,1) x smooth radial envelope representation volumes,
    relu-blob forbidden volumes, SimpleFilter([C]) after manual_seed(1)."""
    import torch
    from deeplocalproteindocking_amd.Models import SimpleFilter
    rec, lig = synthetic_volumes(C, L, seed)
    recf, ligf = synthetic_forbidden(L)
    torch.manual_seed(1)
    filt = SimpleFilter([C])
    return rec, lig, recf, ligf, filt


def synthetic_volumes(C, L, seed=0):
    import torch
    g = torch.Generator().manual_seed(seed)
    ar = (torch.arange(L, dtype=torch.float32) - (L - 1) / 2.0) / (L / 2.0)
    r2 = ar[:, None, None] ** 2 + ar[None, :, None] ** 2 + ar[None, None, :] ** 2
    env = torch.exp(-1.5 * r2)
    amp = 206.0 / L ** 1.5              # per-channel correlation std ~2-4: the +-5 clip bites on the tails only
    rec = torch.randn(C, L, L, L, generator=g) * env * amp
    lig = torch.randn(C, L, L, L, generator=g) * env * amp
    return rec, lig


def synthetic_forbidden(L):
    import torch
    ar = (torch.arange(L, dtype=torch.float32) - (L - 1) / 2.0) / (L / 2.0)

    def blob(shift):
        c = torch.tensor(shift, dtype=torch.float32)
        d2 = (ar[:, None, None] - c[0]) ** 2 + (ar[None, :, None] - c[1]) ** 2 + (ar[None, None, :] - c[2]) ** 2
        return torch.relu(torch.exp(-2.0 * d2) - 0.2)
    return blob((0.1, -0.05, 0.0)), blob((-0.05, 0.1, 0.05))


def clash_threshold(recf, ligf):
    """Median of the (unrotated) clash correlation (about half of the grid masked), floored at
    1e-3 of its maximum so the mask never depends on FFT round-off around an exact zero overlap."""
    import torch
    L = recf.shape[0]
    N = 2 * L
    f = torch.fft.irfftn(torch.fft.rfftn(recf, s=(N, N, N)) * torch.conj(torch.fft.rfftn(ligf, s=(N, N, N))),
                         s=(N, N, N))
    return max(float(f.median()), 1e-3 * float(f.max()))


class StageTimer:
    def __init__(self):
        self.events = []

    def mark(self, name):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()                               # torch's current stream == the launch stream
        self.events.append((name, e))

    def summary(self):
        tot, cnt = {}, {}
        for (n0, e0), (n1, e1) in zip(self.events[:-1], self.events[1:]):
            if n1 == "begin":
                continue
            tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
            cnt[n1] = cnt.get(n1, 0) + 1
        return {k: tot[k] / cnt[k] for k in tot}


def usable_cores():
    """Host cores this process may really use: min(affinity, cgroup CPU quota).  (The GPU boxes show
    256 logical CPUs behind a 16-core quota; oversubscribing it throttles torch ~30x.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(rec, lig, recf, ligf, W, R_sample, group_of, thr, K, V_gpu=None):
    """The oracle (restated reference path incl. the reference's update_top loop) on the host, on
    rotations taken from all four search groups; parity of the GPU scores on the same rotations."""
    import torch
    from oracle import docking_oracle as orc
    L = rec[0].shape[-1]
    n = R_sample.shape[0]
    torch.set_num_threads(usable_cores())
    t0 = time.time()
    top, Vs = orc.dock_volumes([r[None] for r in rec], [l[None] for l in lig], recf[None, None], ligf[None, None],
                               R_sample, *W, thr, K, clip=5.0, faithful_topk=True, return_V=True)
    dt = time.time() - t0
    out = {"value": n * (2 * L) ** 3 / dt, "unit": "pose scores/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d rotations of the same rotation set and pair, %d from each of the four rotation classes "
                     "(source z axis closest to output x / y / z: the per-channel K1's slab orientation x gather "
                     "layout groups), %.1f s; SURVEY 8(d)'s 32 rotations would take about twice "
                     "the 10-30 s the bench contract allots to this leg" % (n, n // 4, dt)}
    if V_gpu is not None:
        # voxels whose clash mask differs (clash correlation within FFT round-off of the threshold)
        # are counted, not compared
        errs, flips, per_group = [], 0, {}
        for i in range(n):
            same = (V_gpu[i] == 0) == (Vs[i] == 0)
            flips += int((~same).sum())
            e = float((V_gpu[i] - Vs[i]).abs()[same].max() / Vs[i].abs().max())
            errs.append(e)
            per_group[group_of[i]] = max(per_group.get(group_of[i], 0.0), e)
        out["parity"] = {"max_err_rel_to_max_abs_score": max(errs), "tolerance": 1e-4,
                         "max_err_by_group": per_group, "mask_flips_at_threshold": flips, "voxels": n * (2 * L) ** 3}
    return out


GROUP_NAMES = ("plain", "quads", "transposed", "transposed+quads")


def visiting_sequence(engine_cls, R_all, ids, nb):
    """This rank's rotations in the order DockingEngine.search visits them: grouped by slab orientation x
    gather layout (per-launch choices), set order inside a group, whole batches only."""
    import numpy as np
    Rn = R_all[ids].numpy()
    gkey = engine_cls.prefers_transposed(Rn).astype(int) * 2 + engine_cls.prefers_quads(Rn).astype(int)
    parts, keys = [], []
    for k in range(4):
        sel = ids[gkey == k]
        sel = sel[:len(sel) // nb * nb]
        parts.append(sel)
        keys.append(np.full(len(sel), k))
    return np.concatenate(parts), np.concatenate(keys)


def algorithmic_bytes(C, L, C1, nb, K, has_clash=True, unfused=False, HP=24, prefilter=True):
    """Per-launch algorithmic bytes of every stage (each kernel's compulsory input + output, f32):
    DESIGN.md section 4."""
    N, NZ, CT = 2 * L, L + 1, C + (1 if has_clash else 0)
    alg = {
        "k1_rotate_zfft": CT * L ** 3 * 4 + nb * CT * NZ * L * L * 8,
        "k2_xy_corr": nb * CT * NZ * L * L * 8 + CT * NZ * N * N * 8 + nb * CT * NZ * N * N * 8,
        # candidate path (steady state): the select reads K3's short candidate lists, not V
        "topk_select": (nb * 4096 * 8 if prefilter else nb * N ** 3 * 4) + nb * K * 8,
        "topk_merge": nb * K * 8 + K * 16,
    }
    if unfused:
        alg["k3_zifft"] = nb * CT * NZ * N * N * 8 + nb * CT * N ** 3 * 4
        alg["filter"] = nb * CT * N ** 3 * 4 + nb * N ** 3 * 4 + (nb * HP * L ** 3 * 4 if C1 else 0)
    else:
        alg["k3_zifft_filter"] = nb * CT * NZ * N * N * 8 + nb * N ** 3 * 4 + (nb * HP * L ** 3 * 4 if C1 else 0)
    if C1:
        L1, N1, NZ1 = L // 2, L, L // 2 + 1
        alg["coarse"] = (C1 * L1 ** 3 * 4 + 2 * nb * C1 * NZ1 * L1 * L1 * 8 + C1 * NZ1 * N1 * N1 * 8 +
                         2 * nb * C1 * NZ1 * N1 * N1 * 8 + nb * HP * N1 ** 3 * 4)
    return alg


def protein_pair_volumes(dev, L=80, res=1.25):
    """Representation volumes of a synthetic protein-sized pair as Docker.dockSE3 prepares them (Docker.py:204-209): PDB
    files -> typed coordinates -> 11-type densities at box 80 -> E3MultiResRepr4x4(multiplier=8) on the HIP convolutions;
    forbidden volumes = the summed densities (Docker.py:221-225).  -> host tensors rec [fine, coarse], lig [...], recf, ligf"""
    import tempfile
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_pdb import write_protein_like_pdb
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    with tempfile.TemporaryDirectory(prefix="dlpd_rp_") as tmp, torch.no_grad():
        be = CoordsBackend()
        torch.manual_seed(3)
        repr_ = E3MultiResRepr4x4(multiplier=8).to(dev).eval()
        centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)
        out = []
        for name, nres, seed in (("rec", 160, 31), ("lig", 110, 32)):
            f = os.path.join(tmp, name + ".pdb")
            write_protein_like_pdb(f, nres, seed)
            coords, _, rn, _, an, nat = be.pdb2coords([f])
            typed, counts, offs = be.assign_types(coords, rn, an, nat)
            a, b = be.get_bbox(typed, be.last_num_typed)
            typed = be.translate(typed, -(a + b) * 0.5 + centre, be.last_num_typed)
            dens = be.project(typed, counts, offs, L, res, dev)
            vols = repr_(dens)
            out.append(([v[0].float().cpu() for v in vols], dens.sum(dim=1)[0].float().cpu()))
        (rec, recf), (lig, ligf) = out
    return rec, lig, recf, ligf


def build_workload(name, args, dev):
    """-> engine (receptor + ligand resident), host copies for the CPU baseline, metadata."""
    import torch
    from deeplocalproteindocking_amd.engine import DockingEngine
    from deeplocalproteindocking_amd.Models import SimpleFilter
    C, L, C1, angle, desc = WORKLOADS[name]
    if name == args.workload and name != "real_protein":
        C, L = args.channels or C, args.box or L
    t_host = time.perf_counter()
    if name == "real_protein":
        rec, lig, recf, ligf = protein_pair_volumes(dev, L)
    else:
        rec0, lig0 = synthetic_volumes(C, L, 0)
        recf, ligf = synthetic_forbidden(L)
        rec, lig = [rec0], [lig0]
        if C1:
            r1, l1 = synthetic_volumes(C1, L // 2, 1)
            rec.append(r1)
            lig.append(l1)
    torch.manual_seed(1)
    filt = SimpleFilter([C] + ([C1] if C1 else []))
    thr = clash_threshold(recf, ligf)
    W = filt.parameters_tuple()
    if name == "real_protein":
        # A pose WITHOUT contact (all correlations zero: most of the grid for protein-shaped volumes) scores
        # W2 relu(b1) + b2 whatever the rotation.  Training drives that constant to the neutral end of the scale; the seeded
        # random filter puts it wherever it falls, possibly below every contact score -- the ranked list is then 2000 copies of
        # the constant and every rotation's candidate list overflows into the full radix select.  The output bias is shifted
        # so that the no-contact score is zero (a property of the workload's model, stated in its description).
        W1_, b1_, W2_, b2_ = [w.clone() for w in W]
        c0 = float((W2_.reshape(1, -1) @ torch.relu(b1_.reshape(-1, 1))).reshape(-1)[0] + b2_.reshape(-1)[0])
        W = (W1_, b1_, W2_, b2_ - c0)
        desc += "; filter output bias shifted so that a pose without contact scores 0"
    if getattr(args, "hidden", None) and name == args.workload:
        # another hidden width (select_model's multiplier moves it: ProteinRepresentationModels.py:24,35-36): Xavier-like
        g = torch.Generator().manual_seed(2)
        Ct, H = C + C1, int(args.hidden)
        W = (torch.randn(H, Ct, generator=g) * (2.0 / (H + Ct)) ** 0.5, torch.zeros(H), torch.randn(1, H, generator=g) * (2.0 / (H + 1)) ** 0.5,
             torch.zeros(1))
    host_inputs_s = time.perf_counter() - t_host
    eng = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, has_clash=True, max_conf=args.max_conf,
                        batch=args.batch, device=dev, coarse_channels=C1, k3_form=args.k3_form, k1_form=getattr(args, "k1_form", 0),
                        packed_receptor=not getattr(args, "natural_receptor", False),
                        sparse_k1={"auto": None, "on": True, "off": False}[getattr(args, "k1_occupancy", "auto")])
    eng.set_receptor(rec[0], recf, rec[1] if C1 else None)
    eng.set_ligand(lig[0], ligf, lig[1] if C1 else None)
    return eng, dict(C=C, L=L, C1=C1, angle=angle, desc=desc, rec=rec, lig=lig, recf=recf, ligf=ligf, W=W, thr=thr,
                     host_inputs_s=host_inputs_s)


def time_steps(eng, Rd, idd, tr_of, qd_of, nb, first, count, mark=None):
    for i in range(first, first + count):
        sl = slice(i * nb, (i + 1) * nb)
        eng.step(Rd[sl], idd[sl], mark=mark, transposed=bool(tr_of[i * nb]), quads=bool(qd_of[i * nb]))


def dry_run(args):
    """The launch path without a GPU (CPU test of `python bench.py --gpus N`): rendezvous, one all-reduce."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        assert float(t) == world * (world + 1) / 2
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def list_sha256(ent):
    """sha256 of a ranked list (rot, flat index, score, pick) in fixed dtypes: two runs compare entry for entry."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for a, ty in zip(ent, (np.int64, np.int64, np.float32, np.int64)):
        h.update(np.ascontiguousarray(a, dtype=ty).tobytes())
    return h.hexdigest()


def sharded_search(eng, R_all, ids_global, rank, world, K, dist, dev):
    """The search as Docker runs it on W ranks (Docker.py:211-236): rank r scores ``ids_global[r::W]``
    (ascending), then ONE all-gather of the per-rank lists and the deterministic merge.  -> (merged entries,
    seconds = max over ranks, barrier to barrier)."""
    import numpy as np
    import torch
    from deeplocalproteindocking_amd.Docker.Docker import all_gather_top_entries
    mine = np.sort(np.asarray(ids_global)[rank::world])
    on_gpu = torch.device(dev).type == "cuda"               # (the CPU tests run this function on the emulated kernels + gloo)
    eng.reset_top()
    if on_gpu:
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if len(mine):
        eng.search(R_all[mine], rot_ids=mine)
    ent = eng.top_entries()
    if dist is not None:
        ent = all_gather_top_entries(ent, K, world, None, dev, always=True)
        dist.barrier()
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=collective_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
    return ent, dt


def collective_device(dist, dev):
    """Where a tensor handed to a collective lives: the rank's GPU on nccl (RCCL), the host on gloo."""
    return dev if dist.get_backend() == "nccl" else "cpu"


def run_rank(args):
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import __graft_entry__ as entry
    entry.build()
    # host threads: the synthetic inputs are generated on the CPU by every rank; never oversubscribe the
    # cgroup quota (the GPU boxes show 256 logical CPUs behind a 16-core quota)
    torch.set_num_threads(max(1, usable_cores() // max(1, world)))
    if args.same_device and args.backend == "nccl" and world > 1:
        raise SystemExit("bench.py: --same_device needs --backend gloo (RCCL wants one device per rank)")
    dev = torch.device("cuda", 0 if args.same_device else local_rank)
    if args.same_device and world > 1 and rank == 0:
        print("bench.py: --same_device puts %d ranks on ONE GPU: kernels of different processes share CUs, which on this hardware can "
              "change low mantissa bits of a few scores (inside the 1e-4 parity band; INTEGRATION.md, 'Sharing the GPU') -- a test mode "
              "for the multi-rank code path, not a way to run searches" % world, file=sys.stderr, flush=True)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)

    from deeplocalproteindocking_amd.engine import DockingEngine
    from deeplocalproteindocking_amd.Utils.Rotations import Rotations
    K, nb = args.max_conf, args.batch
    t_setup = time.perf_counter()
    eng, wl = build_workload(args.workload, args, dev)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup      # host_inputs_s + device_setup_s (split below)
    host_inputs_s, device_setup_s = wl["host_inputs_s"], setup_s - wl["host_inputs_s"]
    C, L, C1 = wl["C"], wl["L"], wl["C1"]
    eng_unfused, eng_hp, eng_prefilter = eng.fine_unfused, eng.HP, eng.prefilter
    N = 2 * L
    angle = args.angle_inc or wl["angle"]
    # the SOI files carry MitchellLab's licence and are not redistributed: read them when DLPD_ROTATIONS_DIR /
    # data/ has them, otherwise the SOI-sized generated set -- and the JSON line says which
    rot = Rotations(angle, allow_generated=True, verbose=False)
    R_all = rot.R
    nrot_total = R_all.shape[0]
    generated = rot.source == "generated"
    shard = np.arange(rank, nrot_total, world)                # this rank's interleaved shard
    seq_ids, seq_key = visiting_sequence(DockingEngine, R_all, shard, nb)
    if args.rotation_order != "set":         # (diagnostic: the list does not depend on the order, K1's gather locality might)
        perm = np.random.RandomState(0).permutation(len(seq_ids)) if args.rotation_order == "random" else \
            np.concatenate([np.arange(j, len(seq_ids), 8) for j in range(8)])
        seq_ids, seq_key = seq_ids[perm], seq_key[perm]
    shard_batches = len(seq_ids) // nb
    # K timed + W warm-up steps = batches spread evenly over the whole sequence (K > shard_batches repeats batches)
    def spread(n):
        return (np.arange(n) * shard_batches // max(n, 1)) % shard_batches if n > shard_batches \
            else np.linspace(0, shard_batches - 1, n).astype(int)
    warm_pick, time_pick = spread(max(args.warmup, 1)), spread(args.steps)
    head_n = min(20, shard_batches)
    pick = np.concatenate([warm_pick, time_pick, np.arange(head_n)])
    rows = np.concatenate([np.arange(j * nb, (j + 1) * nb) for j in pick])
    ids, key_of = seq_ids[rows], seq_key[rows]
    tr_of, qd_of = key_of >= 2, (key_of % 2) == 1
    Rd = R_all[ids].to(device=dev, dtype=torch.float32).contiguous()
    idd = torch.as_tensor(ids, dtype=torch.int32).to(dev)
    w0, t_first, h_first = 0, len(warm_pick), len(warm_pick) + len(time_pick)

    eng.reset_top()
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, w0, args.warmup)
    eng.finish()
    # the warm-up covers the step's collective too: the first all-gather of a communicator sets up its connections
    # (RCCL does that lazily), which belongs to the job's start-up, not to a step
    warm_gather_s = timed_gather_s = 0.0
    if dist is not None:
        from deeplocalproteindocking_amd.Docker.Docker import all_gather_top_entries
        t_g = time.perf_counter()
        all_gather_top_entries(eng.top_entries(), K, world, None, dev, always=True)
        warm_gather_s = time.perf_counter() - t_g

    # CPU-baseline sample: cpu_rotations / 4 rotations from the head of each search group; their GPU scores
    # (each through the K1/K2 variant the search uses for it) are kept for the parity figure
    V_first = R_cpu = grp_cpu = None
    if rank == 0 and world == 1 and args.cpu_rotations > 0:
        per = max(1, args.cpu_rotations // 4)
        sel, grp_cpu = [], []
        for k in range(4):
            g = seq_ids[seq_key == k][:per]
            sel.append(g)
            grp_cpu += [GROUP_NAMES[k]] * len(g)
        sel = np.concatenate(sel)
        R_cpu = R_all[sel].numpy()
        Vs = []
        for j, rid in enumerate(sel):
            k = GROUP_NAMES.index(grp_cpu[j])
            Rj = R_all[rid:rid + 1].to(device=dev, dtype=torch.float32).contiguous()
            Vs.append(eng.score_batch(Rj, transposed=k >= 2, quads=(k % 2) == 1).cpu())
        V_first = torch.cat(Vs)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.reset_top()
    barrier()
    t0 = time.perf_counter()
    timer = StageTimer()
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, t_first, args.steps, mark=timer.mark)
    entries = eng.top_entries()                              # waits for the side stream; D2H of this rank's list
    if dist is not None:                                     # single all-gather + deterministic merge
        t_g = time.perf_counter()
        entries = all_gather_top_entries(entries, K, world, None, dev, always=True)
        timed_gather_s = time.perf_counter() - t_g
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, setup_s, host_inputs_s, device_setup_s], dtype=torch.float64, device=collective_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, setup_s, host_inputs_s, device_setup_s = (float(v) for v in t.tolist())
    stages = timer.summary()                                 # K1/K2/K3 as measured inside the timed region

    # untimed extras: (i) the head of the set (the z-dominant rotations, cheapest K1 gather); (ii) the top-K
    # kernels, which run on the side stream inside the timed region, serially on the main stream
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, h_first, 1)
    eng.finish()
    torch.cuda.synchronize()
    ts = time.perf_counter()
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, h_first, head_n)
    eng.finish()
    torch.cuda.synchronize()
    head_ms = (time.perf_counter() - ts) / head_n * 1e3
    tk = StageTimer()
    cs = eng.top.new_candidate_set() if eng.prefilter else None   # the steady-state path: candidate lists from K3
    for i in range(h_first, h_first + min(3, head_n)):
        sl = slice(i * nb, (i + 1) * nb)
        V = eng.score_batch(Rd[sl], transposed=bool(tr_of[i * nb]), quads=bool(qd_of[i * nb]), cset=cs)
        tk.mark("begin")
        eng.select_batch(V, nb, eng._cset_used)
        tk.mark("topk_select")
        eng.merge_batch(idd[sl], nb)
        tk.mark("topk_merge")
    torch.cuda.synchronize()
    stages.update(tk.summary())

    sustained = None
    if rank == 0 and world == 1 and args.sustained_s > 0:
        sustained = sustained_measurement(eng, R_all, seq_ids, seq_key, nb, N, elapsed / args.steps, args.sustained_s, dev)

    # untimed extras that EVERY world size runs, so that the driver's N = 1 and N = 8 lines can be compared:
    # (i) gather_check: a fixed subset of the set (the first n rotations of the single-process visiting sequence),
    #     sharded r::W, through the one all-gather + merge -- the list hash must not depend on W;
    # (ii) strong: the complete set sharded over the ranks (seconds, rotations/s, list hash, the set-up beside it), or,
    #     where that would not fit --strong_s (N = 1), the prefix of the sequence that does.
    switches = eng.switches()
    gather_check = strong = None
    all_ids = np.arange(nrot_total)
    seq_all, _ = visiting_sequence(DockingEngine, R_all, all_ids, nb)
    if args.gather_rotations > 0:
        sub = seq_all[:min(args.gather_rotations, len(seq_all))]
        ent, dt = sharded_search(eng, R_all, sub, rank, world, K, dist, dev)
        gather_check = {"rotations": int(len(sub)), "list_sha256": list_sha256(ent), "list_entries": int(len(ent[0])),
                        "world_size_seen": (dist.get_world_size() if dist is not None else 1),
                        "backend": (dist.get_backend() if dist is not None else None), "seconds": dt,
                        "sample": "the first %d rotations of the single-process visiting sequence, sharded r::W, one "
                                  "all-gather + deterministic merge; the hash is the same at every world size" % len(sub)}
    if args.strong_s > 0:
        s_per_rot = elapsed / args.steps / nb
        complete = nrot_total / world * s_per_rot <= args.strong_s
        ids_strong = all_ids if complete else seq_all[:max(nb, int(args.strong_s / s_per_rot) // nb * nb)]
        ent, dt = sharded_search(eng, R_all, ids_strong, rank, world, K, dist, dev)
        strong = {"complete_set": bool(complete), "rotations": int(len(ids_strong)), "world_size": world, "seconds": dt,
                  "rot_per_s": len(ids_strong) / dt, "value": len(ids_strong) * float(N) ** 3 / dt, "unit": "pose scores/s",
                  "list_sha256": list_sha256(ent), "list_entries": int(len(ent[0])), "per_rank_setup_s": setup_s,
                  "host_inputs_s": host_inputs_s, "device_setup_s": device_setup_s,
                  "seconds_incl_setup": dt + setup_s, "seconds_incl_device_setup": dt + device_setup_s,
                  "sample": ("the complete %d-rotation set, rank r scoring rotations r::W (Docker.shard), ending with the one "
                             "all-gather + merge" % nrot_total) if complete else
                            ("the first %d rotations of the visiting sequence: the complete set would exceed --strong_s %.0f s "
                             "at this world size" % (len(ids_strong), args.strong_s))}

    real_shapes = c48l80 = e3 = real_protein = None
    if rank == 0 and world == 1 and args.workload == "config2" and not args.no_real_shapes:
        del eng
        torch.cuda.empty_cache()
        real_shapes = short_measurement("real", args, dev, R_all, nb)
        if not args.no_extras:
            c48l80 = short_measurement("c48l80", args, dev, R_all, nb, nsteps=12)
            try:
                # protein-shaped contents at the same shapes: with the K1 occupancy maps (the default for such a ligand) and without
                real_protein = short_measurement("real_protein", args, dev, R_all, nb)
                args.k1_occupancy, keep = "off", args.k1_occupancy
                try:
                    dense = short_measurement("real_protein", args, dev, R_all, nb)
                finally:
                    args.k1_occupancy = keep
                real_protein["without_k1_occupancy_maps"] = {"ms_per_step": dense["ms_per_step"], "stages": dense["stages"]}
            except Exception as exc:
                real_protein = {"error": repr(exc)}
            try:
                e3 = e3_measurement(dev, nb)
            except Exception as exc:                             # an extra must never cost the headline line
                e3 = {"error": repr(exc)}

    if rank == 0:
        poses = float(args.steps) * nb * N ** 3 * world
        alg = algorithmic_bytes(C, L, C1, nb, K, unfused=eng_unfused, HP=eng_hp, prefilter=eng_prefilter)
        main_stages = [k for k in stages if k in alg and not k.startswith("topk")]
        dom = max(main_stages, key=lambda k: stages[k])
        traffic = traffic_src = None
        if world == 1 and args.cpu_rotations > 0 and not args.no_pmc:
            # the dominant kernel's HBM-side bytes measured NOW: two short child runs of this script under
            # rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in separate passes, as the guide prescribes)
            traffic, traffic_src = live_pmc_traffic(args, dom, N, eng_hp)
        if traffic is None:
            traffic, traffic_src = pmc_traffic(args.workload, C, L, nb, dom)
        secondary = {}
        if world == 1 and args.cpu_rotations > 0 and not args.no_pmc:
            secondary = live_pmc_secondary(args, main_stages, N)

        def roof(k):
            ach = alg[k] / (stages[k] * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": k, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg[k], "avg_launch_ms": stages[k]}
        roofline = roof(dom)
        roofline.update({"traffic": traffic, "traffic_source": traffic_src})
        # `bound`: the HBM figure stays the contract's (algorithmic bytes / launch time); beside it the units the counters say
        # the kernel keeps busier than the HBM.  If vector issue + LDS (they add up in these kernels: EXPERIMENTS.md) exceed
        # the HBM occupancy by MEASURED traffic, the kernel is not an HBM-bound one and the line says so.
        if dom in secondary:
            roofline["secondary"] = secondary[dom]
            hbm_frac_traffic = (traffic / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else roofline["frac"]
            roofline["hbm_frac_on_measured_traffic"] = hbm_frac_traffic
            if secondary[dom]["valu_plus_lds_frac"] > hbm_frac_traffic:
                roofline["bound"] = "valu+lds"
                roofline["bound_note"] = ("vector issue (>= %.2f) + LDS (%.2f) = %.2f of their cycles against %.2f of the HBM peak on "
                                          "measured traffic: no single unit is saturated; the kernel is paced by its vector + LDS "
                                          "work, the HBM fraction is what is left of 8 TB/s, not what is missing"
                                          % (secondary[dom]["valu_issue_frac"], secondary[dom]["lds_active_frac"],
                                             secondary[dom]["valu_plus_lds_frac"], hbm_frac_traffic))
        rot_src = "generated SOI-sized substitute set" if generated else os.path.basename(rot.source)
        if (C, L, C1) == (48, 64, 0):
            metric = "pose correlations/sec (48ch x 64^3 pair, %s)" % (
                "%d-degree SOI-sized generated rotation set" % angle if generated else os.path.basename(rot.source))
        else:
            metric = "pose correlations/sec (%s pair)" % (("[%dch x %d^3, %dch x %d^3]" % (C, L, C1, L // 2)) if C1
                                                         else "%dch x %d^3" % (C, L))
        ms_step = elapsed / args.steps * 1e3
        moved = sum(alg[k] for k in main_stages)
        sb_mb, floor_mb = SURVEY_MB_PER_ROT.get(args.workload, (None, None)) if (args.channels, args.box) == (None, None) \
            else (None, None)
        step = {"ms_per_step": ms_step, "bytes_moved_per_step": moved,
                "achieved_GBps": moved / (ms_step * 1e-3) / 1e9, "frac_of_peak_on_bytes_moved": moved / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "bytes_moved = sum of the stage kernels' algorithmic input + output (K1 + K2 + K3 ...), what the fused "
                        "pipeline really transfers per batch; the compulsory floor and the stage-boundary model are SURVEY 8(d)'s"}
        if sb_mb:
            rps = args.steps * nb / elapsed
            step.update({"compulsory_floor_MB_per_rotation": floor_mb,
                         "frac_of_peak_on_compulsory_floor": rps * floor_mb * 1e6 / 1e9 / HBM_PEAK_GBS,
                         "stage_boundary_model_MB_per_rotation": sb_mb,
                         "stage_boundary_model_GBps_nominal": rps * sb_mb * 1e6 / 1e9})
        out = {
            "metric": metric, "value": poses / elapsed, "unit": "pose scores/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # a step = one launch of config.rotations_per_step rotations (32 since round 6, 16 before): the figure comparable with
            # rounds 1-5's ms_per_step is this one
            "ms_per_16_rotations": ms_step * 16.0 / nb,
            **({"same_device_note": "ranks share one GPU: co-resident kernels of different processes can perturb low bits of a few scores "
                                    "on this hardware (parity band only; INTEGRATION.md)"} if (args.same_device and world > 1) else {}),
            "config": {"workload": "%s, %d-degree SOI-sized rotation set (%d rotations, %s), max_conf=%d" %
                                   (wl["desc"] if (args.channels, args.box) == (None, None) else
                                    "synthetic %d-channel %d^3 pair" % (C, L), angle, nrot_total, rot_src, K),
                       "rotations_per_step": nb, "ms_per_16_rotations": ms_step * 16.0 / nb,
                       "rotations_timed_per_gpu": args.steps * nb,
                       "timed_steps": "batches spread evenly over the rank's whole %d-batch visiting sequence "
                                      "(all four search groups)" % shard_batches,
                       "translations_per_rotation": N ** 3, "sharding": "rotations interleaved over %d rank(s)" % world,
                       "world_size_seen_by_the_collective": (dist.get_world_size() if dist is not None else 1),
                       "collective_backend": (dist.get_backend() if dist is not None else None),
                       "kernel_switches": switches,
                       "environment": {k: v for k, v in os.environ.items() if k.startswith("DLPD_")},
                       "clip": 5.0, "threshold_clash": wl["thr"],
                       "masked_fraction": None if V_first is None else float((V_first == 0).float().mean())},
            "rot_per_s": args.steps * nb * world / elapsed,
            "roofline": roofline,
            "rooflines": {k: dict(roof(k), **({"secondary": secondary[k]} if k in secondary else {})) for k in stages if k in alg},
            "stages": {k: {"ms_per_launch": v, "alg_GBps": (alg[k] / (v * 1e-3) / 1e9 if k in alg else None)}
                       for k, v in stages.items()},
            "step": step,
            "head_of_set": {"value": nb * N ** 3 * world / (head_ms * 1e-3), "unit": "pose scores/s", "ms_per_step": head_ms,
                            "sample": "the first %d batches of the visiting sequence (rotations about z, cheapest K1 gather); "
                                      "untimed extra, NOT the headline" % head_n},
            "top_entries": int(len(entries[0])),
        }
        out["per_rank_setup_s"] = setup_s
        if dist is not None:       # rank 0's all-gather + merge: inside the timed region, and the untimed first one of the warm-up
            out["collective"] = {"gather_and_merge_s_in_timed_region": timed_gather_s, "first_gather_s_in_warmup": warm_gather_s}
        # the part a real pair pays once (upload, receptor spectrum, channels-last copy, workspaces) vs the part that only
        # the synthetic benchmark has (drawing the volumes on the host; a real pair gets them from the representation)
        out["setup"] = {"host_inputs_s": host_inputs_s, "device_setup_s": device_setup_s,
                        "note": "host_inputs_s = synthetic volumes + filter + clash threshold drawn on the CPU (a bench artefact, "
                                "no part of a real pair); device_setup_s = engine workspaces, upload, receptor spectrum, "
                                "channels-last ligand copy (paid once per pair); max over ranks"}
        if sustained is not None:
            out["sustained"] = sustained
        if gather_check is not None:
            out["gather_check"] = gather_check
        if strong is not None:
            out["strong"] = strong
        if real_shapes is not None:
            out["real_shapes"] = real_shapes
        if c48l80 is not None:
            out["c48l80"] = c48l80
        if real_protein is not None:
            out["real_protein"] = real_protein
        if e3 is not None:
            out["e3"] = e3
        if world == 1 and args.cpu_rotations > 0 and V_first is not None:
            out["cpu_baseline"] = cpu_baseline(wl["rec"], wl["lig"], wl["recf"], wl["ligf"], [w.cpu() for w in wl["W"]],
                                               R_cpu, grp_cpu, wl["thr"], K, V_first)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def sustained_measurement(eng, R_all, seq_ids, seq_key, nb, N, s_per_step, seconds, dev):
    """Untimed extra of the default line: `seconds` worth of CONSECUTIVE batches (the search as Docker runs it, from the
    head of the visiting sequence) -- long enough for the power-limited steady-state clock -- with the sha256 of the
    resulting ranked list, so that two driver runs can be compared entry for entry."""
    import numpy as np
    import torch
    nsteps = int(min(len(seq_ids) // nb, max(32, seconds / max(s_per_step, 1e-6))))
    ids, key_of = seq_ids[:nsteps * nb], seq_key[:nsteps * nb]
    tr_of, qd_of = key_of >= 2, (key_of % 2) == 1
    Rd = R_all[ids].to(device=dev, dtype=torch.float32).contiguous()
    idd = torch.as_tensor(ids, dtype=torch.int32).to(dev)
    eng.reset_top()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, 0, nsteps)
    ent = eng.top_entries()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"steps": nsteps, "seconds": dt, "ms_per_step": dt / nsteps * 1e3, "value": nsteps * nb * N ** 3 / dt,
            "unit": "pose scores/s", "rotations": nsteps * nb, "list_sha256": list_sha256(ent), "list_entries": int(len(ent[0])),
            "sample": "the first %d batches of the visiting sequence, consecutively (untimed extra, NOT the headline)" % nsteps}


def short_measurement(name, args, dev, R_all, nb, nsteps=24):
    """A short stratified measurement of another workload (the N = 160 pipeline of the reference's real
    shapes), reported as an extra object of the default line so that it has a driver-run number."""
    import numpy as np
    import torch
    from deeplocalproteindocking_amd.engine import DockingEngine
    eng, wl = build_workload(name, args, dev)
    L, C, C1 = wl["L"], wl["C"], wl["C1"]
    N = 2 * L
    seq_ids, seq_key = visiting_sequence(DockingEngine, R_all, np.arange(R_all.shape[0]), nb)
    nbat = len(seq_ids) // nb
    pick = np.linspace(0, nbat - 1, nsteps + 2).astype(int)
    rows = np.concatenate([np.arange(j * nb, (j + 1) * nb) for j in pick])
    ids, key_of = seq_ids[rows], seq_key[rows]
    tr_of, qd_of = key_of >= 2, (key_of % 2) == 1
    Rd = R_all[ids].to(device=dev, dtype=torch.float32).contiguous()
    idd = torch.as_tensor(ids, dtype=torch.int32).to(dev)
    eng.reset_top()
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, 0, 2)
    eng.finish()
    torch.cuda.synchronize()
    timer = StageTimer()
    t0 = time.perf_counter()
    time_steps(eng, Rd, idd, tr_of, qd_of, nb, 2, nsteps, mark=timer.mark)
    eng.finish()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / nsteps * 1e3
    stages = timer.summary()
    alg = algorithmic_bytes(C, L, C1, nb, args.max_conf, unfused=eng.fine_unfused, HP=eng.HP)
    sb_mb, floor_mb = SURVEY_MB_PER_ROT[name]
    rps = nb / (ms * 1e-3)
    sw = eng.switches()
    del eng
    torch.cuda.empty_cache()
    return {"workload": wl["desc"], "kernel_switches": sw, "steps": nsteps, "rotations_per_step": nb, "ms_per_step": ms,
            "ms_per_16_rotations": ms * 16.0 / nb, "rot_per_s": rps,
            "value": rps * N ** 3, "unit": "pose scores/s",
            "stages": {k: {"ms_per_launch": v, "alg_GBps": (alg[k] / (v * 1e-3) / 1e9 if k in alg else None),
                           "frac_of_peak": (alg[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS if k in alg else None)}
                       for k, v in stages.items()},
            "frac_of_peak_on_compulsory_floor": rps * floor_mb * 1e6 / 1e9 / HBM_PEAK_GBS,
            "stage_boundary_model_GBps_nominal": rps * sb_mb * 1e6 / 1e9}


def e3_measurement(dev, nb, nsteps=6):
    """`e3` extra: what a batch of Docker.dockE3 (Docker.py:135-182) costs at the reference's box 80 -- the ligand
    rotated in coordinate space and re-projected (Docker.py:163-165), re-represented by the plugin
    (E3MultiResRepr4x4(multiplier=8): every convolution on dlpd_conv3d, Docker.py:166-167) and scored by the fused
    engine from the batch's own volumes -- on a synthetic protein-sized pair, per launch of `nb` rotations."""
    import tempfile
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_pdb import write_protein_like_pdb
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4, GlobalDockingModel, SimpleFilter
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    from deeplocalproteindocking_amd.Utils.Rotations import Rotations
    with tempfile.TemporaryDirectory(prefix="dlpd_e3_") as tmp, torch.no_grad():
        rec_pdb, lig_pdb = os.path.join(tmp, "rec.pdb"), os.path.join(tmp, "lig.pdb")
        write_protein_like_pdb(rec_pdb, 160, 31)
        write_protein_like_pdb(lig_pdb, 110, 32)
        torch.manual_seed(3)
        repr_ = E3MultiResRepr4x4(multiplier=8)
        model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=3.0).to(dev)
        model.eval()
        # (as in the real_protein workload: the seeded random filter's score of a pose WITHOUT contact is shifted to zero, so
        #  that the ranked list is made of contact poses and the candidate lists of K3 work as they do for a trained model)
        W1_, b1_, W2_, b2_ = model.filter.parameters_tuple()
        model.filter.fc[2].bias.data -= (W2_.reshape(1, -1) @ torch.relu(b1_.reshape(-1, 1))).reshape(-1)[0] + b2_.reshape(-1)[0]
        be = CoordsBackend()
        rot = Rotations(20, allow_generated=True, verbose=False)
        dk = Docker(model, angle_inc=20, box_size=80, resolution=1.25, max_conf=2000, device=dev, coords_backend=be,
                    rotations=rot.R.numpy())
        L, res = 80, 1.25
        lcoords, lnat, loff, _, _ = dk.load_batch([lig_pdb], bbox_center=False)
        rcoords, rnat, roff, _, rnatoms = dk.load_batch([rec_pdb], bbox_center=False)
        rcoords = be.translate(rcoords, dk.box_center, rnatoms)
        lc, ln, lo = be.to_device(lcoords, lnat, loff, dev)
        receptor = be.project(rcoords, rnat, roff, L, res, dev)
        rv = model.representation(receptor)
        from deeplocalproteindocking_amd.Models.DockingModels import fused_filter_parameters
        eng = dk._make_engine([v.reshape((-1,) + tuple(v.shape[-3:])) for v in rv], receptor.sum(dim=1)[0], nb,
                              fused_filter_parameters(model))
        eng.reset_top()
        Rb = rot.R[:nb].to(device=dev, dtype=torch.float32).contiguous()
        ids = torch.arange(nb, dtype=torch.int32, device=dev)

        def timed(fn, n=nsteps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, r
        ms_proj, lig = timed(lambda: be.project(lc, ln, lo, L, res, dev, R=Rb, shift=dk.box_center))
        # the plugin as Docker._dockE3_fused calls it: tiles away from the ligand are neither computed nor written, the
        # volumes travel with their occupancy maps (round 6); "written": the same with every voxel written (round 5)
        import contextlib
        with_maps = bool(getattr(repr_, "supports_unwritten_outputs", False)) and dk.unwritten_activations

        def represent_as_docker(x):
            with (repr_.outputs_with_maps() if with_maps else contextlib.nullcontext()):
                return model.representation(x)

        def maps_of(v):
            o = (getattr(v[0], "dlpd_occupancy", None), getattr(v[1], "dlpd_occupancy", None) if eng.C1 else None)
            return o if o[0] is not None else None
        ms_repr, vols = timed(lambda: represent_as_docker(lig))
        ms_repr_written, _ = timed(lambda: model.representation(lig))
        ms_repr_dense = None
        if getattr(repr_, "use_tile_occupancy", False):            # the same network computing every tile (same bits)
            repr_.use_tile_occupancy = False
            try:
                ms_repr_dense, _ = timed(lambda: model.representation(lig))
            finally:
                repr_.use_tile_occupancy = True
        forb = lig.sum(dim=1)

        def engine_step():
            eng.step(None, ids, volumes=(vols[0], forb, vols[1] if eng.C1 else None), occupancy=maps_of(vols))
            eng.finish()
        ms_eng, _ = timed(engine_step)

        def body():
            l = be.project(lc, ln, lo, L, res, dev, R=Rb, shift=dk.box_center)
            v = represent_as_docker(l)
            eng.step(None, ids, volumes=(v[0], l.sum(dim=1), v[1] if eng.C1 else None), occupancy=maps_of(v))
        ms_serial, _ = timed(body, n=2 * nsteps)
        eng.finish()
        # ... and through Docker's own batch loop (Docker._dockE3_fused: the same two halves, one stream)
        nbat = 4 * nsteps
        batches = [list(range(nb))] * nbat

        def represent(bid):                          # (as Docker.dockE3's own closure)
            l = be.project(lc, ln, lo, L, res, dev, R=Rb, shift=dk.box_center, cells=bool(getattr(dk, "_project_cells", False)))
            l.dlpd_type_sum = be.project(lc, ln, lo, L, res, dev, R=Rb, shift=dk.box_center, sum_types=True)[:, 0]
            return l, model.representation(l)

        def docker_loop():
            dk._dockE3_fused(eng, batches, represent, None, nb)
            eng.finish()
        ms_loop, _ = timed(docker_loop, n=2)
        ms_all = ms_loop / nbat
        eng.finish()
        natoms = int(lnat.sum()) if hasattr(lnat, "sum") else None
        out = {"workload": "Docker.dockE3 at box 80, E3MultiResRepr4x4(multiplier=8) -> %s channels, synthetic %d / %d-residue pair"
                           % (repr_.get_num_outputs(), 160, 110),
               "rotations_per_launch": nb, "ms_projection": ms_proj, "ms_representation": ms_repr, "ms_engine": ms_eng,
               # the headline is ONE designated measurement: Docker's own batch loop (Docker._dockE3_fused); the hand-written
               # serial loop of the same three calls is reported beside it, not mixed in
               "ms_per_launch": ms_all, "ms_per_16_rotations": ms_all * 16.0 / nb, "ms_per_launch_serial": ms_serial,
               "ms_per_launch_docker_loop": ms_all,
               "tile_occupancy": bool(getattr(repr_, "use_tile_occupancy", False)),
               "unwritten_activations": with_maps,
               "ms_representation_writing_every_voxel": ms_repr_written,
               "ms_representation_computing_every_tile": ms_repr_dense,
               "rot_per_s": nb / (ms_all * 1e-3),
               "value": nb / (ms_all * 1e-3) * (2.0 * L) ** 3,
               "unit": "pose scores/s", "ligand_atoms": natoms, "path": "fused engine on the batch's own volumes",
               "filter": "seeded SimpleFilter, output bias shifted so that a pose without contact scores 0 (round 6)",
               "conv_precision": __import__("deeplocalproteindocking_amd.ops", fromlist=["CONV_PRECISION"]).CONV_PRECISION}
        dk.release_engine()
        del eng
        torch.cuda.empty_cache()
        return out


def pmc_traffic(workload, C, L, nb, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 note), newest round first."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            cfg = pmc["config"]
            if (cfg.get("channels"), cfg.get("box"), cfg.get("batch")) == (C, L, nb) and kernel in pmc["bytes_per_launch"]:
                t = pmc["bytes_per_launch"][kernel]
                scale = float(t.get("fetch_scale", 2.0))
                return (scale * t["fetch_kb"] + t["write_kb"]) * 1024.0, \
                    "profiles/%s (rocprofv3 --pmc FETCH_SIZE x %.0f / WRITE_SIZE; gfx950 tallies 128-byte read requests at 64 bytes)" % (
                        os.path.basename(path), scale)
        except Exception:
            continue
    return None, None


# FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section): gfx950 tallies a 128-byte read request at 64 bytes -> x 2.
# (Until round 4 K3 at N = 160 read its spectra in 64-byte runs -- 8-row tiles --, which are tallied at face value: an
# entry {("k3_zifft_filter", 160): 1.0} here; its 16-row tiles read whole lines like every other kernel.)
FETCH_SCALE = {}


def fetch_scale(stage, N, hidden_pad=None):
    """x 2 for kernels whose reads are whole 128-byte lines; 1 for K3's WIDE configurations (hidden widths above 24 at
    N = 160 / above 32 at N = 128: two voxels per filter thread on 8-row tiles, i.e. 64-byte runs, tallied at face value)."""
    if stage == "k3_zifft_filter" and hidden_pad is not None and hidden_pad > (24 if N == 160 else 32):
        return 1.0
    return FETCH_SCALE.get((stage, N), 2.0)
# FETCH_SIZE per byte read, measured per load shape on a 4 GiB buffer read once (scripts/micro/fetch_size_shapes.hip,
# profiles/r06_fetch_size_shapes.txt): 16 bytes per lane, 1 KB per wave instruction 0.500 (the guide's rule); 8 bytes per lane in
# 64-byte runs 1 KB apart -- K2<128>'s receptor loads -- 0.727 (a mix of 128- and 64-byte requests); 16 bytes per lane in scattered
# 64-byte runs 0.605
FETCH_PER_BYTE = {"wide16": 0.500, "runs64_8B": 0.727, "runs64_16B": 0.605}


def k2_natural_read_bytes(raw_fetch, nb, CT, L):
    """K2<64 / 128> (k_xy_corr: natural-layout receptor) issues two load shapes: its A rows (wide: tallied at half) and the
    receptor values (64-byte runs: tallied at 0.727).  raw = 0.5 A + 0.727 rec_fetched with A = the algorithmic A bytes (read
    once, streaming) -> bytes really read, and how often the receptor slab crossed the fabric per launch."""
    A = nb * CT * (L + 1) * L * L * 8.0
    rec = CT * (L + 1) * (2 * L) ** 2 * 8.0
    rec_fetched = max(raw_fetch - FETCH_PER_BYTE["wide16"] * A, 0.0) / FETCH_PER_BYTE["runs64_8B"]
    return A + rec_fetched, rec_fetched / rec


STAGE_KERNELS = {"k1_rotate_zfft": ("k_rotate_zfft_cl<%d,", "k_rotate_zfft_cl_rs<%d>", "k_rotate_zfft<%d>"),
                 "k2_xy_corr": ("k_xy_corr<%d, 1>", "k_xy_corr_q4<%d,", "k_xy_corr_quad<%d,"),
                 "k3_zifft_filter": ("k_zifft_filter_rs<%d,",)}


def run_in_own_group(cmd, cwd, env, timeout):
    """Run ``cmd`` as the leader of a new process group, output discarded; on a timeout the WHOLE group is killed.
    -> return code (None after a timeout)."""
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        return None


def live_pmc_traffic(args, stage, N, hidden_pad=None):
    """(bytes per launch, source) of the stage's kernels from two rocprofv3 child passes of this script, or (None, None)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    pats = STAGE_KERNELS.get(stage)
    if not os.path.exists(exe) or pats is None:
        return None, None
    # never start a profiler from inside a profiled process (this run itself under rocprofv3 / rocprof)
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, None
    pats = tuple(p % N for p in pats)
    child = [sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--cpu_rotations", "0", "--no_real_shapes",
             "--sustained_s", "0", "--strong_s", "0", "--gather_rotations", "0", "--workload", args.workload,
             "--batch", str(args.batch), "--max_conf", str(args.max_conf), "--k3_form", str(args.k3_form), "--k1_form", str(args.k1_form)]
    for flag, val in (("--channels", args.channels), ("--box", args.box), ("--angle_inc", args.angle_inc), ("--hidden", args.hidden)):
        if val is not None:
            child += [flag, str(val)]
    per_launch = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(prefix="dlpd_pmc_") as tmp:
                env = dict(os.environ, TMPDIR="/tmp")
                # the profiler and the benchmark under it in a process group of their own: a timeout ends BOTH (killing
                # only rocprofv3 would leave its child on the GPU)
                rc = run_in_own_group([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--"] + child,
                                      cwd="/tmp", env=env, timeout=240)
                files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
                if rc != 0 or not files:
                    return None, None
                total, launches = 0.0, set()
                for row in csv.DictReader(open(files[0])):
                    name = row["Kernel_Name"].replace("void ", "")
                    if row["Counter_Name"] == counter and any(name.startswith(p) for p in pats):
                        total += float(row["Counter_Value"])
                        if name.startswith(pats[0]):
                            launches.add(row["Dispatch_Id"])
                if not launches:
                    return None, None
                per_launch[counter] = total / len(launches) * 1024.0            # KB -> bytes
        scale = fetch_scale(stage, N, hidden_pad)
        if stage == "k2_xy_corr" and N in (64, 128) and not getattr(args, "channels", None) and not getattr(args, "box", None):
            C_, L_, _, _, _ = WORKLOADS[args.workload]
            reads, refetch = k2_natural_read_bytes(per_launch["FETCH_SIZE"], args.batch, C_ + 1, L_)
            return reads + per_launch["WRITE_SIZE"], \
                ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes of bench.py --steps 6, per "
                 "launch of %s...): raw FETCH_SIZE %.0f bytes taken apart by load shape (scripts/micro/fetch_size_shapes.hip: the A rows, "
                 "16 bytes per lane, are tallied at 0.500 of their bytes, the receptor values, 64-byte runs, at 0.727) = %.0f bytes read -- "
                 "the A rows once, the receptor slab %.2f times per launch of %d rotations -- + raw WRITE_SIZE %.0f bytes"
                 % (pats[0], per_launch["FETCH_SIZE"], reads, refetch, args.batch, per_launch["WRITE_SIZE"]))
        return scale * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"], \
            ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes of bench.py "
             "--steps 6, per launch of %s...): raw FETCH_SIZE %.0f bytes x %.0f (gfx950 tallies a 128-byte read request at 64 "
             "bytes; a kernel reading 64-byte runs is taken at face value: FETCH_SCALE in bench.py) + raw WRITE_SIZE %.0f bytes"
             % (pats[0], per_launch["FETCH_SIZE"], scale, per_launch["WRITE_SIZE"]))
    except Exception:
        return None, None


SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")
N_XCD, N_CU, N_SIMD = 8, 256, 1024


def secondary_from_counters(c, dur_ns):
    """Per-launch counter averages of one kernel -> how busy the units that are NOT the HBM were (`roofline.secondary`).
    Units of the counters on this chip (calibrated on K2<128>, profiles/r05_pmc_units.txt): GRBM_GUI_ACTIVE is summed over
    the 8 XCDs (/ 8 = the kernel's cycles: 2.0 GHz x its duration); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles
    summed over all waves (x 4 / 1024 SIMDs / kernel cycles = 1.9 resident waves per SIMD for a kernel compiled for 2);
    SQ_LDS_IDX_ACTIVE counts cycles summed over the 256 CUs; SQ_INSTS_VALU counts wave instructions.
    A wave64 vector instruction holds its SIMD's issue port for 2 cycles when another wave fills the gaps and 4 when the
    wave is alone (MI355X_MICROARCH.md, cycle constants; v_pk_*_f32: twice that): valu_issue_frac prices every instruction at
    2 cycles (a LOWER bound of the port's occupancy), valu_active_frac is the hardware's own quad-cycle count."""
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
    clock_ghz = cyc / dur_ns if dur_ns else None
    assumed = False
    if not cyc or clock_ghz is None or not (0.5 <= clock_ghz <= 2.7):
        cyc, clock_ghz, assumed = (dur_ns or 0.0) * 2.0, 2.0, True                        # no usable cycle counter: 2.0 GHz assumed
    if not cyc:
        return None
    valu_issue = c.get("SQ_INSTS_VALU", 0.0) * 2.0 / (N_SIMD * cyc)
    valu_active = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (N_SIMD * cyc)
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (N_CU * cyc)
    return {"valu_issue_frac": valu_issue, "valu_active_frac": valu_active, "lds_active_frac": lds,
            "valu_plus_lds_frac": valu_issue + lds,
            "lds_bank_conflict_frac_of_lds_active": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
            "waves_per_simd": c.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / (N_SIMD * cyc),
            "kernel_cycles": cyc, "clock_GHz": clock_ghz, "clock_assumed": assumed,
            "launch_ms_under_the_counters": dur_ns * 1e-6 if dur_ns else None,
            "counters_per_launch": {k: c[k] for k in sorted(c)},
            "definitions": "valu_issue_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles); valu_active_frac = "
                           "SQ_ACTIVE_INST_VALU x 4 / (1024 x cycles); lds_active_frac = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles); "
                           "kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs"}


def live_pmc_secondary(args, stages, N):
    """{stage: secondary dict} from ONE more rocprofv3 child pass of this script with the SQ / GRBM counters (their own
    pass, --kernel-trace only beside them), or {}."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--cpu_rotations", "0", "--no_real_shapes",
             "--sustained_s", "0", "--strong_s", "0", "--gather_rotations", "0", "--workload", args.workload,
             "--batch", str(args.batch), "--max_conf", str(args.max_conf), "--k3_form", str(args.k3_form), "--k1_form", str(args.k1_form)]
    for flag, val in (("--channels", args.channels), ("--box", args.box), ("--angle_inc", args.angle_inc), ("--hidden", args.hidden)):
        if val is not None:
            child += [flag, str(val)]
    out = {}
    try:
        with tempfile.TemporaryDirectory(prefix="dlpd_pmc_") as tmp:
            rc = run_in_own_group([exe, "--kernel-trace", "--pmc"] + list(SQ_COUNTERS) + ["--output-format", "csv", "-d", tmp, "--"] + child,
                                  cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=240)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return {}
            rows = list(csv.DictReader(open(files[0])))
        for stage in stages:
            pats = STAGE_KERNELS.get(stage)
            if pats is None:
                continue
            main = pats[0] % N
            tot, launches, dur = {}, set(), {}
            for row in rows:
                name = row["Kernel_Name"].replace("void ", "")
                if name.startswith(main):
                    tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    launches.add(row["Dispatch_Id"])
                    if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                        dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            if launches:
                sec = secondary_from_counters({k: v / len(launches) for k, v in tot.items()},
                                              sum(dur.values()) / len(dur) if dur else None)
                if sec is not None:
                    sec["kernel"] = main
                    out[stage] = sec
    except Exception:
        return {}
    return out


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and (world_env is None or int(world_env) != args.gpus):
        if world_env is not None and int(world_env) > 1:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, world_env))
        raise SystemExit(spawn_ranks(args))
    if args.dry_run:
        return dry_run(args)
    run_rank(args)


if __name__ == "__main__":
    main()
