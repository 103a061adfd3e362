#!/usr/bin/env python3
"""Benchmark of the rotation x translation correlation search (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of `--batch` rotations of a synthetic
48-channel 64^3 pair (BASELINE config 2: oim06-sized rotation set, max_conf=2000): trilinear
rotation + z R2C (K1), per-slab 2-D FFT / conj-multiply / 2-D inverse (K2), z C2R + clip + filter MLP
+ clash mask (K3), per-rotation top-K select and the running global merge.  Inputs are resident in
HBM before the timed region.  With N ranks every rank scores its own interleaved shard of the
rotation set (weak scaling: per-GPU work fixed) and the timed region ends with the single all-gather
of the per-rank top lists and the deterministic merge.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (algorithmic
bytes per launch / its average launch duration measured with HIP events on the launch stream),
"stages" (ms per launch of every stage), "pipeline" (whole step against SURVEY.md 8(d)'s
3,080.3 MB/rotation stage-boundary model) and "cpu_baseline" (the CPU oracle = restated reference
path, timed on this box's host cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STAGE_BOUNDARY_MB_PER_ROT = {(48, 64): 3080.3, (4, 32): 41.6}


def synthetic_pair(C, L, seed=0):
    """SURVEY.md 8(d) synthetic inputs: N(0,1) x smooth radial envelope representation volumes,
    relu-blob forbidden volumes, SimpleFilter([C]) after manual_seed(1)."""
    from deeplocalproteindocking_amd.Models import SimpleFilter
    g = torch.Generator().manual_seed(seed)
    ar = (torch.arange(L, dtype=torch.float32) - (L - 1) / 2.0) / (L / 2.0)
    r2 = ar[:, None, None] ** 2 + ar[None, :, None] ** 2 + ar[None, None, :] ** 2
    env = torch.exp(-1.5 * r2)
    amp = 206.0 / L ** 1.5              # per-channel correlation std ~2: the +-5 clip bites on the tails only
    rec = torch.randn(C, L, L, L, generator=g) * env * amp
    lig = torch.randn(C, L, L, L, generator=g) * env * amp

    def blob(shift):
        c = torch.tensor(shift, dtype=torch.float32)
        d2 = (ar[:, None, None] - c[0]) ** 2 + (ar[None, :, None] - c[1]) ** 2 + (ar[None, None, :] - c[2]) ** 2
        return torch.relu(torch.exp(-2.0 * d2) - 0.2)
    recf, ligf = blob((0.1, -0.05, 0.0)), blob((-0.05, 0.1, 0.05))
    torch.manual_seed(1)
    filt = SimpleFilter([C])
    return rec, lig, recf, ligf, filt


def clash_threshold(recf, ligf):
    """Median of the (unrotated) clash correlation (about half of the grid masked), floored at
    1e-3 of its maximum so the mask never depends on FFT round-off around an exact zero overlap."""
    L = recf.shape[0]
    N = 2 * L
    f = torch.fft.irfftn(torch.fft.rfftn(recf, s=(N, N, N)) * torch.conj(torch.fft.rfftn(ligf, s=(N, N, N))),
                         s=(N, N, N))
    return max(float(f.median()), 1e-3 * float(f.max()))


class StageTimer:
    def __init__(self):
        self.events = []

    def mark(self, name):
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


def cpu_baseline(rec, lig, recf, ligf, W, R, thr, K, nrot_sample, V_gpu=None):
    """The oracle (restated reference path incl. the reference's update_top loop) on the host."""
    from oracle import docking_oracle as orc
    L = rec.shape[-1]
    torch.set_num_threads(usable_cores())
    t0 = time.time()
    top, Vs = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], R[:nrot_sample],
                               *W, thr, K, clip=5.0, faithful_topk=True, return_V=True)
    dt = time.time() - t0
    out = {"value": nrot_sample * (2 * L) ** 3 / dt, "unit": "pose scores/s", "cores": torch.get_num_threads(),
           "kind": "port", "sample": "first %d rotations of the same rotation set and pair, %.1f s" % (nrot_sample, dt)}
    if V_gpu is not None:
        # parity of the GPU scores on the same rotations: voxels whose clash mask differs (clash
        # correlation within FFT round-off of the threshold) are counted, not compared
        errs, flips = [], 0
        for i in range(nrot_sample):
            same = (V_gpu[i] == 0) == (Vs[i] == 0)
            flips += int((~same).sum())
            errs.append(float((V_gpu[i] - Vs[i]).abs()[same].max() / Vs[i].abs().max()))
        out["parity"] = {"max_err_rel_to_max_abs_score": max(errs), "tolerance": 1e-4,
                         "mask_flips_at_threshold": flips, "voxels": nrot_sample * (2 * L) ** 3}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="rotations per step")
    ap.add_argument("--channels", type=int, default=48)
    ap.add_argument("--box", type=int, default=64)
    ap.add_argument("--angle_inc", type=int, default=6)
    ap.add_argument("--max_conf", type=int, default=2000)
    ap.add_argument("--cpu_rotations", type=int, default=8, help="rotations of the CPU baseline sample (0: skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    import __graft_entry__ as entry
    entry.build()
    # host threads: the synthetic inputs are generated on the CPU by every rank; never oversubscribe the
    # cgroup quota (the GPU boxes show 256 logical CPUs behind a 16-core quota)
    torch.set_num_threads(max(1, usable_cores() // max(1, world)))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from deeplocalproteindocking_amd.engine import DeviceTopList, DockingEngine
    from deeplocalproteindocking_amd.Utils.Rotations import Rotations
    C, L, K, nb = args.channels, args.box, args.max_conf, args.batch
    N = 2 * L
    rec, lig, recf, ligf, filt = synthetic_pair(C, L)
    thr = clash_threshold(recf, ligf)
    W = filt.parameters_tuple()
    rot = Rotations(args.angle_inc, verbose=False)
    R_all = rot.R
    nrot_total = R_all.shape[0]
    ids = np.arange(rank, nrot_total, world)                 # this rank's interleaved shard
    # visiting order of DockingEngine.search: rotations grouped by slab orientation (a per-launch choice),
    # set order inside a group; a step is one batch of that sequence
    Rn = R_all[ids].numpy()
    gkey = DockingEngine.prefers_transposed(Rn).astype(int) * 2 + DockingEngine.prefers_quads(Rn).astype(int)
    parts, keys = [], []
    for k in range(4):                                         # groups in the engine's order, whole batches only
        sel = ids[gkey == k]
        sel = sel[:len(sel) // nb * nb]
        parts.append(sel)
        keys.append(np.full(len(sel), k))
    ids, key_of = np.concatenate(parts), np.concatenate(keys)
    tr_of, qd_of = key_of >= 2, (key_of % 2) == 1
    shard_batches = len(ids) // nb
    need = (args.steps + args.warmup) * nb
    reps = (need + shard_batches * nb - 1) // (shard_batches * nb)
    ids_all, tr_all, qd_all = ids[:shard_batches * nb], tr_of[:shard_batches * nb], qd_of[:shard_batches * nb]
    ids, tr_of, qd_of = (np.tile(a, reps)[:need] for a in (ids_all, tr_all, qd_all))   # (wraps only if steps*batch > shard)
    Rd = R_all[ids].to(device=dev, dtype=torch.float32).contiguous()
    idd = torch.as_tensor(ids, dtype=torch.int32).to(dev)

    eng = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, has_clash=True, max_conf=K, batch=nb, device=dev)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    eng.reset_top()

    def step(i, mark=None):
        # K1,K2,K3 on the main stream; select+merge of the same batch on the engine's side stream
        # (overlapping the next batch), see DockingEngine.step
        sl = slice(i * nb, (i + 1) * nb)
        eng.step(Rd[sl], idd[sl], mark=mark, transposed=bool(tr_of[i * nb]), quads=bool(qd_of[i * nb]))

    def step_serial(i, mark):
        sl = slice(i * nb, (i + 1) * nb)
        V = eng.score_batch(Rd[sl], mark=mark, transposed=bool(tr_of[i * nb]), quads=bool(qd_of[i * nb]))
        eng.select_batch(V, nb)
        mark("topk_select")
        eng.merge_batch(idd[sl], nb)
        mark("topk_merge")

    for i in range(args.warmup):
        step(i)
    V_first = None
    eng.finish()
    if rank == 0 and world == 1 and args.cpu_rotations > 0:
        Rs = R_all[:args.cpu_rotations].to(device=dev, dtype=torch.float32).contiguous()
        V_first = torch.cat([eng.score_batch(Rs[i:i + nb]).cpu() for i in range(0, Rs.shape[0], nb)])
    timer = StageTimer()
    for i in range(min(args.warmup, 3)):                   # untimed: per-stage launch durations
        step_serial(i, timer.mark)
    torch.cuda.synchronize()
    stages = timer.summary()
    eng.reset_top()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    timer = StageTimer()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i, mark=timer.mark)
    entries = eng.top_entries()                              # waits for the side stream; D2H of this rank's list
    if world > 1:                                            # single all-gather + deterministic merge
        from deeplocalproteindocking_amd.Docker import Docker
        dk = Docker.__new__(Docker)
        dk.world_size, dk.max_conf, dk.process_group, dk.device = world, K, None, dev
        entries = dk._gather(entries)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # untimed extra: the rotation-dependent part (K1's gather) makes the head of the set cheaper than its
    # bulk, so also time batches spaced evenly over the WHOLE search sequence of this rank
    nsample = min(48, shard_batches)
    pick = np.linspace(0, shard_batches - 1, nsample).astype(int)
    Rs = R_all[np.concatenate([ids_all[j * nb:(j + 1) * nb] for j in pick])].to(device=dev, dtype=torch.float32).contiguous()
    Is = torch.as_tensor(np.concatenate([ids_all[j * nb:(j + 1) * nb] for j in pick]), dtype=torch.int32).to(dev)
    eng.step(Rs[:nb], Is[:nb], transposed=bool(tr_all[pick[0] * nb]), quads=bool(qd_all[pick[0] * nb]))
    eng.finish()
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for j in range(nsample):
        eng.step(Rs[j * nb:(j + 1) * nb], Is[j * nb:(j + 1) * nb], transposed=bool(tr_all[pick[j] * nb]),
                 quads=bool(qd_all[pick[j] * nb]))
    eng.finish()
    torch.cuda.synchronize()
    sustained_ms = (time.perf_counter() - ts) / nsample * 1e3

    if rank == 0:
        poses = float(args.steps) * nb * N ** 3 * world
        stages.update(timer.summary())                      # K1/K2/K3 as measured inside the timed region
        CT, NZ = C + 1, L + 1
        alg = {   # algorithmic bytes per launch (each kernel's compulsory input + output, fp32)
            "k1_rotate_zfft": CT * L ** 3 * 4 + nb * CT * NZ * L * L * 8,
            "k2_xy_corr": nb * CT * NZ * L * L * 8 + CT * NZ * N * N * 8 + nb * CT * NZ * N * N * 8,
            "k3_zifft_filter": nb * CT * NZ * N * N * 8 + nb * N ** 3 * 4,
            "topk_select": nb * N ** 3 * 4 + nb * K * 8,
            "topk_merge": nb * K * 8 + K * 16,
        }
        dom = max((k for k in stages if k in alg), key=lambda k: stages[k])
        ach = alg[dom] / (stages[dom] * 1e-3) / 1e9
        traffic = None           # PMC-measured bytes per launch of the same command (profiles/)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pmc["config"] == {"channels": C, "box": L, "batch": nb} and dom in pmc["bytes_per_launch"]:
                t = pmc["bytes_per_launch"][dom]
                traffic = (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0
        except Exception:
            traffic = None
        out = {
            "metric": "pose correlations/sec (48ch x 64^3 pair, oim06.eul)" if (C, L) == (48, 64)
                      else "pose correlations/sec (%dch x %d^3 pair)" % (C, L),
            "value": poses / elapsed, "unit": "pose scores/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: single synthetic %d-channel %d^3 pair, %d-degree SOI-sized "
                                   "rotation set (%d rotations, %s), max_conf=%d" %
                                   (C, L, args.angle_inc, nrot_total,
                                    "generated substitute" if rot.source == "generated" else os.path.basename(rot.source), K),
                       "rotations_per_step": nb, "rotations_timed_per_gpu": args.steps * nb,
                       "translations_per_rotation": N ** 3, "sharding": "rotations interleaved over %d rank(s)" % world,
                       "clip": 5.0, "threshold_clash": thr,
                       "masked_fraction": None if V_first is None else float((V_first == 0).float().mean())},
            "rot_per_s": args.steps * nb * world / elapsed,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per gfx950 note)",
                         "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": stages[dom]},
            "stages": {k: {"ms_per_launch": v, "alg_GBps": alg[k] / (v * 1e-3) / 1e9} for k, v in stages.items()},
        }
        sb = STAGE_BOUNDARY_MB_PER_ROT.get((C, L))
        if sb:
            gbs = out["rot_per_s"] / world * sb * 1e6 / 1e9
            out["pipeline"] = {"model": "SURVEY 8(d) stage-boundary %.1f MB/rotation" % sb, "achieved_GBps_per_gpu": gbs,
                               "frac_of_8TBps": gbs / HBM_PEAK_GBS}
        out["whole_set"] = {"value": nb * N ** 3 * world / (sustained_ms * 1e-3), "unit": "pose scores/s",
                            "ms_per_step": sustained_ms,
                            "sample": "%d batches evenly spaced over the %d-batch search sequence of a rank (untimed "
                                      "extra; the timed steps are the first batches of that sequence)" % (nsample, shard_batches)}
        out["top_entries"] = int(len(entries[0]))
        if world == 1 and args.cpu_rotations > 0:
            out["cpu_baseline"] = cpu_baseline(rec, lig, recf, ligf, [w.cpu() for w in W], R_all.numpy(), thr, K,
                                               args.cpu_rotations, V_first)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
