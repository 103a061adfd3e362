"""The benchmark sweep of /root/reference/src/local_test.py:16-75 -- same flags, same directories, same calls
(``get_benchmark_stream``, ``select_model``, ``GlobalDockingModel(...).cuda()``, ``.load``, ``Docker(...,
max_conf=2000, randomize_rot=True)``, ``new_log`` / ``dockSE3`` / ``dockE3`` per target, the ``-start`` / ``-end``
window and the ``-rewrite 0`` resume rule) -- made rank-aware for one node of MI355Xs (BASELINE config 5: the
DockingBenchmark sweep, every target's rotation set sharded over the ranks):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        deeplocalproteindocking_amd/local_test.py -experiment ... -dataset DockingBenchmarkV5:Table_BM5.csv \\
        -angle_inc 4 -start 0 -end 230

or plain ``python deeplocalproteindocking_amd/local_test.py ...`` for one GPU.  What is added to the reference's loop:

  * one process per GPU (``LOCAL_RANK``), ``rank`` / ``world_size`` handed to ``Docker``: rank r scores rotations
    r, r + W, ... of EVERY target, one all-gather of the per-rank lists per target;
  * only rank 0 opens, truncates and writes the ``.dat`` files; its ``new_log`` decision (the resume rule of
    Docker.py:63-79) is broadcast, so all ranks skip or process the same targets;
  * the next target is PREPARED while the current one is searched: PDB parsing, typing, projection, representation
    and the receptor spectrum (``Docker.prepare``) run on a host thread of their own, into the second of two engines --
    the host part of the per-target serial term disappears behind the search (its kernels stay in stream order with the
    search's: see ``sweep``);
  * ``-report 1`` prints one JSON line (rank 0): per-target seconds, rotations/s, and ``targets_per_s`` of the sweep.

``sweep()`` only needs an object with the ``Docker`` interface and is what the tests drive (two gloo ranks on the
emulated kernels); ``main()`` is the command line.
"""
import argparse
import json
import os
import sys
import time

import torch


def _collective_device(docker):
    import torch.distributed as dist
    return docker.device if dist.get_backend(docker.process_group) == "nccl" else torch.device("cpu")


def _broadcast_flags(docker, flags, n):
    """rank 0's list of n booleans -> every rank (one broadcast of n int64; identity for a single rank)."""
    if docker.world_size <= 1 and not docker.collectives_with_one_rank:
        return [bool(f) for f in flags]
    import torch.distributed as dist
    group = docker.process_group
    src = dist.get_global_rank(group, 0) if group is not None else 0
    buf = torch.zeros(n, dtype=torch.int64)
    if docker.rank == 0:
        buf[:] = torch.as_tensor([int(bool(f)) for f in flags], dtype=torch.int64)
    buf = buf.to(_collective_device(docker))
    dist.broadcast(buf, src=src, group=group)
    return [bool(v) for v in buf.cpu().tolist()]


def sweep(docker, targets, test_dir, group="SE3", rewrite=False, batch_size=2, prefetch=True, say=print):
    """Dock every target of ``targets`` = [(pdb_name, receptor_path, ligand_path), ...] into ``test_dir/<pdb_name>.dat``
    (local_test.py:57-75), on all ranks of ``docker``'s process group.  -> report dict (the same on every rank except
    for the timings, which are the rank's own)."""
    if group not in ("SE3", "E3"):
        raise Exception("Unknown equivariance group", group)
    n_targets = len(targets)
    paths = [os.path.join(test_dir, "%s.dat" % t[0]) for t in targets]
    # what rank 0 expects to process (the resume rule, read-only): decides what is prepared ahead; the binding
    # decision is new_log's, taken -- and broadcast -- when the target's turn comes
    plan = _broadcast_flags(docker, [bool(rewrite) or not docker.log_is_complete(p) for p in paths]
                            if docker.rank == 0 else [], n_targets)
    dev = docker.device
    on_gpu = dev.type == "cuda"
    pool = None
    if prefetch and n_targets > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dlpd-prepare")
        # The preparing thread enqueues its device work on the CALLER'S stream, in order with the search's launches: what
        # overlaps with the search is the host side (PDB parsing, typing, launch overhead).  A stream of its own would also
        # overlap the projection / representation kernels -- nothing to gain (a target's preparation is 0.01-0.1 s of the 23 s
        # its search takes, profiles/r05_s_soak_sweep.json), and the plugin's matrix-instruction convolution co-resident with
        # the pipeline's LDS kernels is what perturbs the pipeline's low bits on this hardware (EXPERIMENTS.md R5): not offered.
    pending = {}                                      # target index -> Future of its PreparedPair
    pending_slot = {}                                 # target index -> the engine slot that preparation fills
    caller_stream = torch.cuda.current_stream(dev) if on_gpu else None     # (the current stream is a per-thread setting too)

    def prepare_in_background(j, slot):
        pending_slot[j] = slot
        def work():
            if not on_gpu:
                return docker.prepare(targets[j][1], targets[j][2], group, slot=slot)
            torch.cuda.set_device(dev)                # (the device is a per-thread setting)
            with torch.cuda.stream(caller_stream):    # literally the stream the search is enqueued on
                return docker.prepare(targets[j][1], targets[j][2], group, slot=slot)
        pending[j] = pool.submit(work)

    sync = (lambda: torch.cuda.synchronize(dev)) if on_gpu else (lambda: None)
    report, processed, skipped, hidden = [], 0, 0, 0.0
    t_sweep = time.perf_counter()
    try:
        for n, (name, rec_path, lig_path) in enumerate(targets):
            go = _broadcast_flags(docker, [docker.new_log(paths[n], rewrite=bool(rewrite))] if docker.rank == 0 else [], 1)[0]
            ahead = pending.pop(n, None)
            pending_slot.pop(n, None)
            if not go:
                say("Skipping", name)
                skipped += 1
                if ahead is not None:
                    ahead.result()                    # prepared for nothing (finished by someone else meanwhile): drop it
                continue
            say("Processing", name)
            t0 = time.perf_counter()
            # an unprepared target must not take the engine slot a LATER target is being prepared into (possible when the plan
            # expected this one to be skipped and its log changed meanwhile)
            slot = next(k for k in (0, 1) if k not in pending_slot.values()) if pool is not None else 0
            prepared = ahead.result() if ahead is not None else docker.prepare(rec_path, lig_path, group, slot=slot)
            t_ready = time.perf_counter()
            if pool is not None:
                nxt = next((j for j in range(n + 1, n_targets) if plan[j]), None)
                if nxt is not None and nxt not in pending:    # (already under way when an earlier target was expected to be skipped)
                    prepare_in_background(nxt, 1 - prepared.slot)
            with torch.no_grad():
                (docker.dockE3 if group == "E3" else docker.dockSE3)(rec_path, lig_path, batch_size, prepared=prepared)
            sync()
            dt = time.perf_counter() - t0
            nrot = int(docker.rot.R.shape[0])
            if ahead is not None:
                hidden += prepared.seconds
            report.append({"target": name, "seconds": dt, "waited_for_preparation_s": t_ready - t0,
                           "preparation_s": prepared.seconds, "prepared_ahead": ahead is not None,
                           "rotations": nrot, "rot_per_s": nrot / dt, "launch_batch": docker.launch_batch,
                           "path": getattr(docker, "path", None), "poses": len(docker.top_list),
                           "randR": docker.randR.reshape(3, 3).tolist() if docker.randomize_rot else None,
                           "receptor": rec_path, "ligand": lig_path})
            processed += 1
    finally:
        if pool is not None:
            for f in pending.values():
                f.cancel()
            pool.shutdown(wait=True)
        docker.cleanup()
    total = time.perf_counter() - t_sweep
    backend = None
    if docker.world_size > 1 or docker.collectives_with_one_rank:
        import torch.distributed as dist
        backend = dist.get_backend(docker.process_group)
    return {"test_dir": test_dir, "world_size": docker.world_size, "rank": docker.rank, "group": group,
            "collective_backend": backend,
            "processed": processed, "skipped": skipped, "seconds": total,
            "targets_per_s": processed / total if total > 0 else 0.0, "prepared_ahead": pool is not None,
            "preparation_s_behind_a_search": hidden, "targets": report}


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="DockingBenchmark sweep on the MI355X build (local_test.py, rank-aware)")
    # the reference's flags and defaults (local_test.py:17-28)
    ap.add_argument("-experiment", default="LocalDebugSE3", help="Experiment name")
    ap.add_argument("-dataset", default="DebugDockingBenchmark:Table.csv", help="Dataset name")
    ap.add_argument("-angle_inc", default=15, help="Angle increment, int", type=int)
    ap.add_argument("-threshold_clash", default=300.0, help="Clash theshold for excluding conformations", type=float)
    ap.add_argument("-group", default="SE3", help="Equivariance group of the algorithm", type=str)
    ap.add_argument("-model", default="SE3MultiResReprScalar", help="Name of the representation", type=str)
    ap.add_argument("-filter", default="SimpleFilter", help="Name of the filter", type=str)
    ap.add_argument("-load_epoch", default=299, help="Max epoch", type=int)
    ap.add_argument("-start", default=0, help="Starting id", type=int)
    ap.add_argument("-end", default=1, help="Ending id", type=int)
    ap.add_argument("-rewrite", default=0, help="Rewrite previous output", type=int)
    # this build's extras (all optional)
    ap.add_argument("-seed", default=None, type=int, help="fixes the random receptor rotation")
    ap.add_argument("-init_weights", default=0, type=int, help="write a randomly initialised checkpoint first if none exists (with -seed: the same one every time)")
    ap.add_argument("-report", default=0, type=int, help="print one JSON line with timings (rank 0)")
    ap.add_argument("-prefetch", default=1, type=int, help="prepare the next target while the current one is searched")
    ap.add_argument("-backend", default="nccl", choices=("nccl", "gloo"), help="collective backend (nccl = RCCL)")
    ap.add_argument("-same_device", default=0, type=int, help="every rank on cuda:0 (one-GPU box; needs -backend gloo)")
    ap.add_argument("-force_group", default=0, type=int,
                    help="initialise the process group and run every collective with ONE rank too (one-GPU RCCL check)")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):            # what INTEGRATION.md section 1 asks a user to put on sys.path
        if p not in sys.path:
            sys.path.insert(0, p)
    import __graft_entry__ as entry
    entry.build()
    # the reference's import lines (local_test.py:5-14)
    from Docker import Docker
    from Dataset import get_benchmark_stream
    from Models import GlobalDockingModel, SimpleFilter, E3MultiResRepr4x4, SE3MultiResReprScalar  # noqa: F401
    from src import LOG_DIR, MODELS_DIR, DATA_DIR
    from local_train import select_model

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device and args.backend == "nccl" and world > 1:
        raise SystemExit("local_test.py: -same_device needs -backend gloo (RCCL wants one device per rank)")
    dev = torch.device("cuda", 0 if args.same_device else local_rank)
    if args.same_device and world > 1 and rank == 0:
        print("local_test.py: -same_device puts %d ranks on ONE GPU: kernels of different processes share CUs, which on this hardware "
              "can change low mantissa bits of a few scores (inside the 1e-4 parity band; INTEGRATION.md, 'Sharing the GPU') -- a test "
              "mode, not a way to run a sweep" % world, file=sys.stderr, flush=True)
    torch.cuda.set_device(dev)                          # (local_test.py:44 sets device 0)
    if world > 1 or args.force_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)

    dataset_name, subset_name = args.dataset.split(":")[:2]
    exp_dir = os.path.join(LOG_DIR, args.experiment)
    mdl_dir = os.path.join(MODELS_DIR, args.experiment)
    test_dir = os.path.join(exp_dir, dataset_name + "_%d" % args.angle_inc + "%.1f" % args.threshold_clash)
    data_dir = os.path.join(DATA_DIR, dataset_name)
    if rank == 0:
        os.makedirs(test_dir, exist_ok=True)

    stream_test = get_benchmark_stream(data_dir, struct_folder="Matched", subset=subset_name, debug=False)
    if args.init_weights and args.seed is not None:
        torch.manual_seed(args.seed)                    # the checkpoint -init_weights writes is then the same in every run
    protein_model, conformations_filter = select_model(args)
    docking_model = GlobalDockingModel(representation=protein_model, filter=conformations_filter,
                                       normalize=False, rotate_ligand=False, exclude_clashes=True,
                                       threshold_clash=args.threshold_clash).cuda()
    checkpoint = os.path.join(mdl_dir, "DPD_Model_filter_epoch%d.th" % args.load_epoch)
    if args.init_weights and rank == 0 and not os.path.exists(checkpoint):
        os.makedirs(mdl_dir, exist_ok=True)
        docking_model.save(mdl_dir, epoch=args.load_epoch)
    if world > 1 or args.force_group:
        dist.barrier()                                  # the checkpoint (and test_dir) exist before anyone reads them
    docking_model.load(mdl_dir, epoch=args.load_epoch)

    docker = Docker(docking_model=docking_model, angle_inc=args.angle_inc, box_size=80, resolution=1.25,
                    max_conf=2000, randomize_rot=True, device=dev, rank=rank, world_size=world, rotation_seed=args.seed,
                    collectives_with_one_rank=bool(args.force_group))

    targets = []
    for n, data in enumerate(stream_test):
        if not (args.start <= n < args.end):
            continue
        pdb_name, native_path, ureceptor, uligand, breceptor, bligand, cplx = data
        targets.append((pdb_name[0], ureceptor[0], uligand[0]))
    say = print if rank == 0 else (lambda *a, **k: None)
    rep = sweep(docker, targets, test_dir, group=args.group, rewrite=bool(args.rewrite), batch_size=2,
                prefetch=bool(args.prefetch), say=say)
    if args.report and rank == 0:
        print("SWEEP " + json.dumps(rep), flush=True)
    if world > 1 or args.force_group:
        dist.barrier()
        dist.destroy_process_group()
    return rep


if __name__ == "__main__":
    main()
