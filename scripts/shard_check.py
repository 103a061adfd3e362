"""Rotation-sharded search over RCCL vs the single-process search; launched by
tests/test_gpu_parity.py::test_two_rank_nccl_search_equals_single_process with torch.distributed.run,
one rank per GPU.  Rank 0 writes both ranked lists as JSON."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--nrot", type=int, default=70)
    ap.add_argument("--same_device", type=int, default=0, help="all ranks on cuda:0 (one-GPU box; gloo backend only)")
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import bench
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SyntheticRepr
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", 0 if args.same_device else local)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)
    C, L, K = 48, 64, 2000
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    ang = np.random.RandomState(17).uniform(-np.pi, np.pi, size=(args.nrot, 3))
    from deeplocalproteindocking_amd.Utils.Rotations import euler_to_matrices
    R = euler_to_matrices(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])
    model = GlobalDockingModel(SyntheticRepr((C,)), filt, threshold_clash=thr).to(dev)
    dk = Docker(model, box_size=L, max_conf=K, rotations=R, device=dev, rank=rank, world_size=world)
    sharded = dk.dock_volumes([rec], [lig], recf, ligf, write=False)
    if rank == 0:
        single = Docker(model, box_size=L, max_conf=K, rotations=R, device=dev).dock_volumes(
            [rec], [lig], recf, ligf, write=False)
        json.dump({"world": world, "backend": dist.get_backend(), "K": K, "sharded": [list(t) for t in sharded],
                   "single": [list(t) for t in single]}, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
