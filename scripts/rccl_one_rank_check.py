"""RCCL on a one-GPU box: the rotation-sharded search's ONE collective (all_gather_top_entries) executed
on the ``nccl`` backend with a process group of one rank -- init_process_group(device_id), the device-side pack, RCCL's
all_gather and the deterministic merge all run; the merged list must be the rank's own list, entry for entry and bit for
bit in the scores.  Launched by tests/test_gpu_parity.py with torch.distributed.run --nproc-per-node 1."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import bench
    from deeplocalproteindocking_amd.Docker.Docker import all_gather_top_entries
    from deeplocalproteindocking_amd.engine import DockingEngine
    from deeplocalproteindocking_amd.Utils.Rotations import euler_to_matrices
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
    C, L, K = 48, 64, 2000
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    ang = np.random.RandomState(23).uniform(-np.pi, np.pi, size=(32, 3))
    R = euler_to_matrices(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])
    eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=thr, max_conf=K, batch=16, device=dev)
    eng.set_receptor(rec, recf); eng.set_ligand(lig, ligf); eng.reset_top()
    eng.search(R)
    mine = eng.top_entries()
    merged = all_gather_top_entries(mine, K, dist.get_world_size(), None, dev, always=True)
    same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(mine[:2] + mine[3:], merged[:2] + merged[3:]))
    same = same and np.array_equal(np.asarray(mine[2], dtype=np.float32).view(np.uint32),
                                   np.asarray(merged[2], dtype=np.float32).view(np.uint32))
    json.dump({"world": dist.get_world_size(), "backend": dist.get_backend(), "entries": len(merged[0]), "identical": bool(same)},
              open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
