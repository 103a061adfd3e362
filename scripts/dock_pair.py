"""Dock one receptor/ligand PDB pair the way the reference's local_test.py does (local_test.py:44-71),
without the benchmark table: representation -> exhaustive rotation x translation search -> <out>.dat.

    python scripts/dock_pair.py receptor.pdb ligand.pdb -out pair.dat [-group SE3|E3] [-angle_inc 15]
        [-experiment DIR -load_epoch N]    # optional trained weights (GlobalDockingModel.load)

Multi-GPU: launch with torch.distributed.run; rotations are sharded over the ranks, rank 0 writes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("receptor")
    ap.add_argument("ligand")
    ap.add_argument("-out", default="pair.dat")
    ap.add_argument("-group", default="SE3")
    ap.add_argument("-model", default=None)
    ap.add_argument("-angle_inc", default=15, type=int)
    ap.add_argument("-threshold_clash", default=300.0, type=float)
    ap.add_argument("-box_size", default=80, type=int)
    ap.add_argument("-resolution", default=1.25, type=float)
    ap.add_argument("-max_conf", default=2000, type=int)
    ap.add_argument("-batch_size", default=16, type=int)
    ap.add_argument("-experiment", default=None)
    ap.add_argument("-load_epoch", default=0, type=int)
    args = ap.parse_args()
    entry.build()
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import (E3MultiResRepr4x4, GlobalDockingModel, SE3MultiResReprScalar,
                                                    SimpleFilter)
    from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    repr_ = (SE3MultiResReprScalar if args.group == "SE3" else E3MultiResRepr4x4)(multiplier=8)
    model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=args.threshold_clash)
    if args.experiment:
        model.load(args.experiment, args.load_epoch)
    model = model.to(dev)
    docker = Docker(model, angle_inc=args.angle_inc, box_size=args.box_size, resolution=args.resolution,
                    max_conf=args.max_conf, device=dev, coords_backend=CoordsBackend(), rank=rank, world_size=world)
    if rank == 0:
        docker.new_log(args.out, rewrite=True)
    with torch.no_grad():
        (docker.dockSE3 if args.group == "SE3" else docker.dockE3)(args.receptor, args.ligand, args.batch_size)
    docker.cleanup()
    if rank == 0:
        print("wrote", args.out, "(%d poses, best score %f)" % (len(docker.top_list), docker.top_list[0][4]))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
