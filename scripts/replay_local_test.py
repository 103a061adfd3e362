"""Replays the reference's test driver, call for call, on this build.

/root/reference/src/local_test.py cannot travel to the GPU box, so this script issues the same
sequence of imports and calls as local_test.py:5-14 and :44-71 -- with the reference's own names,
import paths, argument lists and defaults -- against the build's same-named packages:

    from Docker import Docker ; from Dataset import get_benchmark_stream
    from Models import GlobalDockingModel, SimpleFilter, E3MultiResRepr4x4, SE3MultiResReprScalar
    from src import LOG_DIR, MODELS_DIR, DATA_DIR ; from local_train import select_model
    get_benchmark_stream(data_dir, struct_folder='Matched', subset=..., debug=False)
    select_model(args) ; GlobalDockingModel(representation=, filter=, normalize=False,
        rotate_ligand=False, exclude_clashes=True, threshold_clash=).cuda() ; .load(MDL_DIR, epoch=)
    Docker(docking_model=, angle_inc=, box_size=80, resolution=1.25, max_conf=2000, randomize_rot=True)
    docker.new_log(<TEST_DIR>/<pdb>.dat, rewrite=) ; docker.dockSE3 / dockE3(rec, lig, batch_size=2)

Nothing build-specific is passed to any of them.  The directories come from the environment
(DLPD_DATA_DIR / DLPD_MODELS_DIR / DLPD_LOG_DIR, see deeplocalproteindocking_amd/__init__.py); the
rotation files from DLPD_ROTATIONS_DIR (or DLPD_ALLOW_GENERATED_ROTATIONS=1).

Extras that the reference driver does not have (all optional): ``-seed`` fixes the random receptor
rotation, ``-init_weights 1`` writes a randomly initialised checkpoint first when none exists (there
are no trained weights in the reference tree), ``-report`` prints one JSON line with rotations/s, the random receptor rotation and the two file names.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "deeplocalproteindocking_amd")
for p in (ROOT, PKG):                      # what INTEGRATION.md section 1 asks a user to put on sys.path
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()

from Docker import Docker  # noqa: E402
from Dataset import get_benchmark_stream  # noqa: E402
from Models import GlobalDockingModel, SimpleFilter, E3MultiResRepr4x4, SE3MultiResReprScalar  # noqa: E402,F401
from src import LOG_DIR, MODELS_DIR, DATA_DIR  # noqa: E402
from local_train import select_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="replay of local_test.py on the MI355X build")
    ap.add_argument("-experiment", default="LocalDebugSE3")
    ap.add_argument("-dataset", default="DebugDockingBenchmark:Table.csv")
    ap.add_argument("-angle_inc", default=15, type=int)
    ap.add_argument("-threshold_clash", default=300.0, type=float)
    ap.add_argument("-group", default="SE3", type=str)
    ap.add_argument("-model", default="SE3MultiResReprScalar", type=str)
    ap.add_argument("-filter", default="SimpleFilter", type=str)
    ap.add_argument("-load_epoch", default=299, type=int)
    ap.add_argument("-start", default=0, type=int)
    ap.add_argument("-end", default=1, type=int)
    ap.add_argument("-rewrite", default=0, type=int)
    ap.add_argument("-seed", default=None, type=int)
    ap.add_argument("-init_weights", default=0, type=int)
    ap.add_argument("-report", default=0, type=int)
    args = ap.parse_args()

    dataset_name, subset_name = args.dataset.split(":")[:2]
    exp_dir = os.path.join(LOG_DIR, args.experiment)
    mdl_dir = os.path.join(MODELS_DIR, args.experiment)
    test_dir = os.path.join(exp_dir, dataset_name + "_%d" % args.angle_inc + "%.1f" % args.threshold_clash)
    data_dir = os.path.join(DATA_DIR, dataset_name)
    os.makedirs(test_dir, exist_ok=True)

    torch.cuda.set_device(0)
    if args.seed is not None:
        torch.manual_seed(args.seed)
        os.environ["DLPD_ROTATION_SEED"] = str(args.seed)     # Docker draws the receptor rotation from a generator of its own

    stream_test = get_benchmark_stream(data_dir, struct_folder="Matched", subset=subset_name, debug=False)
    protein_model, conformations_filter = select_model(args)
    docking_model = GlobalDockingModel(representation=protein_model, filter=conformations_filter,
                                       normalize=False, rotate_ligand=False, exclude_clashes=True,
                                       threshold_clash=args.threshold_clash).cuda()
    if args.init_weights and not os.path.exists(os.path.join(mdl_dir, "DPD_Model_filter_epoch%d.th" % args.load_epoch)):
        os.makedirs(mdl_dir, exist_ok=True)
        docking_model.save(mdl_dir, epoch=args.load_epoch)
    docking_model.load(mdl_dir, epoch=args.load_epoch)

    docker = Docker(docking_model=docking_model, angle_inc=args.angle_inc, box_size=80, resolution=1.25,
                    max_conf=2000, randomize_rot=True)

    report = []
    for n, data in enumerate(stream_test):
        if not (args.start <= n < args.end):
            continue
        pdb_name, native_path, ureceptor, uligand, breceptor, bligand, cplx = data
        pdb_name, rec_path, lig_path = pdb_name[0], ureceptor[0], uligand[0]
        if docker.new_log(os.path.join(test_dir, "%s.dat" % pdb_name), rewrite=bool(args.rewrite)):
            print("Processing", pdb_name)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                if args.group == "E3":
                    docker.dockE3(rec_path, lig_path, batch_size=2)
                elif args.group == "SE3":
                    docker.dockSE3(rec_path, lig_path, batch_size=2)
                else:
                    raise Exception("Unknown equivariance group", args.group)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            report.append({"target": pdb_name, "seconds": dt, "rotations": int(docker.rot.R.shape[0]),
                           "rot_per_s": docker.rot.R.shape[0] / dt, "launch_batch": docker.launch_batch,
                           "path": getattr(docker, "path", None), "poses": len(docker.top_list),
                           "randR": docker.randR.reshape(3, 3).tolist(), "receptor": rec_path, "ligand": lig_path})
        else:
            print("Skipping", pdb_name)
    docker.cleanup()
    if args.report:
        print("REPLAY " + json.dumps({"test_dir": test_dir, "targets": report}), flush=True)


if __name__ == "__main__":
    main()
