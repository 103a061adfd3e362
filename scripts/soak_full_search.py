"""Soak: the complete BASELINE config 2 search (48 ch x 64^3, K = 2000) over a whole SOI-sized rotation set, several
times and with two launch batch sizes -- the ranked lists must be bit-identical (no race between the compute stream and
the top-K side stream over thousands of batches).  Prints one JSON record (seconds, rotations/s, sha256 of the list).
    soak_full_search.py [--angle_inc 6] [--runs 16,16,12] [--out file.json]
angle_inc 6: 68,760 rotations (config 2); 4: the 232,020-rotation set configs 3 / 5 name (generated substitute)."""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd.Utils.Rotations import Rotations

ap = argparse.ArgumentParser()
ap.add_argument("--angle_inc", type=int, default=6)
ap.add_argument("--runs", default="16,16,12")
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
C, L, K = 48, 64, 2000
rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
thr = bench.clash_threshold(recf, ligf)
rot = Rotations(args.angle_inc, allow_generated=True, verbose=False)
R = rot.R


def list_hash(entries):
    h = hashlib.sha256()
    for a, dt in zip(entries, (np.int64, np.int64, np.float32, np.int64)):
        h.update(np.ascontiguousarray(a, dtype=dt).tobytes())
    return h.hexdigest()


runs, lists = [], []
for nb in [int(x) for x in args.runs.split(",")]:
    eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=thr, max_conf=K, batch=nb, device=dev)
    eng.set_receptor(rec, recf); eng.set_ligand(lig, ligf); eng.reset_top()
    torch.cuda.synchronize(); t0 = time.time()
    eng.search(R)
    torch.cuda.synchronize(); dt = time.time() - t0
    ent = eng.top_entries()
    lists.append(eng.top_list())
    runs.append({"launch_batch": nb, "seconds": dt, "rot_per_s": R.shape[0] / dt, "pose_scores_per_s": R.shape[0] * (2 * L) ** 3 / dt,
                 "list_sha256": list_hash(ent), "best": lists[-1][0][4], "kth": lists[-1][-1][4]})
    print("batch %2d: %d rotations in %.1f s = %.0f rot/s" % (nb, R.shape[0], dt, R.shape[0] / dt), file=sys.stderr)
    del eng
same = all(l == lists[0] for l in lists)
out = {"what": "complete search, 48 ch x 64^3, K = %d, %d-degree set" % (K, args.angle_inc), "rotations": int(R.shape[0]),
       "rotation_set": "generated SOI-sized substitute" if rot.source == "generated" else os.path.basename(rot.source),
       "runs": runs, "identical_lists": same, "entries": len(lists[0]), "rotations_in_list": len({t[0] for t in lists[0]})}
print(json.dumps(out))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
assert same, "ranked lists differ between runs"
