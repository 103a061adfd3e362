"""BASELINE config 5's driver at production size on one GPU: deeplocalproteindocking_amd/local_test.py over three synthetic
protein-sized targets, box 80, the reference's real shapes, the complete 6-degree set per target -- once with and once
without the next target prepared during the search; targets per second and what the preparation costs / hides.
usage: soak_sweep.py --out file.json [--angle_inc 6]"""
import argparse, hashlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_replay_local_test import make_benchmark, _sweep

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--angle_inc", type=int, default=6)
args = ap.parse_args()
root = tempfile.mkdtemp(prefix="dlpd_soak_sweep_")
make_benchmark(root, targets=(("1SYN", 230, 120, 21), ("2SYN", 310, 95, 41), ("3SYN", 180, 140, 61)))
out = {"what": "local_test.py sweep, three synthetic targets, box 80, [16 @ 80^3, 32 @ 40^3], K = 2000, %d-degree set, one GPU" % args.angle_inc, "runs": []}
hashes = {}
for tag, pre in (("plain", 0), ("prepared_ahead", 1)):
    rep, _ = _sweep(root, "log_" + tag, 1, ["-rewrite", "1", "-prefetch", str(pre), "-angle_inc", str(args.angle_inc)])
    h = {t["target"]: hashlib.sha256(open(os.path.join(rep["test_dir"], t["target"] + ".dat"), "rb").read()).hexdigest() for t in rep["targets"]}
    hashes[tag] = h
    out["runs"].append({"mode": tag, "targets_per_s": rep["targets_per_s"], "seconds": rep["seconds"],
                        "per_target": [{k: t[k] for k in ("target", "seconds", "preparation_s", "waited_for_preparation_s", "prepared_ahead", "rot_per_s")}
                                       for t in rep["targets"]], "preparation_s_behind_a_search": rep["preparation_s_behind_a_search"], "dat_sha256": h})
out["identical_dat_files"] = hashes["plain"] == hashes["prepared_ahead"]
json.dump(out, open(args.out, "w"))
print(json.dumps(out)[:1500])
