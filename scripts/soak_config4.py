"""BASELINE config 4 in the shape this build can run it: DockingBenchmark-style unbound pairs END TO END through the
reference driver's own calls (scripts/replay_local_test.py = local_test.py:44-71: get_benchmark_stream, select_model,
GlobalDockingModel(...).cuda().load, Docker(angle_inc=6, box_size=80, resolution=1.25, max_conf=2000,
randomize_rot=True), new_log, dockSE3(rec, lig, batch_size=2)) on synthetic protein-sized two-chain targets (no
benchmark files and no trained weights exist here), the complete 6-degree set (68,760 rotations), twice: the .dat files
must be byte-identical.  Prints one JSON record; --out writes it.
    soak_config4.py [--angle_inc 6] [--targets 2] [--out file.json]"""
import argparse, hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_replay_local_test import make_benchmark

ap = argparse.ArgumentParser()
ap.add_argument("--angle_inc", type=int, default=6)
ap.add_argument("--targets", type=int, default=2)
ap.add_argument("--out", default=None)
args = ap.parse_args()
root = tempfile.mkdtemp(prefix="dlpd_cfg4_")
sizes = [("1SYN", 230, 120, 21), ("2SYN", 310, 95, 33), ("3SYN", 180, 150, 45)][:args.targets]
make_benchmark(root, targets=tuple(sizes))
runs = []
for tag in ("A", "B"):
    env = dict(os.environ)
    env.update({"DLPD_DATA_DIR": os.path.join(root, "data"), "DLPD_MODELS_DIR": os.path.join(root, "models"),
                "DLPD_LOG_DIR": os.path.join(root, "log" + tag), "DLPD_ALLOW_GENERATED_ROTATIONS": "1", "PYTHONDONTWRITEBYTECODE": "1"})
    os.makedirs(os.path.join(env["DLPD_LOG_DIR"], "LocalDebugSE3"), exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "replay_local_test.py"), "-angle_inc", str(args.angle_inc), "-seed", "7",
           "-init_weights", "1", "-report", "1", "-threshold_clash", "40.0", "-rewrite", "1", "-end", str(args.targets)]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rep = json.loads([l for l in out.stdout.splitlines() if l.startswith("REPLAY ")][-1][len("REPLAY "):])
    tg = []
    for t in rep["targets"]:
        dat = os.path.join(rep["test_dir"], t["target"] + ".dat")
        tg.append({k: t[k] for k in ("target", "rotations", "path", "launch_batch", "poses", "rot_per_s", "seconds") if k in t})
        tg[-1]["dat_sha256"] = hashlib.sha256(open(dat, "rb").read()).hexdigest()
        tg[-1]["pose_scores_per_s"] = t["rot_per_s"] * 160 ** 3
    runs.append(tg)
    print("run %s: %s" % (tag, ", ".join("%.0f rot/s" % t["rot_per_s"] for t in tg)), file=sys.stderr)
same = [a["dat_sha256"] for a in runs[0]] == [b["dat_sha256"] for b in runs[1]]
rec = {"what": "config 4 shape: synthetic unbound pairs end to end through the reference driver's calls (dockSE3, SE3MultiResReprScalar "
               "stand-in [16 @ 80^3, 32 @ 40^3], box 80, K = 2000, randomize_rot), %d-degree set; rot/s includes PDB parsing, "
               "typing, the representation, the receptor spectrum and the per-batch clash re-projection" % args.angle_inc,
       "targets": [{"name": s[0], "receptor_residues": s[1], "ligand_residues": s[2]} for s in sizes], "runs": runs, "identical_dat_files": same}
print(json.dumps(rec))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)
assert same, ".dat files differ between the two runs"
