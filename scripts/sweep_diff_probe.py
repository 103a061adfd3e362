"""Diagnostic: local_test.py sweep on a synthetic three-target benchmark under (ranks, prefetch) combinations; which .dat files differ?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_replay_local_test import make_benchmark, _sweep
root = tempfile.mkdtemp(prefix="dlpd_sweep_probe_")
make_benchmark(root, targets=(("1SYN", 150, 90, 21), ("2SYN", 120, 100, 41), ("3SYN", 100, 60, 61)))
runs = {}
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# (the DLPD_DEBUG_PREPARE switches of local_test.sweep that the first version of this probe drove -- preparing thread on the caller's
#  stream / synchronised at its end / prepared pairs kept alive -- were removed again once the cause was narrowed: EXPERIMENTS.md R5)
combos = [("w1_plain", 1, 0)] + [("w2_ahead_%d" % i, 2, 1) for i in range(REPS)]
for tag, nproc, pre in combos:
    rep, _ = _sweep(root, "log_" + tag, nproc, ["-rewrite", "1", "-prefetch", str(pre)], port=29700 + len(runs) % 200)
    runs[tag] = {n: open(os.path.join(rep["test_dir"], n + ".dat")).read().splitlines() for n in ("1SYN", "2SYN", "3SYN")}
    print(tag, "targets/s %.2f" % rep["targets_per_s"], [round(t["seconds"], 2) for t in rep["targets"]], flush=True)
base = runs["w1_plain"]
for tag in runs:
    for n in ("1SYN", "2SYN", "3SYN"):
        a, b = base[n], runs[tag][n]
        nd = sum(1 for x, y in zip(a, b) if x != y)
        msg = ""
        if nd:
            i = next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)
            msg = " first differing line %d:\n      %s\n      %s" % (i, a[i], b[i])
        if nd:
            print("%-16s %s differing lines %d%s" % (tag, n, nd, msg))
            sa, sb = set(a), set(b)
            print("      lines only in base: %d, only in this run: %d" % (len(sa - sb), len(sb - sa)))

import collections
bad = collections.Counter()
for tag in runs:
    if any(base[n] != runs[tag][n] for n in base):
        bad[tag.rsplit("_", 1)[0]] += 1
print("runs with a differing file, by mode:", dict(bad), "of", REPS, "each")
