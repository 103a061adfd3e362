#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/clock_$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $ROOT/bench.py --steps 40 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 "$@" > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections
cc = glob.glob("$OUT/c/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$OUT/c/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k = r["Kernel_Name"].split("(")[0][:40]
    a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += dur[r["Dispatch_Id"]]; a[2] += 1
for k, (c, d, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]:
    print("%-42s n=%d avg %.3f ms  clock %.2f GHz" % (k, n, d / n / 1e6, c / 8 / d))
PY
rm -rf $OUT/c
