#!/bin/bash
# What the SQ / GRBM counters of rocprofv3 mean on this chip, per dispatch of the bench kernels: raw per-dispatch averages beside the
# dispatch duration from the same CSV's timestamps -- the calibration behind bench.py's roofline.secondary.
# usage: scripts/pmc_units_probe.sh <tag> [bench args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 0 --gather_rotations 0 $@"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -c . $OUT/counters_list.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
tail -3 $OUT/sq.log
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True)
print(fs)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    if "Start_Timestamp" in r:
        dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open("$OUT/units.txt", "w") as f:
    for k in sorted(acc, key=lambda k: -sum(dur[k].values()) if dur[k] else 0)[:8]:
        m = len(n[k]); d = sum(dur[k].values()) / max(1, len(dur[k])) if dur[k] else float("nan")
        line = "%-42s n=%d dur_us=%.1f " % (k, m, d / 1e3) + " ".join("%s=%.4g" % (c, v / m) for c, v in sorted(acc[k].items()))
        print(line); f.write(line + "\n")
PY
grep -i -A3 "SQ_ACTIVE_INST_VALU\b\|SQ_LDS_IDX_ACTIVE\|GRBM_GUI_ACTIVE\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU\b" $OUT/counters_list.txt | head -60 > $OUT/counter_descriptions.txt
rm -rf $OUT/sq
