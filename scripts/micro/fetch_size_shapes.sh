#!/bin/bash
# FETCH_SIZE per byte read for K2's load shapes (scripts/micro/fetch_size_shapes.hip) -> stdout
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/fetch_shapes}
case $OUT in /*) ;; *) OUT=$ROOT/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/raw -- $ROOT/scripts/micro/fetch_size_shapes > $OUT/run.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/raw2 -- $ROOT/scripts/micro/fetch_size_shapes >> $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
BYTES = 4 << 30
for sub in ("raw", "raw2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        print(sub, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        for c, v in cs.items():
            m = sum(v) / len(v)
            extra = "  (FETCH_SIZE is in KiB units here -> x1024 = %.3f of the bytes read)" % (m * 1024 / BYTES) if c == "FETCH_SIZE" and m < BYTES / 100 else ""
            print("%-16s %-24s mean %.6g per launch = %.4f of the %d bytes read%s" % (k, c, m, m / BYTES, BYTES, extra))
PY
rm -rf $OUT/raw $OUT/raw2
