#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py's dockE3 measurement (scripts/e3_profile.py) -> gpurun_out/<tag>/e3_kernel_stats.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_e3prof}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e3prof -- python3 $ROOT/scripts/e3_profile.py > $OUT/e3_profile.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/e3prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$OUT/e3_kernel_stats.txt", "w") as out:
    for r in rows[:30]:
        line = "%-50s calls %5s  avg %8.1f us  total %8.2f ms  %s%%" % (r["Name"].split("(")[0].replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"])
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/e3prof
tail -1 $OUT/e3_profile.log | cut -c1-900
