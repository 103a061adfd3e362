#!/bin/bash
# Round-5 evidence run (one gpurun call): the whole GPU suite, the driver's bench line, rocprofv3 stats + PMC passes of the
# timed command at config 2 / the real shapes / 48 ch x 80^3, the convolution benchmark under rocprofv3, the complete 6- and
# 4-degree searches and the config-4-shaped end-to-end soak.  Outputs under gpurun_out/r05_z/ (copied to profiles/r05_z_*).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=${1:-r05_z}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
timeout 600 python bench.py --gpus 2 --backend gloo --same_device --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 4 > $OUT/bench_two_ranks_one_gpu.json 2>/dev/null
timeout 600 python bench.py --workload real --cpu_rotations 0 --no_pmc > $OUT/bench_real.json 2>/dev/null
timeout 600 python bench.py --workload c48l80 --cpu_rotations 0 --no_pmc > $OUT/bench_c48l80.json 2>/dev/null
timeout 600 python bench.py --workload config1 --cpu_rotations 0 --no_pmc > $OUT/bench_config1.json 2>/dev/null
bash scripts/profile_gpu.sh ${TAG} > /dev/null 2>&1; cp gpurun_out/prof_${TAG}/summary.txt $OUT/summary.txt; cp gpurun_out/prof_${TAG}/kernel_stats.csv $OUT/kernel_stats.csv; cp gpurun_out/prof_${TAG}/command.txt $OUT/command.txt
bash scripts/profile_gpu.sh ${TAG}_real --workload real > /dev/null 2>&1; cp gpurun_out/prof_${TAG}_real/summary.txt $OUT/real_shapes_summary.txt; cp gpurun_out/prof_${TAG}_real/kernel_stats.csv $OUT/real_shapes_kernel_stats.csv; cp gpurun_out/prof_${TAG}_real/command.txt $OUT/real_shapes_command.txt
bash scripts/profile_gpu.sh ${TAG}_c48l80 --workload c48l80 > /dev/null 2>&1; cp gpurun_out/prof_${TAG}_c48l80/summary.txt $OUT/c48l80_summary.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/convprof -- python3 $ROOT/scripts/conv_bench.py > $OUT/conv_bench.txt 2>&1; cp $(find $OUT/convprof -name "*kernel_stats.csv" | head -1) $OUT/conv_kernel_stats.csv; rm -rf $OUT/convprof)
timeout 600 python scripts/soak_full_search.py --angle_inc 6 --runs 16,12 --out $OUT/soak_full_search_6deg.json > /dev/null 2>&1
timeout 900 python scripts/soak_full_search.py --angle_inc 4 --runs 16 --out $OUT/soak_full_search_4deg.json > /dev/null 2>&1
timeout 900 python scripts/soak_config4.py --out $OUT/soak_config4.json > $OUT/soak_config4.log 2>&1
head -c 600 $OUT/summary.txt; python - <<PY
import json
for f in ("soak_full_search_6deg", "soak_full_search_4deg", "soak_config4"):
    try:
        d = json.load(open("$OUT/%s.json" % f)); print(f, json.dumps(d)[:400])
    except Exception as e:
        print(f, "FAILED", e)
PY
