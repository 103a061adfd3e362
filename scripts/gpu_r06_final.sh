#!/bin/bash
# Round-6 evidence run (one gpurun call): the whole GPU suite, the driver's bench line, the 32-rotation CPU baseline, the other
# workloads, rocprofv3 stats + PMC passes of the timed command, the dockE3 kernel profile, the complete 6- and 4-degree searches,
# the config-4-shaped soak, the co-residency reproducer and the FETCH_SIZE probe.  Outputs under gpurun_out/<tag>/ (copied to profiles/<tag>_*).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=${1:-r06_z}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu --durations=12 -rx > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
timeout 900 python bench.py --cpu_rotations 32 --no_real_shapes --sustained_s 0 --strong_s 0 --no_pmc > $OUT/bench_cpu32.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --backend gloo --same_device --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 4 > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench_two_ranks_one_gpu.err
timeout 600 python bench.py --workload real --cpu_rotations 0 --no_pmc > $OUT/bench_real.json 2>/dev/null
timeout 600 python bench.py --workload real_protein --cpu_rotations 0 --no_pmc > $OUT/bench_real_protein.json 2>/dev/null
timeout 600 python bench.py --workload real_protein --k1_occupancy off --cpu_rotations 0 --no_pmc > $OUT/bench_real_protein_k1_occupancy_off.json 2>/dev/null
timeout 600 python bench.py --workload c48l80 --cpu_rotations 0 --no_pmc > $OUT/bench_c48l80.json 2>/dev/null
timeout 600 python bench.py --workload config1 --cpu_rotations 0 --no_pmc > $OUT/bench_config1.json 2>/dev/null
bash scripts/profile_gpu.sh ${TAG} > /dev/null 2>&1; cp gpurun_out/prof_${TAG}/summary.txt $OUT/summary.txt; cp gpurun_out/prof_${TAG}/kernel_stats.csv $OUT/kernel_stats.csv; cp gpurun_out/prof_${TAG}/command.txt $OUT/command.txt
bash scripts/profile_gpu.sh ${TAG}_real --workload real > /dev/null 2>&1; cp gpurun_out/prof_${TAG}_real/summary.txt $OUT/real_shapes_summary.txt; cp gpurun_out/prof_${TAG}_real/kernel_stats.csv $OUT/real_shapes_kernel_stats.csv
bash scripts/profile_gpu.sh ${TAG}_rp --workload real_protein > /dev/null 2>&1; cp gpurun_out/prof_${TAG}_rp/summary.txt $OUT/real_protein_summary.txt; cp gpurun_out/prof_${TAG}_rp/kernel_stats.csv $OUT/real_protein_kernel_stats.csv
bash scripts/gpu_r06_e3prof.sh ${TAG} > /dev/null 2>&1
(timeout 120 scripts/micro/coresidency_repro 300 0; timeout 200 scripts/micro/coresidency_repro 300 1) > $OUT/coresidency_repro.log 2>&1
bash scripts/micro/fetch_size_shapes.sh $OUT/fetch_shapes > $OUT/fetch_size_shapes.txt 2>&1; rm -rf $OUT/fetch_shapes
timeout 600 python scripts/soak_full_search.py --angle_inc 6 --runs 32,16,12 --out $OUT/soak_full_search_6deg.json > /dev/null 2>&1
timeout 900 python scripts/soak_full_search.py --angle_inc 4 --runs 32 --out $OUT/soak_full_search_4deg.json > /dev/null 2>&1
timeout 900 python scripts/soak_config4.py --out $OUT/soak_config4.json > $OUT/soak_config4.log 2>&1
head -c 700 $OUT/summary.txt; python - <<PY
import json
for f in ("soak_full_search_6deg", "soak_full_search_4deg", "soak_config4"):
    try:
        d = json.load(open("$OUT/%s.json" % f)); print(f, json.dumps(d)[:400])
    except Exception as e:
        print(f, "FAILED", e)
for f in ("bench_default", "bench_cpu32", "bench_real", "bench_real_protein", "bench_real_protein_k1_occupancy_off", "bench_c48l80", "bench_config1", "bench_two_ranks_one_gpu"):
    try:
        d = json.loads([l for l in open("$OUT/%s.json" % f) if l.startswith("{")][-1]); print(f, round(d["ms_per_step"], 3), "%.3e" % d["value"], {k: round(v["ms_per_launch"], 3) for k, v in d["stages"].items()})
    except Exception as e:
        print(f, "FAILED", e)
d = json.load(open("$OUT/bench_default.json"))
print("roofline", json.dumps(d["roofline"])[:900])
print("e3", json.dumps(d.get("e3"))[:700])
print("real_protein", json.dumps(d.get("real_protein"))[:600])
print("cpu_baseline", json.dumps(json.load(open("$OUT/bench_cpu32.json")).get("cpu_baseline"))[:500])
PY
