#!/bin/bash
# round 4, call e: the whole GPU suite, the default bench line with its new objects, hidden width 32 on 4- vs 2-voxel filter waves
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_e
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v["ms_per_launch"], 3) for k, v in d["stages"].items()})
for k in ("sustained", "gather_check", "strong", "real_shapes", "c48l80", "e3"):
    v = d.get(k)
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str)) or a in ("list_sha256",)} if v else None)
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:80])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity"))
PY
for wl in config2 real; do
  bash scripts/gpu_ab_now.sh r04_e_h32_$wl 40 --workload $wl --hidden 32 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3wide24
done
