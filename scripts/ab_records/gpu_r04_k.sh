#!/bin/bash
# round 4, call k: GPU suite + default bench line with the split-bf16 convolution as the plugins' default
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_k
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v["ms_per_launch"], 3) for k, v in d["stages"].items()})
for k in ("gather_check", "strong", "real_shapes", "c48l80", "e3"):
    v = d.get(k)
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str)) or a in ("list_sha256", "conv_precision")} if v else None)
PY
timeout 300 python scripts/conv_bench.py 2>&1 | grep -v amdgpu > $OUT/conv_bench.txt; tail -6 $OUT/conv_bench.txt
