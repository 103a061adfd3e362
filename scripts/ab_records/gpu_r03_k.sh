#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "candidate_lists or full_rotation_set or scores_match" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
run() { # tag, bench args
  local tag=$1; shift
  timeout 300 python bench.py --steps 60 --warmup 5 --cpu_rotations 0 --no_real_shapes "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json"))
    print("%-22s ms/step %.3f sustained %.3f | " % ("$tag", d["ms_per_step"], d.get("sustained", {}).get("ms_per_step", 0)) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()), d.get("sustained", {}).get("list_sha256", "")[:12])
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-600:])
PY
}
run c2_prefetch
run c2_inline --no_k1_prefetch
run c2_prefetch_b
run c1_prefetch --workload config1
run c1_inline --workload config1 --no_k1_prefetch
