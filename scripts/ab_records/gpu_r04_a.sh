#!/bin/bash
# round 4, call a: parity of the channels-last pre-activation planes + A/B on the real shapes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "role_split or real or multires or two_res or candidate" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
run() {
  local tag=$1; shift
  timeout 300 python bench.py --steps 60 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --no_pmc "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json"))
    print("%-22s ms/step %.3f | " % ("$tag", d["ms_per_step"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-600:])
PY
}
run warm --workload real
for rep in a b c; do
  run real_planes_$rep --workload real --preact_layout planes
  run real_cl_$rep --workload real --preact_layout channels_last
done
run config2_a --workload config2
run c48l80_a --workload c48l80
