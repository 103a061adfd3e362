#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "role_split or scores_match_oracle or candidate_lists or reference_model_shapes" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
run() { # tag, env lib, bench args
  local tag=$1 lib=$2; shift 2
  if [ "$lib" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$lib.so; fi
  timeout 300 python bench.py --steps 60 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json"))
    print("%-22s ms/step %.3f | " % ("$tag", d["ms_per_step"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-600:])
PY
}
run c2_three default
run c2_two k3r_two
run c2_three_b default
run real_three default --workload real
run real_two k3r_two --workload real
