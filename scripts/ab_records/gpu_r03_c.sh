#!/bin/bash
# round 3: the 4-lane-pencil K2 at N = 160 against the quad kernel it replaces
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "role_split or scores_match_oracle or config5 or reference_model_shapes or candidate_lists" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
run() { # tag, env lib, bench args
  local tag=$1 lib=$2; shift 2
  if [ "$lib" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$lib.so; fi
  timeout 300 python bench.py --steps 60 --warmup 5 --cpu_rotations 0 --no_real_shapes "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json"))
    print("%-22s ms/step %.3f | " % ("$tag", d["ms_per_step"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-600:])
PY
}
run real_q4 default --workload real
run real_quad k2_quad --workload real
run real_q4_h5 k2q_h5 --workload real
run real_q4_early k2q_early --workload real
run c48l80_q4 default --workload c48l80 --steps 30
run c48l80_quad k2_quad --workload c48l80 --steps 30
run c48l80_q4_h5 k2q_h5 --workload c48l80 --steps 30
