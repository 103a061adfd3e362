#!/bin/bash
# round 3, first GPU call: parity of the role-split K3 + A/B of the two K3 formulations and of variant builds
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "role_split or scores_match_oracle or candidate_lists" > $OUT/pytest.log 2>&1
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
run c2_form1 default --k3_form 1
run c2_form2 default --k3_form 2
run c2_prio2 k3r_prio2
run c2_prio3 k3r_prio3
run c2_tpb2 k3r_tpb2
run c2_raw1 k3r_raw1
run c2_form2_again default --k3_form 2
run real_form1 default --workload real --k3_form 1
run real_form2 default --workload real --k3_form 2
run real_m10 k3r_m10 --workload real
run real_prio2 k3r_prio2 --workload real
run real_tpb2 k3r_tpb2 --workload real
run c48l80_form1 default --workload c48l80 --k3_form 1 --steps 30
run c48l80_form2 default --workload c48l80 --k3_form 2 --steps 30
