#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_e
mkdir -p $OUT
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
run c2_form2 default --k3_form 2
run c2_prio1 k3r_prio1 --k3_form 2
run c2_prio3 k3r_prio3 --k3_form 2
run c2_form2_b default --k3_form 2
run real_prio3 k3r_prio3 --workload real --k3_form 2
run real_form2 default --workload real --k3_form 2
