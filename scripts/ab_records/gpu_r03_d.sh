#!/bin/bash
# round 3: the flag-paced ring formulation of the role-split K3 (form 3) against the barrier-paced one (form 2)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "role_split" > $OUT/pytest.log 2>&1
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
run c2_form2 default --k3_form 2
run c2_form3 default --k3_form 3
run c2_form3_f6 k3g_f6 --k3_form 3
run c2_form3_f4 k3g_f4 --k3_form 3
run real_form2 default --workload real --k3_form 2
run real_form3 default --workload real --k3_form 3
run real_form3_5_10 k3g_160_5_10 --workload real --k3_form 3
run real_form3_5_5 k3g_160_5_5 --workload real --k3_form 3
run c48l80_form2 default --workload c48l80 --steps 30 --k3_form 2
run c48l80_form3 default --workload c48l80 --steps 30 --k3_form 3
