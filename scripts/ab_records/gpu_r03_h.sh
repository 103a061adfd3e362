#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_h
mkdir -p $OUT
run() { # tag, bench args
  local tag=$1; shift
  timeout 300 python bench.py --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json"))
    print("%-22s ms/step %.3f rot/s %.0f | " % ("$tag", d["ms_per_step"], d["rot_per_s"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-600:])
PY
}
run c2_b16 --steps 60 --batch 16
run c2_b24 --steps 40 --batch 24
run c2_b32 --steps 30 --batch 32
run c2_b8 --steps 120 --batch 8
run real_b16 --steps 60 --batch 16 --workload real
run real_b32 --steps 30 --batch 32 --workload real
run real_b8 --steps 120 --batch 8 --workload real
