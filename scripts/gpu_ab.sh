#!/bin/bash
# quick A/B of the default library on the GPU box: stage times of bench.py (no CPU leg, no extras)
# usage: scripts/gpu_ab.sh <tag> [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd $ROOT
mkdir -p gpurun_out/$TAG
python bench.py --steps 60 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json"))
print("$TAG value %.4g ms/step %.3f | " % (d["value"], d["ms_per_step"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()) + " | head %.3f" % d["head_of_set"]["ms_per_step"])
PY
