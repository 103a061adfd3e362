#!/bin/bash
# refresh of the judged evidence on the GPU box: bench lines of every workload + rocprofv3 stats / PMC of the default one
# usage: scripts/gpu_refresh.sh <tag>   -> gpurun_out/<tag>/bench_*.json, gpurun_out/prof_<tag>/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
cd $ROOT
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
for w in real c48l80 config1; do
  python bench.py --workload $w --cpu_rotations 0 > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.err
done
bash scripts/profile_gpu.sh $TAG > gpurun_out/$TAG/profile.log 2>&1
tail -30 gpurun_out/$TAG/profile.log
