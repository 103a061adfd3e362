#!/bin/bash
# end-of-round evidence with the final kernels: the whole -m gpu suite, the bench line of every workload,
# rocprofv3 stats + PMC of config 2 and of the reference's real shapes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=${1:-r03_final}
bash scripts/gpu_tests.sh $TAG | tail -25
bash scripts/gpu_refresh.sh $TAG | tail -40
bash scripts/profile_gpu.sh ${TAG}_real --workload real | tail -30
