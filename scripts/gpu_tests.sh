#!/bin/bash
# the whole -m gpu suite on the GPU box: scripts/gpu_tests.sh <tag> [pytest args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd $ROOT
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -q --durations=15 -s "$@" > gpurun_out/$TAG/pytest.log 2>&1
tail -40 gpurun_out/$TAG/pytest.log
