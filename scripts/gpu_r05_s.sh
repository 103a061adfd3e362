#!/bin/bash
# the one-rank RCCL tests (bench.py --force_group, local_test.py -force_group 1) + two ranks on one GPU, 30 sweeps: any differing file left?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "one_rank or rccl" 2>&1 | tail -5
timeout 1500 python scripts/sweep_diff_probe.py 30 2>&1 | tail -12
