#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_c
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "no_kernel_writes or k1_role_split or never_wrote" --durations=5 > $OUT/pytest_guard.log 2>&1; tail -25 $OUT/pytest_guard.log
