#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_e
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_atoms.py tests/test_replay_local_test.py tests/test_gpu_parity.py -q -m gpu -x -k "overlapped_plugin or rank_aware or e3_plugin or bench_multi_rank or no_kernel or never_wrote or k1_role" --durations=8 -s > $OUT/pytest_new.log 2>&1; grep -v "Adding random\|\[\[\|^  *\[" $OUT/pytest_new.log | tail -25
python scripts/sweep_diff_probe.py 12 "" 2>&1 | grep -v "targets/s" | tail -6
