#!/bin/bash
# round 4, call b: K3's first layer on the matrix pipe (4x4x1 blocks) vs the vector FMAs (variant k3valu)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "role_split or hidden or real or multires or two_res or candidate or scores_match" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for wl in real config2 c48l80; do
  bash scripts/gpu_ab_now.sh r04_b_$wl 60 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3valu
done
