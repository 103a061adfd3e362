#!/bin/bash
# round 4, call zc: K2 q4<160> with the radix-4 stage of its 80-point transforms across the lanes of a quad (default) against
# the two-pass in-place transforms (k2qold)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -3
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_zc_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k2qold
done
