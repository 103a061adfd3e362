#!/bin/bash
# round 4, call zo: K3's filter waves with nothing pending on entry to the channel blocks (default: the compiler's wait for the
# NEXT channel's operands sits behind the current channel's 48 packed FMAs) against the wait in front of them (k3early,
# built from the previous source)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role or real or two_res or hidden or fused or pipeline" 2>&1 | tail -1
for wl in config2 real c48l80; do
  bash scripts/gpu_ab_now.sh r04_zo_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3early
done
