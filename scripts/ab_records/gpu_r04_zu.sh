#!/bin/bash
# round 4, call zu: K3 decoding its block's plane / x' / rotation once (default) against a division chain per step (k3prev)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role or real or two_res or hidden or fused or pipeline" 2>&1 | tail -1
for wl in config2 real; do
  bash scripts/gpu_ab_now.sh r04_zu_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3prev
done
