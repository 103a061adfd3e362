#!/bin/bash
# round 4, call zt: second-pass twiddles of the N = 128 transform waves in registers (default) against the LDS table (k3prev)
# per instruction (k3prev, built from the previous source)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role or real or two_res or hidden or fused or pipeline" 2>&1 | tail -1
for wl in config2; do
  bash scripts/gpu_ab_now.sh r04_zt_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3prev
done
