#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_p_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3diag1 k3diag2 k3diag3 k3diag4 | grep -v "_b "
done
