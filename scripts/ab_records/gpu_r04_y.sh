#!/bin/bash
# round 4, call y: K3<160>: priorities for the younger filter waves (1: thirds 0/1/2, 2: upper half 1, 3: all filter waves 1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_y_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3fprio1 k3fprio2 k3fprio3
done
