#!/bin/bash
# round 4, call zl: diagnostic (wrong results): K3<160> without its per-group block barrier -- the upper bound of what any finer
# synchronisation between transform and filter waves could buy
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_zl_$wl 30 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3nobar | grep -v "_b "
done
