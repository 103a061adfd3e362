#!/bin/bash
# round 4, call ze: K3 with an unconditional clamp (no clip = +-infinity; default) against clamp + select (k3select, built from
# the previous source)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in config2 real; do
  bash scripts/gpu_ab_now.sh r04_ze_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3select
done
