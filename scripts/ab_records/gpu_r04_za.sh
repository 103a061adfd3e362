#!/bin/bash
# round 4, call za: K2 q4<160> without its receptor loads (diagnostic: how much do the 64-byte-run gathers of rec cost?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_za_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k2qnorec | grep -v "_b "
done
