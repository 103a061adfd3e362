#!/bin/bash
# round 4, call s: what bounds K3<160> on 16-row tiles: diagnostic builds (1 no FMAs, 2 no transforms, 3 neither, 4 no
# pre-activation loads) and the stamps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_s_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- k3ty16 k3ty16d1 k3ty16d2 k3ty16d3 k3ty16d4 | grep -v "_b "
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3ty16_stamps.so python scripts/stamps_k3r.py $wl 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s_$wl/stamps.txt
done
