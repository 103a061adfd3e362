#!/bin/bash
# round 4, call r: K3<160> on 16-row tiles (128-byte DMA runs), 5 transform + 10 filter waves
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_r_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3ty16
done
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3ty16.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "real or fine or coarse or two_res or multires or k3 or hidden" 2>&1 | tail -5
