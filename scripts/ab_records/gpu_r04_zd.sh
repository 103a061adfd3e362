#!/bin/bash
# round 4, call zd: K2 q4<160> holding 4 / 5 of H_0's five pairs in registers (k2qh4 / k2qh5; default 3, the rest parked in LDS)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k2qh5.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "packed or real or two_res" 2>&1 | tail -2
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_zd_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k2qh4 k2qh5
done
