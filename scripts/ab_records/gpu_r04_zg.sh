#!/bin/bash
# round 4, call zg: K1<160> with 32 two-row pencils per block (8 rows x 8 channels: k1y8c8; 4 rows x 16 channels: k1y4c16 --
# 44 KB, three blocks per CU instead of one) and 16 rows x 8 channels (k1y16c8, 64 pencils, re-measured)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in k1y8c8 k1y4c16; do
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role or real or two_res or packed or orientation or pipeline" 2>&1 | tail -1
done
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_zg_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k1y8c8 k1y4c16 k1y16c8
done
