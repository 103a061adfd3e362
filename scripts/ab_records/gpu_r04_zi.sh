#!/bin/bash
# round 4, call zi: K3<128> at hidden width 48 (two voxels per filter thread, 8-row tiles) with two pencil buffers (k3w128)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3w128.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "hidden or wide" 2>&1 | tail -1
bash scripts/gpu_ab_now.sh r04_zi_h48 40 --workload config2 --hidden 48 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3w128
