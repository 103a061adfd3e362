#!/bin/bash
# round 4, call zh: K3<160> at hidden widths 32 / 48 (two voxels per filter thread, 8-row tiles) with two pencil buffers (k3w2)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3w2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "hidden or wide" 2>&1 | tail -1
for h in 48 32; do
  bash scripts/gpu_ab_now.sh r04_zh_h$h 40 --workload real --hidden $h --no_pmc --gather_rotations 0 --strong_s 0 -- default k3w2
done
