#!/bin/bash
# round 4, call x: K3<128> with two pencil buffers (k3p128; + two raw buffers: k3p128r2; five / six transform waves: k3p128f5 / f6)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3p128.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3 or form or config or fused or hidden" 2>&1 | tail -3
bash scripts/gpu_ab_now.sh r04_x 40 --workload config2 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3p128 k3p128r2 k3p128f5 k3p128f6
