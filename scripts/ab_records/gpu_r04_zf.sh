#!/bin/bash
# round 4, call zf: the coarse grid's K3 (N = 80, MODE 2) with two pencil buffers (k3p80)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3p80.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role or real or two_res or packed" 2>&1 | tail -2
bash scripts/gpu_ab_now.sh r04_zf_real 40 --workload real --no_pmc --gather_rotations 0 --strong_s 0 -- default k3p80
