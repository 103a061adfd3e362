#!/bin/bash
# round 4, call zr: K3<128> with five / six transform waves (the stamps say the four transform waves are the busiest role: 79 %)
# -- with two pencil buffers (k3p2f5 / k3p2f6: no spill at 128 registers) and with one (k3f5: 20 spilled registers)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3p2f6.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role" 2>&1 | tail -1
bash scripts/gpu_ab_now.sh r04_zr 40 --workload config2 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3p2f5 k3p2f6 k3f5
