#!/bin/bash
# round 4, call zy: K3<128> transform waves meeting the previous step's "pencils free" barrier behind this step's raw reads
# (default) against in front of them (k3prev, built from the previous source)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "k3_role or hidden or fused or pipeline" 2>&1 | tail -1
bash scripts/gpu_ab_now.sh r04_zy_config2 40 --workload config2 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3prev
