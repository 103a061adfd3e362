#!/bin/bash
# round 4, call zp: K3<128> with two pencil buffers re-measured after the wait-placement fix (k3p128; + two raw buffers k3p128r2;
# two raw buffers alone k3r2)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/gpu_ab_now.sh r04_zp 40 --workload config2 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3p128 k3p128r2 k3r2
