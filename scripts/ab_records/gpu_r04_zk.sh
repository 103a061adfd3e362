#!/bin/bash
# round 4, call zk: batch split of the persistent K2 blocks: coarse grid (k_xy_corr_s4<80>, 1,312 slabs x 2 blocks per CU) in 3 / 4
# parts (s4n3 / s4n4; default 2: 2,624 blocks = 5.1 rounds of 512), fine grid in 3 (q4n3)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/gpu_ab_now.sh r04_zk_real 40 --workload real --no_pmc --gather_rotations 0 --strong_s 0 -- default s4n3 s4n4 q4n3
