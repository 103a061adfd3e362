#!/bin/bash
# round 4, call zx: K3<128> with two raw staging buffers (the next group's DMA issued a whole first pass earlier) on the final kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/gpu_ab_now.sh r04_zx_config2 40 --workload config2 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3r2
