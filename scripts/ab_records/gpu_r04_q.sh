#!/bin/bash
# round 4, call q: LDS-DMA gather microbenchmark (K3's access pattern, nothing computed)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04_q
timeout 600 scripts/micro/dma_gather | tee gpurun_out/r04_q/dma_gather.txt
