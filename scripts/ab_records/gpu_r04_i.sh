#!/bin/bash
# round 4, call i: split-bf16 convolution, rows per wave 1 / 2 / 4 (16 / 8 / 4 waves per block)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_i
mkdir -p $OUT
for v in default convrw4 convrw1 default convrw4; do
  if [ "$v" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$v.so; fi
  echo == $v; timeout 600 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/  torch.*//' 
done
