#!/bin/bash
# round 4, call j: split-bf16 convolution with A and B fragments one tap group ahead; rows per wave 4 (default) / 2 / 8
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in default convrw2 convrw8 default; do
  if [ "$v" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$v.so; fi
  echo == $v; timeout 600 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/  torch.*//' 
done
