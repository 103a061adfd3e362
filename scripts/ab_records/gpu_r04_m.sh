#!/bin/bash
# round 4, call m: split-bf16 convolution, A fragments requested one (default) or two (conva2) tap groups ahead
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in default conva2 default conva2; do
  if [ "$v" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$v.so; fi
  echo == $v; timeout 600 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/  torch.*//'
done
