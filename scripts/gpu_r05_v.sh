#!/bin/bash
# where the split-bf16 convolution's time goes: diagnostic builds without matrix instructions / staging stores / B reads
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lib in ${LIBS:-default convdiag1 convdiag2 convdiag4 default}; do
  if [ "$lib" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$lib.so; fi
  python scripts/conv_split_bench.py 2>&1 | grep "network"
done
