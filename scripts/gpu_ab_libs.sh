#!/bin/bash
# A/B of variant libraries on the GPU box: scripts/gpu_ab_libs.sh "<bench args>" name1 name2 ... ("default" = the product library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
BARGS=$1; shift
for v in "$@"; do
  if [ "$v" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$v.so; fi
  bash scripts/gpu_ab.sh ab_$v $BARGS
done
