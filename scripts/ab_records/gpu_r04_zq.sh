#!/bin/bash
# round 4, call zq: in-kernel stamps of every transform / filter wave of K3<128> at config 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04_zq
for w in 0 1 2 3 4 5 6 7; do
  echo "== stamps wave $w (config2)"
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3st$w.so python scripts/stamps_k3r.py config2 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04_zq/stamps_all_config2.txt
