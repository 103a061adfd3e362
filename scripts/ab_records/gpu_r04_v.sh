#!/bin/bash
# round 4, call v: pre-activation touches without the non-temporal hint (k3touchkeep) against none (default); stamps of every wave
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/gpu_ab_now.sh r04_v_real 40 --workload real --no_pmc --gather_rotations 0 --strong_s 0 -- default k3touchkeep
for wl in c48l80 real; do
  for w in 0 1 2 3 4 5 6 7 8 9; do
    echo "== stamps wave $w ($wl)"
    DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3st$w.so python scripts/stamps_k3r.py $wl 2>&1 | grep -v amdgpu.ids
  done | tee gpurun_out/r04_v_real/stamps_all_$wl.txt
done
