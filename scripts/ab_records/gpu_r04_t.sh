#!/bin/bash
# round 4, call t: K3<160> with two pencil buffers on 16-row tiles (default) against one buffer (k3pb1) and round-3 shape (k3old)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -5
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_t_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3pb1 k3old
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3r_stamps.so python scripts/stamps_k3r.py $wl 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_t_$wl/stamps.txt
done
