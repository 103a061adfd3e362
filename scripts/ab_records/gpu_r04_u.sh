#!/bin/bash
# round 4, call u: K3<160> (two pencil buffers, 16-row tiles): pre-activation touches by the transform waves (default) against
# none (k3notouch); transform-wave priority 1 / 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "real or fine or coarse or two_res or multires or k3 or hidden or form" 2>&1 | tail -3
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_u_$wl 40 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3notouch k3prio1 k3prio3
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k3r_stamps.so python scripts/stamps_k3r.py $wl 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_u_$wl/stamps.txt
done
