#!/bin/bash
# round 4, call o: K3<160> with TWO pencil buffers (4 transform + 5 filter waves, one barrier per group, the transform
# waves up to a group ahead) against the single-buffer 5 + 5 blocks (k3pb1); parity against the channel-owning reference first
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_o
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_atoms.py -x -q -m gpu -k "role_split or real or hidden or config5 or candidate or multires or two_res or box or dockSE3" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_o_$wl 60 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3pb1
done
