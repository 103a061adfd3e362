#!/bin/bash
# round 4, call w: VALU / LDS / scalar busy counters of K3<160> (two pencil buffers) on the real shapes and 48 ch x 80^3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04_w; mkdir -p $OUT
for wl in real c48l80; do
  A="--workload $wl --steps 12 --warmup 2 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 0 --gather_rotations 0 --no_extras --no_pmc"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    i=$((i+1))
    mkdir -p $OUT/${wl}_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${wl}_$i/sq -- python3 $ROOT/bench.py $A > $OUT/${wl}_$i.log 2>&1
    python3 $ROOT/scripts/summarize_prof.py $OUT/${wl}_$i 2>/dev/null | grep -i "k_zifft_filter_rs<160\|k_xy_corr_q4\|k_rotate_zfft_cl<160"
    rm -rf $OUT/${wl}_$i
  done
done
