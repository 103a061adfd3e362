#!/bin/bash
# one extra PMC pass: how busy the vector ALU is (packed f32 instructions occupy it twice as long as their count says)
# usage: scripts/profile_valu.sh <tag> [bench args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 12 --warmup 2 --cpu_rotations 0 --no_real_shapes --sustained_s 0 $@"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py $ARGS > $OUT/valu.log 2>&1
tail -3 $OUT/valu.log
python3 $ROOT/scripts/summarize_prof.py $OUT > $OUT/valu_summary.txt 2>&1
rm -rf $OUT/sq
cat $OUT/valu_summary.txt
