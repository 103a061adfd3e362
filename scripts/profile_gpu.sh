#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box; summaries land in gpurun_out/prof_<tag>/
# usage: scripts/profile_gpu.sh <tag> [bench args...]
# (counters in their own runs, --kernel-trace only beside them; the program itself after --, no wrappers)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 12 --warmup 2 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 0 --gather_rotations 0 $@"
echo "python3 bench.py $ARGS" > $OUT/command.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
python3 $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# keep the merged scratch small: raw traces are not needed once summarised
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
cat $OUT/summary.txt
