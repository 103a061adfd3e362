#!/bin/bash
# after the k_topk_merge tweak: top-K / golden GPU tests, the default bench line and the rocprofv3 passes of the timed command again
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_zz
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -k "topk or top_list or golden or candidate or replay or sweep or radix" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 200 $OUT/bench_default.err
bash scripts/profile_gpu.sh r05_zz > /dev/null 2>&1; cp gpurun_out/prof_r05_zz/summary.txt $OUT/summary.txt; cp gpurun_out/prof_r05_zz/kernel_stats.csv $OUT/kernel_stats.csv; cp gpurun_out/prof_r05_zz/command.txt $OUT/command.txt
head -c 1200 $OUT/summary.txt
python -c "
import json
d = json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][0])
print('bench: %.3f ms/step, value %.3e, hash %s' % (d['ms_per_step'], d['value'], d['gather_check']['list_sha256'][:16]))"
