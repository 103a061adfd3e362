#!/bin/bash
# round 5, call A: the new multi-rank tests, then the whole GPU suite, then the default bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_a
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bench_multi_rank or bench_refuses or two_rank" > $OUT/pytest_new.log 2>&1; tail -30 $OUT/pytest_new.log
timeout 2400 python -m pytest tests -q -m gpu --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err; head -c 1500 $OUT/bench_default.json
