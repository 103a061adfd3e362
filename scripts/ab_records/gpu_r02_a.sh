#!/bin/bash
# round-2 first GPU pass: the whole -m gpu suite, the default bench line, kernel stats
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r02_a
python -m pytest tests -m gpu -x -q --durations=12 -s 2>&1 | tail -60 > gpurun_out/r02_a/pytest.log
tail -5 gpurun_out/r02_a/pytest.log
python bench.py > gpurun_out/r02_a/bench.json 2> gpurun_out/r02_a/bench.err
tail -c 1500 gpurun_out/r02_a/bench.json
tail -5 gpurun_out/r02_a/bench.err
