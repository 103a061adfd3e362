#!/bin/bash
# round 4, call h: the split-bf16 convolution (three bf16 terms, six products) against the exact-f32 kernel and torch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_atoms.py -x -q -m gpu -k "conv3d or plugin or dockE3 or E3" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python scripts/conv_bench.py > $OUT/conv_bench.txt 2>&1; tail -7 $OUT/conv_bench.txt
