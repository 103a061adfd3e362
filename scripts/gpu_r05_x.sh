#!/bin/bash
# split-bf16 convolution: XCD-aware tile order on / off, input split per block / once per layer
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "conv3d" 2>&1 | tail -2
for rep in 1 2; do
DLPD_CONV_PRESPLIT=0 python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/xcd unstaged /'
DLPD_CONV_PRESPLIT=0 DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_convnoxcd.so python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/noxcd unstaged /'
python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/xcd staged /'
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_convnoxcd.so python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/noxcd staged /'
done
