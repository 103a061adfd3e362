#!/bin/bash
# split-bf16 convolution with the input split once per layer: tests, then the layer bench old / new (DMA) / new (through registers)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "conv3d or e3 or E3 or plugin" 2>&1 | tail -3
DLPD_CONV_PRESPLIT=0 python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/unstaged /'
python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/staged-dma /'
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_convnodma.so python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/staged-regs /'
DLPD_CONV_PRESPLIT=0 python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/unstaged /'
python scripts/conv_split_bench.py 2>&1 | grep network | sed 's/^/staged-dma /'
