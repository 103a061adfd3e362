#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:" ; }
echo "== co-runner bf16x3 convolution, no B-fragment LDS reads, PSEUDO-RANDOM finite operand bits in registers"
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_convdiag12.so python scripts/search_race_probe.py 50 conv1bf16 2>&1 | f
echo "== the same with CONSTANT operands"
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_convdiag4.so python scripts/search_race_probe.py 50 conv1bf16 2>&1 | f
