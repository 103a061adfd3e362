#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:\|search\|wsA1" | head -3; }
export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_conv_vgpr.so
echo "== bf16 x 3 convolution compiled with -amdgpu-mfma-vgpr-form (no AccVGPRs) as the co-runner"; python scripts/search_race_probe.py 100 conv1bf16 2>&1 | f
python scripts/search_race_probe.py 100 repr 2>&1 | f
echo "== its speed"; python scripts/conv_bench.py 2>&1 | grep -v "^[WE]2" | tail -6
unset DLPD_LIB_PATH
echo "== default library"; python scripts/conv_bench.py 2>&1 | grep -v "^[WE]2" | tail -6
