#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:\|search\|wsA1" | head -4; }
echo "== co-runner: exact-f32 convolution (MFMA 16x16x4 f32, no AccVGPRs)"; python scripts/search_race_probe.py 80 conv1f32 2>&1 | f
echo "== co-runner: bf16 x 3 convolution (MFMA 16x16x32 bf16, AccVGPR accumulators)"; python scripts/search_race_probe.py 80 conv1bf16 2>&1 | f
timeout 900 python -m pytest tests/test_atoms.py -q -m gpu -x -k "overlapped_plugin" 2>&1 | tail -3
