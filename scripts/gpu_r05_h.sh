#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:\|search\|wsA1" | head -3; }
echo "== co-runner: torch bf16 matmul (library GEMM on the bf16 matrix instructions)"; python scripts/search_race_probe.py 100 matmul_bf16 2>&1 | f
echo "== co-runner: torch f32 matmul"; python scripts/search_race_probe.py 60 matmul 2>&1 | f
