#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "load \|iteration\|top-K" | head -5; }
echo "== one batch scored 300 times: plugin convolutions AND the radix select beside it"; python scripts/stage_race_probe.py 300 repr topk 2>&1 | f
echo "== radix select alone beside it"; python scripts/stage_race_probe.py 300 topk 2>&1 | f
