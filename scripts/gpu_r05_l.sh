#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:\|wsA1\|  c:" | head -4; }
echo "== victim side: dlpd_corr.hip compiled with -amdgpu-waitcnt-forcezero (every wait is a full s_waitcnt vmcnt(0) lgkmcnt(0))"
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_forcezero.so python scripts/search_race_probe.py 60 conv1bf16 2>&1 | f
