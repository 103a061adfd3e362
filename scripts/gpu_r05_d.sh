#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:\|search\|wsA1\|   " | head -${2:-6}; }
echo "== default library, load repr"; python scripts/search_race_probe.py 60 repr 2>&1 | f
echo "== default library, load matmul"; python scripts/search_race_probe.py 60 matmul 2>&1 | f
echo "== default library, load conv1 (one 16->16 convolution)"; python scripts/search_race_probe.py 60 conv1 2>&1 | f
echo "== default library, load copy"; python scripts/search_race_probe.py 60 copy 2>&1 | f
echo "== dlpd_corr.hip without the packed-math inline asm (-DDLPD_PK=0), load repr"; DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_nopk.so python scripts/search_race_probe.py 60 repr 2>&1 | f
