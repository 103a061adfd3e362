#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:\|wsA1\|  c:\|  k:\|  b:" | head -8; }
echo "== victim K1cl<80> with the lane <-> pencil map of its transform passes XOR 32 (pencils 48-63 on lanes 16-31)"
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k1pxor.so python scripts/search_race_probe.py 50 conv1bf16 2>&1 | f
echo "== victim K1cl<80> with pencil rows padded by 29 instead of 13 elements (every pencil at another LDS address)"
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k1pad.so python scripts/search_race_probe.py 50 conv1bf16 2>&1 | f
