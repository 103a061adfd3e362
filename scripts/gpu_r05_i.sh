#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "done:" ; }
for v in 1 2 4; do
  echo "== co-runner bf16x3 convolution, diagnostic build $v (1 no matrix instructions, 2 no staging stores, 4 no B-fragment reads)"
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_convdiag$v.so python scripts/search_race_probe.py 50 conv1bf16 2>&1 | f
done
echo "== default"; python scripts/search_race_probe.py 50 conv1bf16 2>&1 | f
