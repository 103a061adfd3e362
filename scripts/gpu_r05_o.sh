#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep "load " | cut -c1-220; }
echo "== hist only, its LDS atomics replaced by plain LDS read-modify-writes"; DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_topkdiag30.so python scripts/stage_race_probe.py 150 repr topk 2>&1 | f
echo "== hist only, without its global atomics"; DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_topkdiag46.so python scripts/stage_race_probe.py 150 repr topk 2>&1 | f
