#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
f() { grep "load " | cut -c1-220; }
for v in 14 13 11 7; do
  echo "== radix select reduced to ONE kernel kind (14: hist only, 13: scan only, 11: collect only, 7: sort only) beside the convolutions: $v"
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_topkdiag$v.so python scripts/stage_race_probe.py 150 repr topk 2>&1 | f
done
