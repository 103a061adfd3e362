#!/bin/bash
# flakiness check of SURVEY 8(e)'s CPU evidence: the multi-rank (gloo) and replay test files N times (default 10)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=${2:-profiles/r06_cpu_multirank_repeat.log}
: > $OUT
for i in $(seq 1 ${1:-10}); do
  timeout 3000 python -m pytest -q -m "not gpu" -p no:cacheprovider tests/test_distributed_gloo.py tests/test_replay_local_test.py > /tmp/mr_pass_$i.log 2>&1
  echo "pass $i: $(tail -1 /tmp/mr_pass_$i.log)" >> $OUT; grep "^FAILED\|^ERROR" /tmp/mr_pass_$i.log >> $OUT
done
