#!/bin/bash
# flakiness check: the whole -m gpu suite N times on one box (default 2), failures listed per pass
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
for i in $(seq 1 ${1:-2}); do
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/suite_pass_$i.log 2>&1
  echo "pass $i: $(tail -1 gpurun_out/suite_pass_$i.log)"; grep "^FAILED\|^ERROR" gpurun_out/suite_pass_$i.log
done
python __graft_entry__.py > gpurun_out/graft_entry_main.log 2>&1; tail -2 gpurun_out/graft_entry_main.log
python -c "
import __graft_entry__ as e
e.build(); e.smoke(); print('smoke ok')" 2>&1 | tail -2
