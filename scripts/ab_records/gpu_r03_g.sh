#!/bin/bash
# round 3: new GPU tests (hidden 48, generic boxes, replay vs oracle, dockSE3/E3 vs oracle) + whole suite
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_g
mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=10 -x > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
