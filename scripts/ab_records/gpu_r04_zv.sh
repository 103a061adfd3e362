#!/bin/bash
# round 4, call zv: the complete 15-degree search at the real shapes through Docker.dock_volumes (three runs, list identity), and
# the GPU suite a second time on another box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_zv; mkdir -p $OUT
timeout 900 python scripts/soak_real_shapes.py > $OUT/soak_real_shapes.json 2> $OUT/soak_real_shapes.err; tail -4 $OUT/soak_real_shapes.err; head -c 700 $OUT/soak_real_shapes.json; echo
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -2
