#!/bin/bash
# round 4, call zm: rotations per launch re-swept on the final kernels (real shapes and config 2)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_zm; mkdir -p $OUT
for wl in real config2; do
  for nb in 16 12 20 24 32 16; do
    timeout 300 python bench.py --workload $wl --batch $nb --steps 40 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --no_pmc --gather_rotations 0 --strong_s 0 > $OUT/${wl}_$nb.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/${wl}_$nb.json')); print('$wl batch $nb: %.3f ms/step  %.1f rotations/s' % (d['ms_per_step'], $nb / d['ms_per_step'] * 1e3))"
  done
done
