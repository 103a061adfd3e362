#!/bin/bash
# round 4, call d: (i) hidden width 32 on the 4-voxel filter waves (spilling) vs the 2-voxel WIDE configuration (k3wide24);
# (ii) rotations per launch 8 .. 32 on config 2 and the real shapes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for wl in config2 real; do
  bash scripts/gpu_ab_now.sh r04_d_h32_$wl 40 --workload $wl --hidden 32 --no_pmc --gather_rotations 0 --strong_s 0 -- default k3wide24
done
OUT=gpurun_out/r04_d_batch
mkdir -p $OUT
for wl in config2 real; do
  for nb in 8 12 16 20 24 32; do
    timeout 300 python bench.py --steps 40 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --no_pmc --gather_rotations 0 --strong_s 0 --workload $wl --batch $nb > $OUT/${wl}_$nb.json 2> $OUT/${wl}_$nb.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/${wl}_$nb.json"))
    print("%-8s nb=%-3d ms/step %.3f  rot/s %.1f | " % ("$wl", $nb, d["ms_per_step"], d["rot_per_s"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print("$wl $nb FAILED", e, open("$OUT/${wl}_$nb.err").read()[-400:])
PY
  done
done
