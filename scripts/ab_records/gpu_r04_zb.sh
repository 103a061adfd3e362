#!/bin/bash
# round 4, call zb: K2 of boxes 80 / 40 on the packed receptor spectrum (default) against the natural layout (--natural_receptor)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_zb; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "packed or real or two_res or k3_role" 2>&1 | tail -3
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-22s ms/step %.3f | " % (sys.argv[1], d["ms_per_step"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for wl in real c48l80; do
  for rep in a b; do
    for v in packed natural; do
      X=""; [ $v = natural ] && X="--natural_receptor"
      timeout 300 python bench.py --workload $wl --steps 40 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --no_pmc --gather_rotations 0 --strong_s 0 $X > $OUT/${wl}_${v}_$rep.json 2> $OUT/${wl}_${v}_$rep.err
      show ${wl}_${v}_$rep $OUT/${wl}_${v}_$rep.json
    done
  done
done
