#!/bin/bash
# round 3: whole -m gpu suite, soaks (complete searches, bit-identical lists) and the default bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r03_f
mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
python scripts/soak_full_search.py --angle_inc 6 --runs 16,16,12 --out $OUT/soak_full_search_6deg.json > /dev/null 2> $OUT/soak6.err; tail -3 $OUT/soak6.err
python scripts/soak_full_search.py --angle_inc 4 --runs 16,16 --out $OUT/soak_full_search_4deg.json > /dev/null 2> $OUT/soak4.err; tail -2 $OUT/soak4.err
python scripts/soak_real_shapes.py $OUT/soak_real_shapes.json > /dev/null 2> $OUT/soakr.err; tail -3 $OUT/soakr.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --angle_inc 4 --cpu_rotations 0 --no_real_shapes > $OUT/bench_angle4.json 2> $OUT/bench_angle4.err
python - <<PY
import json
for f in ("bench_default", "bench_angle4"):
    d = json.load(open("$OUT/%s.json" % f))
    print(f, "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v["ms_per_launch"], 3) for k, v in d["stages"].items()},
          "sustained", d.get("sustained", {}).get("ms_per_step"), "real", (d.get("real_shapes") or {}).get("ms_per_step"), "setup_s", d.get("per_rank_setup_s"))
PY
