#!/bin/bash
# round 6, call d: the new GPU tests (unwritten activations, K1 occupancy maps), dockE3 / replay tests, and the bench line with the
# real_protein and e3 extras
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/${1:-r06_d}
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -k "unwritten or dockE3 or tile_occupancy or maxpool or e3_plugin or occupancy_maps or replay or sweep" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python bench.py --steps 40 --cpu_rotations 0 --sustained_s 0 --no_pmc > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms_per_step", d["ms_per_step"], {k: round(v["ms_per_launch"], 3) for k, v in d["stages"].items()})
for k in ("real_shapes", "real_protein"):
    r = d.get(k) or {}
    print(k, r.get("error") or (round(r["ms_per_step"], 3), {a: round(b["ms_per_launch"], 3) for a, b in r["stages"].items()}))
rp = d.get("real_protein") or {}
if "without_k1_occupancy_maps" in rp:
    w = rp["without_k1_occupancy_maps"]
    print("  without maps", round(w["ms_per_step"], 3), {a: round(b["ms_per_launch"], 3) for a, b in w["stages"].items()})
    print("  switches", rp["kernel_switches"]["k1_occupancy_maps"])
print("e3", json.dumps(d.get("e3"))[:1500])
PY
