#!/bin/bash
# same-box A/B of bench.py over variant libraries: scripts/gpu_ab_now.sh <out tag> <steps> <bench args ...> -- <lib> <lib> ...
# (lib = "default" or the name given to scripts/build_variant.py); every library twice, alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=$1; STEPS=$2; shift 2
ARGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {
  local tag=$1 lib=$2
  if [ "$lib" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$lib.so; fi
  timeout 300 python bench.py --steps $STEPS --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 "${ARGS[@]}" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json"))
    print("%-22s ms/step %.3f | " % ("$tag", d["ms_per_step"]) + " ".join("%s=%.3f" % (k, v["ms_per_launch"]) for k, v in d["stages"].items()))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-600:])
PY
}
run warm default
for rep in a b; do for lib in "$@"; do run ${lib}_$rep $lib; done; done
