#!/bin/bash
# round 4, call n: driver-facing entry points on the final tree: smoke(), wall time of `python bench.py`, bench under torch.distributed.run with one rank
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_n
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s.%N); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; t1=$(date +%s.%N)
python -c "print('python bench.py wall seconds: %.1f' % ($t1 - $t0))"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value %.4g ms/step %.3f e3 %.2f ms real %.3f c48l80 %.3f strong %s gather %s" % (d["value"], d["ms_per_step"], d["e3"]["ms_per_launch"], d["real_shapes"]["ms_per_step"], d["c48l80"]["ms_per_step"], d["strong"]["seconds"], d["gather_check"]["list_sha256"][:12]))
PY
t0=$(date +%s.%N); timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; t1=$(date +%s.%N)
python -c "print('torchrun 1-rank wall seconds: %.1f' % ($t1 - $t0))"
tail -c 300 $OUT/bench_torchrun1.err
python - <<PY
import json
d = json.load(open("$OUT/bench_torchrun1.json"))
print("torchrun: value %.4g n_gpus %d gather %s" % (d["value"], d["n_gpus"], d["gather_check"]["list_sha256"][:12]))
PY
