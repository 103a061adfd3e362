#!/bin/bash
# round 4, call zj: the committed tree once more -- GPU suite, smoke(), the driver's bench command, the one-rank torchrun launch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_zj; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value %.3e ms/step %.3f frac %.3f real %.3f c48l80 %.3f e3 %.2f gather %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["real_shapes"]["ms_per_step"], d["c48l80"]["ms_per_step"], d["e3"]["ms_per_launch"], d["gather_check"]["list_sha256"][:12]))
PY
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; python -c "
import json; d=json.load(open('$OUT/bench_torchrun1.json')); print('torchrun: value %.3e n_gpus %d gather %s' % (d['value'], d['n_gpus'], d['gather_check']['list_sha256'][:12]))"
