#!/bin/bash
# the last tree of round 5: whole GPU suite, smoke, the driver's bench line (profiles/r05_zz4_*)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_zz4
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "
import __graft_entry__ as e
e.build(); e.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 200 $OUT/bench_default.err
python -c "
import json
d = json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][0])
print('bench: %.3f ms/step, value %.3e, hash %s, roofline frac %.3f' % (d['ms_per_step'], d['value'], d['gather_check']['list_sha256'][:16], d['roofline']['frac']))
print('e3:', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d['e3'].items() if k != 'workload'})"
