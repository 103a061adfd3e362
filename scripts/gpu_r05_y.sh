#!/bin/bash
# bench.py's multi-rank branches after the collective warm-up: the two-rank (gloo) and one-rank (RCCL) tests, and the gather times
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "bench" 2>&1 | tail -3
python bench.py --gpus 1 --force_group --backend nccl --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 0 --no_pmc 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('one-rank RCCL: %.3f ms/step, collective %s' % (d['ms_per_step'], d['collective']))"
python bench.py --gpus 2 --backend gloo --same_device --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 0 --no_pmc 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('two ranks gloo one GPU: %.3f ms/step, collective %s' % (d['ms_per_step'], d['collective']))"
