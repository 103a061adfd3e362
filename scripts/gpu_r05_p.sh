#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_p
mkdir -p $OUT
f() { grep -v "Adding random\|\[\[\|^  *\[\|Warn\|warn\|amdgpu.ids" | grep "load \|done:" | cut -c1-220; }
echo "== k_topk_hist without LDS atomics: one batch beside the plugin convolutions AND the radix select"; python scripts/stage_race_probe.py 300 repr topk 2>&1 | f
echo "== whole searches beside the bf16 x 3 convolution"; python scripts/search_race_probe.py 100 conv1bf16 2>&1 | f
echo "== whole searches beside the plugin"; python scripts/search_race_probe.py 100 repr 2>&1 | f
timeout 2400 python -m pytest tests -q -m gpu -x --durations=6 > $OUT/pytest_gpu.log 2>&1; tail -9 $OUT/pytest_gpu.log | grep -v "^$"
timeout 600 python bench.py --steps 40 --cpu_rotations 0 --no_real_shapes --sustained_s 4 --strong_s 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k: round(v['ms_per_launch'],3) for k,v in d['stages'].items()}, 'sustained', d['sustained']['ms_per_step'], d['gather_check']['list_sha256'][:16])"
