#!/bin/bash
# faster atomic-free histogram: GPU top-K tests, the regression test, the stand-alone select time, the bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "topk or radix_select or top_list or candidate" 2>&1 | tail -3
unset DLPD_LIB_PATH
python - <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import torch
from deeplocalproteindocking_amd.engine import DeviceTopList
from deeplocalproteindocking_amd._lib import get_lib
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
for N in (128, 160):
    V = -torch.rand(16, N ** 3, device=dev, generator=g) * (torch.rand(16, N ** 3, device=dev, generator=g) < 0.7)
    top = DeviceTopList(2000, 16, dev, get_lib())
    top.select(V, 16, None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        top.select(V, 16, None)
    torch.cuda.synchronize()
    print("full radix select, 16 rotations x %d^3, K = 2000: %.3f ms" % (N, (time.perf_counter() - t0) / 20 * 1e3))
PY
python bench.py --cpu_rotations 0 --sustained_s 0 --strong_s 0 --no_real_shapes 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('bench: %.3f ms/step, value %.3e, gather hash %s' % (d['ms_per_step'], d['value'], d['gather_check']['list_sha256'][:16]))"
timeout 900 python scripts/search_race_probe.py 20 repr 2>&1 | tail -2
