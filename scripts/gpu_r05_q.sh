#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "scores_do_not_change_beside or topk" --durations=4 2>&1 | tail -8
python - <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import torch
import __graft_entry__ as e
e.build()
from deeplocalproteindocking_amd.engine import DeviceTopList
from deeplocalproteindocking_amd._lib import get_lib
dev = torch.device("cuda:0")
for N in (128, 160):
    V = -torch.rand(16, N ** 3, device=dev) * (torch.rand(16, N ** 3, device=dev) < 0.7)
    top = DeviceTopList(2000, 16, dev, get_lib())
    top.select(V, 16, None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        top.select(V, 16, None)
    torch.cuda.synchronize()
    print("full radix select, 16 rotations x %d^3, K = 2000: %.3f ms" % (N, (time.perf_counter() - t0) / 20 * 1e3))
PY
