#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lib in default topk_atomic; do
if [ "$lib" = "default" ]; then unset DLPD_LIB_PATH; else export DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$lib.so; fi
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
    print("$lib: full radix select, 16 rotations x %d^3, K = 2000: %.3f ms" % (N, (time.perf_counter() - t0) / 20 * 1e3))
PY
done
