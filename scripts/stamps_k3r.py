"""Diagnostic: per-phase cycle shares of a transform wave and of a filter wave of the role-split K3
(k_zifft_filter_rs; needs a -DDLPD_STAMPS=0 build of dlpd_k3r.hip: scripts/build_variant.py k3r_stamps --k3r=-DDLPD_STAMPS=0).
Never part of the product or of a timed number.   usage: stamps_k3r.py [config2|real|c48l80]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from deeplocalproteindocking_amd._lib import get_lib

FFT = ["dma_wait", "first pass (raw -> pencils)", "dma_issue + second pass", "B1 wait", "B2 wait", "-", "-", "-"]
MLP = ["prologue (preact / bias)", "B1 wait", "copy values + B2 wait", "multiply-adds", "epilogue (layer 2, store)", "-", "-", "-"]


class A:
    workload = sys.argv[1] if len(sys.argv) > 1 else "config2"
    channels = box = hidden = None
    max_conf, batch, k3_form = 2000, 16, 2


eng, wl = bench.build_workload(A.workload, A, torch.device("cuda:0"))
nb = A.batch
from oracle import docking_oracle as orc
R = torch.from_numpy(orc.euler_to_matrix([0.7] * nb, [1.2] * nb, [-0.9] * nb)).float().cuda().contiguous()
eng.score_batch(R, cset=None, mark=lambda n: None)
torch.cuda.synchronize()
dll = get_lib()._dll
buf = (ctypes.c_ulonglong * 32)()
dll.dlpd_debug_read_stamps_k3r(buf)
for _ in range(3):
    eng.score_batch(R, mark=lambda n: None)
torch.cuda.synchronize()
dll.dlpd_debug_read_stamps_k3r(buf)
v = np.array(list(buf), dtype=np.float64)
for name, o, labels in (("transform wave 0", 0, FFT), ("filter wave 0", 16, MLP)):
    nblk, tot = v[o + 15], v[o:o + 8].sum()
    print(name, "blocks", int(nblk), "cycles/block %.0f" % (tot / max(nblk, 1)))
    for i in range(8):
        if v[o + i]:
            print("  %-32s %9.0f cyc/block  %5.1f %%" % (labels[i], v[o + i] / nblk, 100 * v[o + i] / tot))
