"""Diagnostic: per-phase cycle shares of a gather wave and of a transform / store wave of the role-split K1
(k_rotate_zfft_cl_rs; needs a -DDLPD_STAMPS=0 build of dlpd_k1r.hip: scripts/build_variant.py k1r_stamps --k1r=-DDLPD_STAMPS=0).
Never part of the product or of a timed number.   usage: stamps_k1r.py [config2|real|c48l80]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from deeplocalproteindocking_amd._lib import get_lib

G = ["prologue (twiddles, item 0)", "gather (addresses, loads, LDS stores)", "-", "-", "wait: item barrier", "wait: input-free barrier", "-", "-"]
X = ["prologue", "-", "first pass: inputs -> registers", "first-pass stores, second pass, untangle, global stores", "wait: item barrier",
     "wait: input-free barrier", "-", "-"]


class A:
    workload = sys.argv[1] if len(sys.argv) > 1 else "config2"
    channels = box = hidden = None
    max_conf, batch, k3_form, k1_form = 2000, 16, 0, 2


eng, wl = bench.build_workload(A.workload, A, torch.device("cuda:0"))
nb = A.batch
from oracle import docking_oracle as orc
ang = np.random.RandomState(3).uniform(-np.pi, np.pi, size=(nb, 3))
R = torch.from_numpy(orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])).float().cuda().contiguous()
eng.score_batch(R, cset=None, mark=lambda n: None)
torch.cuda.synchronize()
dll = get_lib()._dll
buf = (ctypes.c_ulonglong * 32)()
dll.dlpd_debug_read_stamps_k1r(buf)
for _ in range(3):
    eng.score_batch(R, mark=lambda n: None)
torch.cuda.synchronize()
dll.dlpd_debug_read_stamps_k1r(buf)
v = np.array(list(buf), dtype=np.float64)
for name, o, labels in (("gather wave 0", 0, G), ("transform/store wave 0", 16, X)):
    nblk, tot = v[o + 15], v[o:o + 8].sum()
    print(name, "blocks", int(nblk), "cycles/block %.0f" % (tot / max(nblk, 1)))
    for i in range(8):
        if v[o + i]:
            print("  %-58s %9.0f cyc/block  %5.1f %%" % (labels[i], v[o + i] / nblk, 100 * v[o + i] / tot))
