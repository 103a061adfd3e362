"""Diagnostic: per-phase cycle shares of one K2<128> wave (needs a --k2=-DDLPD_STAMPS=<wave> variant build, DLPD_LIB_PATH).
Never part of the product or of a timed number."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd._lib import get_lib
K2 = ["staging", "barrier waits", "fwd_y", "columns", "inv_y + out", "copy_out", "-", "loop_top"]
C, L, nb = 48, 64, 16
rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=bench.clash_threshold(recf, ligf),
                    max_conf=2000, batch=nb, device="cuda:0")
eng.set_receptor(rec, recf); eng.set_ligand(lig, ligf)
ang = np.random.RandomState(3).uniform(-3, 3, size=(nb, 3))
from deeplocalproteindocking_amd.Utils.Rotations import euler_to_matrices
R = torch.from_numpy(euler_to_matrices(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])).float().cuda().contiguous()
eng.score_batch(R); torch.cuda.synchronize()
dll = get_lib()._dll
buf = (ctypes.c_ulonglong * 16)()
dll.dlpd_debug_read_stamps_k2(buf)
for _ in range(3): eng.score_batch(R)
torch.cuda.synchronize()
dll.dlpd_debug_read_stamps_k2(buf)
v = np.array(list(buf), dtype=np.float64)
nblk, tot = v[15], v[:8].sum()
print("K2 blocks", int(nblk), "cycles/block %.0f = %.0f per slab and rotation" % (tot / nblk, tot / nblk / nb))
for i in range(8):
    if v[i]: print("  %-14s %9.0f cyc/slab  %5.1f %%" % (K2[i], v[i] / nblk / nb, 100 * v[i] / tot))
