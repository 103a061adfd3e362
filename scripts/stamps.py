"""Diagnostic: per-phase cycle shares of one K2 / K3 wave (needs a -DDLPD_STAMPS=<wave> build).
Never part of the product or of a timed number."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd._lib import get_lib
K3 = ["dma_wait", "zbuild", "dma_issue", "fft", "barrierA", "accumulate", "barrierB", "-"]
K2 = ["A->LDS", "barriers", "fwd_y", "columns", "inv_y", "copy_out", "-", "loop_top"]
K1 = ["gather", "barrier", "z FFT (block passes)", "untangle+write", "-", "-", "-", "-"]
C, L, nb = (48, 64, 16) if len(sys.argv) < 2 else (16, 80, 16)       # any argument: the N = 160 grid
rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=bench.clash_threshold(recf, ligf),
                    max_conf=2000, batch=nb, device="cuda:0")
eng.set_receptor(rec, recf); eng.set_ligand(lig, ligf)
R = torch.eye(3).repeat(nb, 1, 1).cuda().contiguous()
if os.environ.get("DLPD_STAMP_OBLIQUE"):
    from oracle import docking_oracle as orc
    R = torch.from_numpy(orc.euler_to_matrix([0.7] * nb, [1.2] * nb, [-0.9] * nb)).float().cuda().contiguous()
eng.score_batch(R); torch.cuda.synchronize()
dll = get_lib()._dll
buf = (ctypes.c_ulonglong * 16)()
dll.dlpd_debug_read_stamps(buf); dll.dlpd_debug_read_stamps_k2(buf); dll.dlpd_debug_read_stamps_k1(buf)
for _ in range(3): eng.score_batch(R)
torch.cuda.synchronize()
for name, fn, labels in (("K3", dll.dlpd_debug_read_stamps, K3), ("K2", dll.dlpd_debug_read_stamps_k2, K2), ("K1", dll.dlpd_debug_read_stamps_k1, K1)):
    fn(buf)
    v = np.array(list(buf), dtype=np.float64)
    nblk, tot = v[15], v[:8].sum()
    print(name, "blocks", int(nblk), "cycles/block %.0f" % (tot / nblk))
    for i in range(8):
        if v[i]: print("  %-11s %9.0f cyc/block  %5.1f %%" % (labels[i], v[i] / nblk, 100 * v[i] / tot))
