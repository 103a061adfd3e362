"""Per-stage launch times of the two-resolution engine at the reference shapes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd.engine import DockingEngine, _ptr, _stream
from deeplocalproteindocking_amd.Utils.Rotations import Rotations
dev = torch.device("cuda:0")
L, C0, C1, nb = 80, 16, 32, 16
g = torch.Generator().manual_seed(0)
H = 24
eng = DockingEngine(L, C0, torch.randn(H, C0 + C1, generator=g), torch.randn(H, generator=g), torch.randn(1, H, generator=g),
                    torch.randn(1, generator=g), max_conf=2000, batch=nb, device=dev, coarse_channels=C1)
eng.set_receptor(torch.randn(C0, L, L, L, generator=g), torch.rand(L, L, L, generator=g), torch.randn(C1, 40, 40, 40, generator=g))
eng.set_ligand(torch.randn(C0, L, L, L, generator=g), torch.rand(L, L, L, generator=g), torch.randn(C1, 40, 40, 40, generator=g))
R = Rotations(15, verbose=False).R[:nb].to(device=dev, dtype=torch.float32).contiguous()
lib, st = eng.lib, _stream(dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
L1 = 40
print("coarse K1 %.3f" % t(lambda: lib.call("dlpd_zfft", _ptr(eng.lig1), _ptr(R), _ptr(eng.wsA1), nb, C1, L1, 0, 1, 20.0, st)))
print("coarse K2 %.3f" % t(lambda: lib.call("dlpd_xy_correlate", _ptr(eng.wsA1), _ptr(eng.recF1), _ptr(eng.wsB1), nb, C1, L1, 0, st)))
print("coarse K3 %.3f" % t(lambda: lib.call("dlpd_zifft_real", _ptr(eng.wsB1), _ptr(eng.aux), nb, C1, L1, 1, 5.0, st)))
print("fine   K1 %.3f" % t(lambda: lib.call("dlpd_zfft", _ptr(eng.lig), _ptr(R), _ptr(eng.wsA), nb, eng.CT, L, 0, 1, 40.0, st)))
print("fine   K2 %.3f" % t(lambda: lib.call("dlpd_xy_correlate", _ptr(eng.wsA), _ptr(eng.recF), _ptr(eng.wsB), nb, eng.CT, L, 0, st)))
print("fine   K3 %.3f" % t(lambda: lib.call("dlpd_zifft_filter_aux", _ptr(eng.wsB), _ptr(eng.V), nb, C0, 1, L, _ptr(eng.W1t), _ptr(eng.b1), _ptr(eng.W2), eng.b2, eng.HP, 1, 5.0, 100.0, _ptr(eng.aux), C1, st)))
print("topk      %.3f" % t(lambda: (eng.select_batch(eng.V, nb), eng.merge_batch(torch.arange(nb, dtype=torch.int32, device=dev), nb))))
print("whole     %.3f ms per %d rotations" % (t(lambda: eng.score_batch(R)), nb))
# unfused alternative for the fine grid: real correlation volumes + generic per-voxel filter
conv = torch.empty(nb, eng.CT, 160, 160, 160, device=dev)
print("fine K3 plain (17 ch -> real) %.3f" % t(lambda: lib.call("dlpd_zifft_real", _ptr(eng.wsB), _ptr(conv), nb, eng.CT, L, 1, 5.0, st)))
W1t = eng.W1t[:, :24].contiguous()
norm = torch.empty(nb, 160, 160, 160, device=dev)
print("generic filter (16+32 ch)     %.3f" % t(lambda: lib.call("dlpd_filter_mask", _ptr(conv), 16, 160, _ptr(eng.aux), 32, 80, _ptr(norm), 1.0, 1, _ptr(W1t), _ptr(eng.b1), _ptr(eng.W2), eng.b2, 24, _ptr(eng.V), nb, st)))
