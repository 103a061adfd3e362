"""Diagnostic: K1 (rotate + z FFT) time by rotation class, for the pair and the quad gather, with the slab
orientation the engine would choose.  Classes: which output axis carries the source z direction, and how
strongly (bins of the largest in-plane |source-z| component after orientation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd.engine import DockingEngine, _ptr, _stream
from deeplocalproteindocking_amd.Utils.Rotations import Rotations
dev = torch.device("cuda:0")
C, L, nb = 48, 64, 16
g = torch.Generator().manual_seed(0)
eng = DockingEngine(L, C, torch.randn(24, C, generator=g), torch.randn(24, generator=g), torch.randn(1, 24, generator=g),
                    torch.randn(1, generator=g), max_conf=2000, batch=nb, device=dev)
eng.set_ligand(torch.randn(C, L, L, L, generator=g), torch.rand(L, L, L, generator=g))
R = Rotations(6, allow_generated=True, verbose=False).R.numpy()
tr = DockingEngine.prefers_transposed(R)
inplane = np.maximum(np.where(tr, np.abs(R[:, 0, 2]), np.abs(R[:, 1, 2])), np.abs(R[:, 2, 2]))   # best in-plane |source z|
zdom = np.abs(R[:, 2, 2]) >= np.where(tr, np.abs(R[:, 0, 2]), np.abs(R[:, 1, 2]))
st = _stream(dev)
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): fn()
    torch.cuda.synchronize(); return (time.time() - t0) * 100
print("bin of best in-plane |source z| ; share of set ; K1 ms per 16 rotations: pairs / quads")
for lo, hi in ((0.57, 0.7), (0.7, 0.8), (0.8, 0.9), (0.9, 0.97), (0.97, 1.01)):
    for zd in (True, False):
        ids = np.nonzero((inplane >= lo) & (inplane < hi) & (zdom == zd))[0]
        if len(ids) < nb: continue
        sel = ids[np.linspace(0, len(ids) - 1, nb).astype(int)]
        flag = int(tr[sel].mean() > 0.5)
        Rd = torch.from_numpy(R[sel]).float().to(dev).contiguous()
        a = t(lambda: eng.lib.call("dlpd_zfft_oriented", _ptr(eng.lig), _ptr(Rd), _ptr(eng.wsA), nb, eng.CT, eng.CT, 0, L, 0, 1, eng.center, flag, st))
        b = t(lambda: eng.lib.call("dlpd_zfft_quads", _ptr(eng.ligq), _ptr(Rd), _ptr(eng.wsA), nb, eng.CT, eng.CT, 0, L, eng.center, flag, st))
        print("[%.2f, %.2f) %s-dominant  %5.1f %%   %.3f / %.3f" % (lo, hi, "z" if zd else "y", 100.0 * len(ids) / len(R), a, b))
