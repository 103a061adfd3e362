"""Diagnostic: K1 (rotate + z FFT) time by rotation class -- which output axis carries the source z direction."""
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
R = Rotations(6, verbose=False).R.numpy()
# sample matrix of the kernel: p = c + M d with M[i][j] = R[j][i]?  classify by the kernel's own coefficients:
# pz = r2*dx + r5*dy + r8*dz with r = R row-major  ->  (R[0][2], R[1][2], R[2][2])
comp = np.abs(np.stack([R[:, 0, 2], R[:, 1, 2], R[:, 2, 2]], axis=1))
cls = comp.argmax(axis=1)
st = _stream(dev)
for k, name in enumerate(("source z along output x (|r2| max)", "along output y (|r5| max)", "along output z (|r8| max)")):
    ids = np.nonzero(cls == k)[0]
    sel = ids[np.linspace(0, len(ids) - 1, nb).astype(int)]
    Rd = torch.from_numpy(R[sel]).float().to(dev).contiguous()
    f = lambda: eng.lib.call("dlpd_zfft", _ptr(eng.lig), _ptr(Rd), _ptr(eng.wsA), nb, eng.CT, L, 0, 1, eng.center, st)
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): f()
    torch.cuda.synchronize()
    print("%-40s %5.1f %% of the set   K1 %.3f ms per %d rotations" % (name, 100.0 * len(ids) / len(R), (time.time() - t0) * 100, nb))
