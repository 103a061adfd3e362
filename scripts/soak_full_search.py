"""Soak: the complete BASELINE config 2 search (48 ch x 64^3, 6-degree set, 68,760 rotations, K = 2000) run
three times -- twice at batch 16, once at batch 12 -- must give bit-identical ranked lists (no race between
the compute stream and the top-K side stream over ~4,300 batches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd.Utils.Rotations import Rotations
dev = torch.device("cuda:0")
C, L, K = 48, 64, 2000
rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
thr = bench.clash_threshold(recf, ligf)
R = Rotations(6, allow_generated=True, verbose=False).R
lists = []
for nb in (16, 16, 12):
    eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=thr, max_conf=K, batch=nb, device=dev)
    eng.set_receptor(rec, recf); eng.set_ligand(lig, ligf); eng.reset_top()
    torch.cuda.synchronize(); t0 = time.time()
    eng.search(R)
    torch.cuda.synchronize(); dt = time.time() - t0
    lists.append(eng.top_list())
    print("batch %2d: %d rotations in %.1f s = %.0f rot/s, %.3g poses/s; best %.4f, K-th %.4f" %
          (nb, R.shape[0], dt, R.shape[0] / dt, R.shape[0] * (2 * L) ** 3 / dt, lists[-1][0][4], lists[-1][-1][4]))
    del eng
assert lists[0] == lists[1], "rerun differs"
assert lists[0] == lists[2], "batch size changed the result"
print("identical lists:", len(lists[0]), "entries, rotations spanned:", len({t[0] for t in lists[0]}))
