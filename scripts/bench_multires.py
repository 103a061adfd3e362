"""Throughput of the reference's real model shapes [16 @ 80^3, 32 @ 40^3] -> 160^3 (BASELINE configs 4/5
geometry, synthetic volumes) through Docker.dock_volumes' multi-resolution path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
from deeplocalproteindocking_amd.Utils.Rotations import Rotations
dev = torch.device("cuda:0")
L, K, nrot, nb = 80, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(1)
repr_ = SyntheticRepr(num_outputs=(16, 32), seed=5, amplitude=0.12)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=0.02 * L ** 3).to(dev)
rec, lig = repr_.make(L, "rec"), repr_.make(L, "lig")
g = torch.Generator().manual_seed(2)
recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
R = Rotations(15, allow_generated=True, verbose=False).R.numpy()[:nrot]
dk = Docker(model, box_size=L, max_conf=K, rotations=R, device=dev)
dk.dock_volumes(rec, lig, recf, ligf, batch_size=nb, write=False, rot_indices=np.arange(nb))
torch.cuda.synchronize(); t = time.time()
dk.dock_volumes(rec, lig, recf, ligf, batch_size=nb, write=False)
torch.cuda.synchronize(); dt = time.time() - t
print("multires [16@80,32@40] N=160: %.1f rot/s, %.3g poses/s (batch %d)" % (nrot / dt, nrot * 160 ** 3 / dt, nb))
