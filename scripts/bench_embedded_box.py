"""A box size WITHOUT a compiled plan (box_size is a free argument of the reference): throughput of the search on the
fused kernels inside the next compiled box (Docker._dock_volumes_embedded, the default) against the plan-free stand-alone
ops (embed_uncompiled_boxes = False), same ranked list.  [16 @ L^3, 32 @ (L/2)^3], K = 2000.
    bench_embedded_box.py [box=72] [rotations=512] [rotations of the slow path=32]"""
import os as _os; _os.environ.setdefault("DLPD_ALLOW_GENERATED_ROTATIONS", "1")
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
from deeplocalproteindocking_amd.Utils.Rotations import Rotations
L = int(sys.argv[1]) if len(sys.argv) > 1 else 72
nrot = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nslow = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev, K = torch.device("cuda:0"), 2000
torch.manual_seed(1)
repr_ = SyntheticRepr(num_outputs=(16, 32), seed=5, amplitude=0.12)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=0.02 * L ** 3).to(dev)
rec, lig = repr_.make(L, "rec"), repr_.make(L, "lig")
g = torch.Generator().manual_seed(2)
recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
R = Rotations(15, allow_generated=True, verbose=False).R
out = {"box": L, "grid": 2 * L}
lists = {}
for name, embed, n in (("embedded", True, nrot), ("ops", False, nslow)):
    dk = Docker(model, box_size=L, max_conf=K, rotations=R[:n], device=dev)
    dk.embed_uncompiled_boxes = embed
    dk.dock_volumes(rec, lig, recf, ligf, rot_indices=np.arange(min(n, 16)), write=False)      # warm-up, buffers
    torch.cuda.synchronize(); t0 = time.time()
    lists[name] = dk.dock_volumes(rec, lig, recf, ligf, write=False)
    torch.cuda.synchronize(); dt = time.time() - t0
    out[name] = {"path": dk.path, "rotations": n, "seconds": dt, "rot_per_s": n / dt, "engine_box": getattr(dk, "engine_box", None)}
    print("%s: %d rotations in %.2f s = %.1f rot/s" % (dk.path, n, dt, n / dt), file=sys.stderr)
# the slow path's list against the embedded path restricted to the same rotations
dk = Docker(model, box_size=L, max_conf=K, rotations=R[:nslow], device=dev)
a, b = dk.dock_volumes(rec, lig, recf, ligf, write=False), lists["ops"]
tol = 1e-4 * max(abs(t[4]) for t in b)
out["max_score_difference"] = max(abs(x[4] - y[4]) for x, y in zip(a, b))
out["same_rows"] = sum(x[:4] == y[:4] for x, y in zip(a, b))
assert out["max_score_difference"] <= tol
print(json.dumps(out))
