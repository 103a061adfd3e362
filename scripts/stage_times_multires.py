"""Per-stage launch times of the engine at the reference's real shapes (diagnostic).
usage: stage_times_multires.py [R|S80]   R = [16@80, 32@40] (default), S80 = 48 @ 80 single resolution"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
entry.build()
from bench import StageTimer
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd.Utils.Rotations import Rotations
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "R"
L, C0, C1, nb = (80, 16, 32, 16) if mode == "R" else (80, 48, 0, 8)
g = torch.Generator().manual_seed(0)
H = 24
eng = DockingEngine(L, C0, torch.randn(H, C0 + C1, generator=g), torch.randn(H, generator=g), torch.randn(1, H, generator=g),
                    torch.randn(1, generator=g), max_conf=2000, batch=nb, device=dev, coarse_channels=C1)
coarse = lambda: torch.randn(C1, 40, 40, 40, generator=g) if C1 else None
eng.set_receptor(torch.randn(C0, L, L, L, generator=g), torch.rand(L, L, L, generator=g), coarse())
eng.set_ligand(torch.randn(C0, L, L, L, generator=g), torch.rand(L, L, L, generator=g), coarse())
R = Rotations(15, allow_generated=True, verbose=False).R[:nb].to(device=dev, dtype=torch.float32).contiguous()
ids = torch.arange(nb, dtype=torch.int32, device=dev)
tm = StageTimer()
for it in range(6):
    V = eng.score_batch(R, mark=tm.mark if it else None)
    if it:
        eng.select_batch(V, nb); eng.merge_batch(ids, nb); tm.mark("topk")
torch.cuda.synchronize()
s = tm.summary()
for k, v in s.items():
    print("%-18s %.3f ms" % (k, v))
print("sum %.3f ms per %d rotations -> %.0f rot/s serial" % (sum(s.values()), nb, nb / sum(s.values()) * 1e3))
