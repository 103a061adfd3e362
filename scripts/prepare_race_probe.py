"""Diagnostic: does work on a second stream of the same process perturb a search?  One dockSE3 on the main stream is
repeated while a host thread keeps a stream of its own busy with ONE kind of work (mode); every ranked list is compared
with the undisturbed one.   usage: prepare_race_probe.py <repeats> <mode,mode,...>"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from synth_pdb import write_protein_like_pdb
from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.engine import DockingEngine
from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
from deeplocalproteindocking_amd.Utils.Rotations import Rotations

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "prepare"]
OPTS = sys.argv[3].split(",") if len(sys.argv) > 3 else []
import deeplocalproteindocking_amd.Docker.Docker as DD
if "noprefilter" in OPTS:
    DD.DockingEngine = lambda *a, **k: DockingEngine(*a, prefilter=False, **k)
if "poison" in OPTS:
    from deeplocalproteindocking_amd._lib import get_lib
    get_lib().call("dlpd_debug_poison_lds", 1)
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="dlpd_race_")
pdb = {}
for name, n, seed in (("r1", 150, 21), ("l1", 90, 22), ("r2", 120, 41), ("l2", 100, 42)):
    pdb[name] = os.path.join(tmp, name + ".pdb")
    write_protein_like_pdb(pdb[name], n, seed)
torch.manual_seed(7)
repr_ = SE3MultiResReprScalar(multiplier=8)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=40.0).to(dev)
model.eval()
R = Rotations(20, allow_generated=True, verbose=False).R.numpy()
dk = Docker(model, box_size=80, resolution=1.25, max_conf=2000, rotations=R, device=dev, randomize_rot=True, rotation_seed=7)


def search():
    with torch.no_grad():
        dk.dockSE3(pdb["r1"], pdb["l1"], batch_size=2)
    torch.cuda.synchronize()
    return list(dk.top_list)


base = search()
assert base == search() and base[0][4] < -0.05, base[0]
side = torch.cuda.Stream(device=dev)
x11 = torch.rand(1, 11, 80, 80, 80, device=dev)
torch.cuda.synchronize()


def worker(mode, stop):
    torch.cuda.set_device(dev)
    engB = None
    with torch.cuda.stream(side), torch.no_grad():
        while not stop.is_set():
            if mode == "prepare":
                dk.prepare(pdb["r2"], pdb["l2"], "SE3", slot=1, stream=side)
            elif mode == "repr":
                model.representation(x11)
            elif mode == "engine_new":
                rec = [v.reshape((-1,) + tuple(v.shape[-3:])) for v in model.representation(x11)]
                from deeplocalproteindocking_amd.Models.DockingModels import fused_filter_parameters
                dk._engine_pool.pop(1, None)
                dk._make_engine(rec, x11.sum(dim=1)[0], dk.launch_batch, fused_filter_parameters(model), slot=1)
            elif mode == "alloc":
                a = [torch.zeros(64, 80, 80, 80, device=dev) for _ in range(8)]
                del a
            elif mode == "project":
                be = dk._need_backend()
                c, nt, of, _, _ = dk.load_batch([pdb["l2"]], bbox_center=False)
                be.project(c, nt, of, 80, 1.25, dev)
            elif mode == "matmul":
                a = torch.randn(2048, 2048, device=dev)
                for _ in range(10):
                    a = (a @ a).tanh_()
            side.synchronize()


for mode in MODES:
    bad, t0 = [], time.time()
    for i in range(REPS):
        stop = threading.Event()
        th = None
        if mode != "none":
            th = threading.Thread(target=worker, args=(mode, stop), daemon=True)
            th.start()
            time.sleep(0.05)
        got = search()
        stop.set()
        if th is not None:
            th.join()
        torch.cuda.synchronize()
        if got != base:
            nd = sum(a != b for a, b in zip(got, base))
            sd = max(abs(a[4] - b[4]) for a, b in zip(got, base))
            bad.append((i, nd, sd))
            if len(bad) <= 2:
                first = next(j for j, (a, b) in enumerate(zip(got, base)) if a != b)
                print("   search %d: first difference at rank %d" % (i, first))
                for j in range(max(0, first - 1), min(2000, first + 4)):
                    print("     %4d base %s | got %s" % (j, base[j], got[j]))
                print("   poses only in base:", sorted(set(base) - set(got))[:4], " only in got:", sorted(set(got) - set(base))[:4])
    print("mode %-12s %d searches, %d differ %s  (%.1f s)" % (mode, REPS, len(bad), bad[:6], time.time() - t0), flush=True)
