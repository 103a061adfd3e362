"""Diagnostic: which STAGE of the scoring pipeline gives different bits when another stream of the process is busy?
One batch of 16 rotations is scored again and again (same inputs) while a host thread keeps a second stream busy with the
representation plugin's convolutions; after every stage the buffer it wrote is compared with the undisturbed run's.
usage: stage_race_probe.py <iterations> [load: repr|e3repr|matmul|none] [topk]"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from synth_pdb import write_protein_like_pdb
from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
from deeplocalproteindocking_amd.Utils.Rotations import Rotations

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LOAD = sys.argv[2] if len(sys.argv) > 2 else "repr"
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="dlpd_stage_")
pdb = {}
for name, n, seed in (("r1", 150, 21), ("l1", 90, 22)):
    pdb[name] = os.path.join(tmp, name + ".pdb")
    write_protein_like_pdb(pdb[name], n, seed)
torch.manual_seed(7)
repr_ = SE3MultiResReprScalar(multiplier=8)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=40.0).to(dev)
model.eval()
R = Rotations(20, allow_generated=True, verbose=False).R.numpy()
dk = Docker(model, box_size=80, resolution=1.25, max_conf=2000, rotations=R[:64], device=dev, randomize_rot=True, rotation_seed=7)
with torch.no_grad():
    dk.dockSE3(pdb["r1"], pdb["l1"], batch_size=2)
torch.cuda.synchronize()
eng = dk.engine
Rb = torch.from_numpy(R[16:32]).to(device=dev, dtype=torch.float32).contiguous()
BUF = {"coarse_k1": lambda: eng.wsA1, "coarse_k2": lambda: eng.wsB1, "coarse": lambda: eng.pre, "k1_rotate_zfft": lambda: eng.wsA,
       "k2_xy_corr": lambda: eng.wsB, "k3_zifft_filter": lambda: eng.V}
ORDER = ("coarse_k1", "coarse_k2", "coarse", "k1_rotate_zfft", "k2_xy_corr", "k3_zifft_filter")
ref = {}


def record(name):
    if name in BUF:
        ref[name] = BUF[name]().clone()


record.sub_stages = True
eng.score_batch(Rb, mark=record)
torch.cuda.synchronize()
again = {}
m2 = lambda n: again.__setitem__(n, BUF[n]().clone()) if n in BUF else None
m2.sub_stages = True
eng.score_batch(Rb, mark=m2)
torch.cuda.synchronize()
print("undisturbed rerun identical:", {k: bool(torch.equal(ref[k], again[k])) for k in ref}, flush=True)
del again
side = torch.cuda.Stream(device=dev)
x11 = torch.rand(1, 11, 80, 80, 80, device=dev)
e3net = x_blob = None
if LOAD == "e3repr":                # the E3 plugin on a protein-like input (zero away from a blob): the tile-occupancy (sparse) kernel
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4
    e3net = E3MultiResRepr4x4(multiplier=8).to(dev).eval()
    x_blob = torch.zeros(4, 11, 80, 80, 80, device=dev)
    x_blob[:, :, 24:52, 20:48, 28:60] = torch.rand(4, 11, 28, 28, 32, device=dev)
stop = threading.Event()


def worker():
    torch.cuda.set_device(dev)
    with torch.cuda.stream(side), torch.no_grad():
        while not stop.is_set():
            if LOAD == "repr":
                model.representation(x11)
            elif LOAD == "e3repr":
                e3net(x_blob)
            elif LOAD == "matmul":
                a = torch.randn(2048, 2048, device=dev)
                for _ in range(10):
                    a = (a @ a).tanh_()
            side.synchronize()


th = None
if LOAD not in ("none", "topk"):
    th = threading.Thread(target=worker, daemon=True)
    th.start()
    time.sleep(0.1)
# third party: the engine's own full radix select (what runs on its top-K stream during the second batch of a search),
# repeated on a stream of its own while the batch is scored ("topk" alone, or "<load>+topk")
TOPK = "topk" in sys.argv[2:] or (len(sys.argv) > 3 and "topk" in sys.argv[3])
th2 = None
if TOPK:
    side2 = torch.cuda.Stream(device=dev)
    Vsel = eng.V.clone()

    def selector():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side2):
            while not stop.is_set():
                for _ in range(4):
                    eng.top.select(Vsel.reshape(16, -1), 16, None)
                side2.synchronize()
    th2 = threading.Thread(target=selector, daemon=True)
    th2.start()
    time.sleep(0.05)
bad = {k: 0 for k in BUF}
first_bad = {}
detail = []
t0 = time.time()
for it in range(ITER):
    diffs = {}

    def check(name):
        if name in BUF:
            d = BUF[name]() != ref[name]
            n = int(d.sum())
            if n:
                diffs[name] = (n, d)
    check.sub_stages = True
    eng.score_batch(Rb, mark=check)
    torch.cuda.synchronize()
    if diffs:
        order = [k for k in ORDER if k in diffs]
        first = order[0]
        bad[first] += 1
        if len(detail) < 6:
            n, d = diffs[first]
            idx = d.reshape(-1).nonzero().reshape(-1)
            msg = "iteration %d: first differing stage %s, %d elements differ (flat indices %s ...), later stages %s" % (
                it, first, n, idx[:6].tolist(), {k: diffs[k][0] for k in order[1:]})
            if first == "coarse_k1":
                L1, C1 = eng.L1, eng.C1
                dd = d.view(16, C1, L1 + 1, L1, L1, 2).nonzero()
                a = ref[first].view(16, C1, L1 + 1, L1, L1, 2)
                g = BUF[first]().view(16, C1, L1 + 1, L1, L1, 2)
                msg += "\n      wsA1 max |diff| %.3g;" % float((a - g).abs().max())
                for dim, nm in enumerate(("b", "c", "k", "x", "y")):
                    u = dd[:, dim].unique()
                    msg += " %s: %s%s" % (nm, u[:12].tolist(), "..." if len(u) > 12 else "")
            else:
                a, g = ref[first], BUF[first]()
                fa, fg = (torch.view_as_real(a), torch.view_as_real(g)) if a.is_complex() else (a, g)
                dd = (fa != fg).nonzero()
                msg += "\n      shape %s, max |diff| %.3g (max |value| %.3g), NaNs %d;" % (tuple(fa.shape), float((fa - fg).abs().nan_to_num(0).max()), float(fa.abs().max()), int(fg.isnan().sum()))
                for dim in range(dd.shape[1]):
                    u = dd[:, dim].unique()
                    msg += " dim%d: %s%s" % (dim, u[:10].tolist(), "... (%d)" % len(u) if len(u) > 10 else "")
            detail.append(msg)
stop.set()
if th is not None:
    th.join()
if th2 is not None:
    th2.join()
print("top-K selector beside it: %s" % TOPK)
print("load %s: %d iterations in %.1f s; iterations whose FIRST differing stage was:" % (LOAD, ITER, time.time() - t0), bad)
for d in detail:
    print("  ", d)
