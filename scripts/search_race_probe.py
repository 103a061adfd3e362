"""Diagnostic: the whole search (two-stream top-K pipeline, clash provider) repeated beside a busy second stream, with an
integer checksum of every stage's output buffer per batch -- which (batch, stage) first differs from the undisturbed run?
usage: search_race_probe.py <searches> [load: repr|none]"""
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

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
PASSES = int(sys.argv[3]) if len(sys.argv) > 3 else 1
LOAD = sys.argv[2] if len(sys.argv) > 2 else "repr"
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="dlpd_srace_")
pdb = {}
for name, n, seed in (("r1", 150, 21), ("l1", 90, 22)):
    pdb[name] = os.path.join(tmp, name + ".pdb")
    write_protein_like_pdb(pdb[name], n, seed)
torch.manual_seed(7)
repr_ = SE3MultiResReprScalar(multiplier=8)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=40.0).to(dev)
model.eval()
R = Rotations(20, allow_generated=True, verbose=False).R.numpy()[:320]          # 20 batches
dk = Docker(model, box_size=80, resolution=1.25, max_conf=2000, rotations=R, device=dev, randomize_rot=True, rotation_seed=7)
with torch.no_grad():
    dk.dockSE3(pdb["r1"], pdb["l1"], batch_size=2)
torch.cuda.synchronize()
eng = dk.engine
Rd = torch.from_numpy(R).to(device=dev, dtype=torch.float32).contiguous()
ids = torch.arange(R.shape[0], dtype=torch.int32, device=dev)
STAGES = ("coarse_k1", "coarse_k2", "coarse", "k1_rotate_zfft", "k2_xy_corr", "k3_zifft_filter")


KEEP = {}


def one_search():
    sums = []
    eng.reset_top()
    NB = R.shape[0] // 16
    for bb in range(PASSES * NB):                 # PASSES > 1: the same batches again without a pause (is it the START of a search?)
        b = bb % NB
        cur = {}

        def mark(name):
            if name == "coarse":
                cur[name] = eng.pre.view(torch.int32).sum(dtype=torch.int64)
            elif name == "coarse_k1":
                cur[name] = eng.wsA1.view(torch.int32).sum(dtype=torch.int64)
                if bb == 1:
                    KEEP["wsA1"] = eng.wsA1.clone()
            elif name == "coarse_k2":
                cur[name] = eng.wsB1.view(torch.int32).sum(dtype=torch.int64)
            elif name == "k1_rotate_zfft":
                cur[name] = eng.wsA.view(torch.int32).sum(dtype=torch.int64)
            elif name == "k2_xy_corr":
                cur[name] = eng.wsB.view(torch.int32).sum(dtype=torch.int64)
            elif name == "k3_zifft_filter":
                k = eng._k ^ 1 if hasattr(eng, "_k") else 0
                cur[name] = eng._Vbuf[k].view(torch.int32).sum(dtype=torch.int64) if hasattr(eng, "_Vbuf") else eng.V.view(torch.int32).sum(dtype=torch.int64)
        mark.sub_stages = True
        eng.step(Rd[16 * b:16 * b + 16], ids[16 * b:16 * b + 16], mark=mark)
        sums.append(cur)
    ent = eng.top_entries()
    torch.cuda.synchronize()
    return [[int(s[k]) for k in STAGES] for s in sums], [np.asarray(x).tolist() for x in ent]


base_sums, base_list = one_search()
base_wsA1 = KEEP["wsA1"]
SHOWN = [0]
s2, l2 = one_search()
print("undisturbed rerun: checksums identical %s, list identical %s" % (s2 == base_sums, l2 == base_list), flush=True)
side = torch.cuda.Stream(device=dev)
x11 = torch.rand(1, 11, 80, 80, 80, device=dev)
x16 = torch.rand(1, 16, 80, 80, 80, device=dev)
from deeplocalproteindocking_amd import ops
w16 = torch.randn(16, 16, 5, 5, 5, device=dev) * 0.05
big1 = torch.rand(64, 1024, 1024, device=dev)
big2 = torch.empty_like(big1)
nbad = 0
for rep in range(REPS):
    stop = threading.Event()

    def worker():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side), torch.no_grad():
            while not stop.is_set():
                if LOAD == "matmul_bf16":                    # a library bf16 GEMM (hipBLASLt: the bf16 matrix instructions)
                    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
                    for _ in range(10):
                        a = (a @ a).tanh_()
                elif LOAD == "matmul":
                    a = torch.randn(2048, 2048, device=dev)
                    for _ in range(10):
                        a = (a @ a).tanh_()
                elif LOAD == "conv1":
                    model.representation.sequence_res0[2](x16)
                elif LOAD == "conv1f32":                      # the exact-f32 convolution: MFMA 16x16x4 f32, no AccVGPRs
                    ops.conv3d(x16, w16, precision="f32")
                elif LOAD == "conv1bf16":                     # the same layer through the bf16 x 3 kernel (AccVGPR accumulators)
                    ops.conv3d(x16, w16, precision="split_bf16")
                elif LOAD == "copy":
                    big2.copy_(big1)
                else:
                    model.representation(x11)
                side.synchronize()
    th = None
    if LOAD != "none":
        th = threading.Thread(target=worker, daemon=True)
        th.start()
        time.sleep(0.05)
    sums, lst = one_search()
    stop.set()
    if th is not None:
        th.join()
    nbad += int(sums != base_sums or lst != base_list)
    if (sums != base_sums or lst != base_list) and nbad <= 3:
        where = [(b, STAGES[j]) for b in range(len(sums)) for j in range(len(STAGES)) if sums[b][j] != base_sums[b][j]]
        print("search %d: list identical %s; differing (batch, stage): %s" % (rep, lst == base_list, where[:8]), flush=True)
        if (1, "coarse_k1") in where and SHOWN[0] < 1:
            SHOWN[0] += 1
            L1, C1 = eng.L1, eng.C1
            a = base_wsA1.view(16, C1, L1 + 1, L1, L1, 2)
            g = KEEP["wsA1"].view(16, C1, L1 + 1, L1, L1, 2)
            d = (a != g)
            idx = d.nonzero()
            print("   wsA1: %d floats differ; max |diff| %.3g (max |a| %.3g)" % (int(d.sum()), float((a - g).abs().max()), float(a.abs().max())))
            for dim, nm in enumerate(("b", "c", "k", "x", "y", "re/im")):
                u = idx[:, dim].unique()
                print("     %s: %s%s (%d distinct)" % (nm, u[:20].tolist(), " ..." if len(u) > 20 else "", len(u)))
            for row in idx[:40].tolist():
                print("       %s base %.9g got %.9g" % (row, float(a[tuple(row)]), float(g[tuple(row)])))
print("done: %d searches under load %s, %d with a differing checksum or list" % (REPS, LOAD, nbad))
