"""Ad-hoc first GPU run: parity at 4x32^3 and 48x64^3 and per-kernel timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import docking_oracle as orc
from deeplocalproteindocking_amd.engine import DockingEngine, _ptr, _stream

def run(L, C, nb, nrep):
    torch.manual_seed(0)
    H = C // 2
    rec, lig = torch.randn(C, L, L, L) * 0.05, torch.randn(C, L, L, L) * 0.05
    recf, ligf = torch.rand(L, L, L), torch.rand(L, L, L)
    W1, b1, W2, b2 = torch.randn(H, C) * 0.3, torch.randn(H) * 0.1, torch.randn(1, H), torch.randn(1)
    ang = np.random.RandomState(1).uniform(-3, 3, size=(nb, 3))
    R = orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])
    thr = float(L) ** 3 * 0.25 * 0.5
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=thr, max_conf=2000, batch=nb, device="cuda:0")
    eng.set_receptor(rec, recf); eng.set_ligand(lig, ligf)
    Rd = torch.from_numpy(R).float().cuda().contiguous()
    V = eng.score_batch(Rd).cpu()
    torch.cuda.synchronize()
    # oracle for first 2 rotations
    for i in range(min(2, nb)):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        lr = orc.rotate_volume(lig[None], Rb); lfr = orc.rotate_volume(ligf[None, None], Rb)
        mask, nrm = orc.clash_mask(recf[None, None], lfr, thr)
        Vo = (mask * orc.score_volumes([rec[None]], [lr], W1, b1, W2, b2, clip=5.0))[0]
        mm = ((V[i] == 0) != (Vo == 0)).sum().item()
        print("L=%d C=%d rot %d: V err %.3g scale %.3g mask-frac %.3f mask mismatches %d" % (
            L, C, i, (V[i] - Vo).abs().max().item(), Vo.abs().max().item(), mask.mean().item(), mm), flush=True)
    # timings per stage
    lib, st = eng.lib, _stream(eng.device)
    def timeit(fn, n=nrep):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t1 = timeit(lambda: lib.call("dlpd_zfft", _ptr(eng.lig), _ptr(Rd), _ptr(eng.wsA), nb, eng.CT, L, 0, 1, eng.center, st))
    t2 = timeit(lambda: lib.call("dlpd_xy_correlate", _ptr(eng.wsA), _ptr(eng.recF), _ptr(eng.wsB), nb, eng.CT, L, 0, st))
    t3 = timeit(lambda: lib.call("dlpd_zifft_filter", _ptr(eng.wsB), _ptr(eng.V), nb, eng.C, 1, L, _ptr(eng.W1t), _ptr(eng.b1), _ptr(eng.W2), eng.b2, eng.HP, 1, 5.0, thr, st))
    t4 = timeit(lambda: eng.select_batch(eng.V, nb))
    ids = torch.arange(nb, dtype=torch.int32, device="cuda")
    eng.reset_top()
    t5 = timeit(lambda: eng.merge_batch(ids, nb))
    tot = timeit(lambda: (eng.score_batch(Rd), eng.select_batch(eng.V, nb), eng.merge_batch(ids, nb)))
    print("L=%d C=%d nb=%d  ms/batch: K1 %.3f K2 %.3f K3 %.3f topk %.3f merge %.3f  total %.3f  -> %.1f rot/s, %.3g poses/s" % (
        L, C, nb, t1, t2, t3, t4, t5, tot, nb / tot * 1e3, nb / tot * 1e3 * (2 * L) ** 3), flush=True)

run(32, 4, 8, 20)
run(64, 48, 8, 10)
run(64, 48, 16, 5)
