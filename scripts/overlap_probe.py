"""Diagnostic (round 3, measured and rejected): do K1 (gather / store bound) and K3 (LDS / VALU bound) of DIFFERENT batches overlap when launched on two
streams?  Times K1, K2, K3 alone, K1 || K3 and K1 || K2 (one launch of 16 rotations each, config 2 or real shapes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench
from deeplocalproteindocking_amd.engine import _ptr


class A:
    workload = sys.argv[1] if len(sys.argv) > 1 else "config2"
    channels = box = None
    max_conf, batch, k3_form = 2000, 16, 0


dev = torch.device("cuda:0")
eng, wl = bench.build_workload(A.workload, A, dev)
nb, L = A.batch, eng.L
from oracle import docking_oracle as orc
rs = np.random.RandomState(1)
ang = rs.uniform(-np.pi, np.pi, size=(nb, 3))
R = torch.from_numpy(orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])).float().to(dev).contiguous()
eng.score_batch(R, mark=lambda n: None)            # fills wsA / wsB once
torch.cuda.synchronize()
call = eng.lib.call
wsA2 = torch.empty_like(eng.wsA)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
has_clip, clip = 1, 5.0


def k1(stream, wsA):
    st = stream.cuda_stream
    call("dlpd_zfft_channels_last", _ptr(eng.ligcl), _ptr(R), _ptr(wsA), nb, eng.C, eng.CT, 0, L, eng.center, st)
    call("dlpd_zfft_oriented", eng.lig.data_ptr() + eng.C * L ** 3 * 4, _ptr(R), _ptr(wsA), nb, 1, eng.CT, eng.C, L, 0, 1,
         eng.center, 0, st)


def k2(stream):
    call("dlpd_xy_correlate_oriented", _ptr(eng.wsA), _ptr(eng.recF), _ptr(eng.wsB), nb, eng.CT, L, 0, 0, stream.cuda_stream)


def k3(stream):
    aux, C1 = (_ptr(eng.pre), eng.C1) if eng.C1 else (0, 0)
    call("dlpd_zifft_filter_form", _ptr(eng.wsB), _ptr(eng.V), nb, eng.C, 1, L, _ptr(eng.W1t), _ptr(eng.b1), _ptr(eng.W2), eng.b2,
         eng.HP, has_clip, clip, eng.threshold, aux, C1, int(C1 > 0), 0, 0, 0, 0, 0, stream.cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def both(fa, fb):
    def run():
        fa()
        fb()
        s1.synchronize()
        s2.synchronize()
    return run


t1 = timed(lambda: (k1(s1, wsA2), s1.synchronize()))
t2 = timed(lambda: (k2(s1), s1.synchronize()))
t3 = timed(lambda: (k3(s1), s1.synchronize()))
t13 = timed(both(lambda: k3(s1), lambda: k1(s2, wsA2)))
t12 = timed(both(lambda: k2(s1), lambda: k1(s2, wsA2)))
t23 = timed(both(lambda: k3(s1), lambda: k2(s2)))
print("K1 %.3f  K2 %.3f  K3 %.3f ms | K1||K3 %.3f (sum %.3f)  K1||K2 %.3f (sum %.3f)  K2||K3 %.3f (sum %.3f)" %
      (t1, t2, t3, t13, t1 + t3, t12, t1 + t2, t23, t2 + t3))
