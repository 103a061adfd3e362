"""Round-6 verdict item 7, measured: would the clash channel be cheaper as a 49th channel of the channels-last K1 launch than as
its own per-channel launch?  Times (HIP events, 16 rotations per launch, box 64): K1 channels-last with 48 channels + the
per-channel K1 of one channel (today's step) against K1 channels-last with 49 channels (Cp = 64: one more 8-channel chunk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd._lib import get_lib
lib = get_lib()
dev = torch.device("cuda:0")
L, nb, NZ = 64, 16, 65
g = torch.Generator().manual_seed(0)
vol = torch.randn(49, L, L, L, generator=g).to(dev)
ang = np.random.RandomState(0).uniform(-np.pi, np.pi, size=(nb, 3))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import docking_oracle as orc
R = torch.from_numpy(orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])).float().contiguous().to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
wsA = torch.empty(nb * 49 * NZ * L * L * 2, device=dev)
p = lambda t: t.data_ptr()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = {}
for C in (48, 49):
    cl = torch.empty(lib.call("dlpd_channels_last_floats", C, L), device=dev)
    lib.call("dlpd_make_channels_last", p(vol), p(cl), C, L, st)
    out["channels_last_%d" % C] = timed(lambda: lib.call("dlpd_zfft_channels_last", p(cl), p(R), p(wsA), nb, C, 49, 0, L, L / 2.0, st))
out["per_channel_clash"] = timed(lambda: lib.call("dlpd_zfft_oriented_ext", vol.data_ptr() + 48 * L ** 3 * 4, p(R), p(wsA), nb, 1, 49, 48, L, 0, 1,
                                                   L / 2.0, 0, 0, st))
print({k: round(v, 4) for k, v in out.items()})
print("today: %.4f ms   folded: %.4f ms" % (out["channels_last_48"] + out["per_channel_clash"], out["channels_last_49"]))
