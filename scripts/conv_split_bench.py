"""Diagnostic: the split-bf16 convolution alone on the layer shapes of E3MultiResRepr4x4(multiplier=8) (no torch / f32 legs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd import ops
dev = torch.device("cuda:0")
B = 16
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3, r
tot = 0.0
out = []
for (cin, cout, ks, D, cnt) in ((11, 16, 5, 80, 1), (16, 16, 3, 80, 3), (16, 16, 5, 80, 1), (16, 32, 5, 40, 1), (32, 32, 3, 40, 3)):
    x = torch.randn(B, cin, D, D, D, device=dev)
    w = torch.randn(cout, cin, ks, ks, ks, device=dev) * 0.05
    ms, _ = t(lambda: ops.conv3d(x, w, relu=True, precision="split_bf16"))
    gf = 2.0 * B * cin * cout * ks ** 3 * D ** 3 / 1e9
    out.append("%d->%d k%d D%d %.2f ms (%.0f TF)" % (cin, cout, ks, D, ms, gf / ms))
    tot += cnt * ms
print(os.environ.get("DLPD_LIB_PATH", "default").split("libdlpd_")[-1], "|", "; ".join(out), "| network %.2f ms" % tot)
