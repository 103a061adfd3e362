"""Diagnostic: dlpd_conv3d vs torch/MIOpen conv3d on the layer shapes of E3MultiResRepr4x4(multiplier=8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
entry.build()
from deeplocalproteindocking_amd import ops
dev = torch.device("cuda:0")
B = 16
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3, r
tot_h = tot_t = 0.0
for (cin, cout, ks, D, cnt) in ((11, 16, 5, 80, 1), (16, 16, 3, 80, 3), (16, 16, 5, 80, 1), (16, 32, 5, 40, 1), (32, 32, 3, 40, 3)):
    x = torch.randn(B, cin, D, D, D, device=dev)
    w = torch.randn(cout, cin, ks, ks, ks, device=dev) * 0.05
    mt, want = t(lambda: torch.relu(torch.nn.functional.conv3d(x, w, padding=ks // 2)))
    mh, got = t(lambda: ops.conv3d(x, w, relu=True, precision="f32"))
    ms, gots = t(lambda: ops.conv3d(x, w, relu=True, precision="split_bf16"))
    ref = torch.relu(torch.nn.functional.conv3d(x[:2].double().cpu(), w.double().cpu(), padding=ks // 2))
    err = float((got[:2].cpu().double() - ref).abs().max() / ref.abs().max())
    errs = float((gots[:2].cpu().double() - ref).abs().max() / ref.abs().max())
    gf = 2.0 * B * cin * cout * ks ** 3 * D ** 3 / 1e9
    print("%2d->%2d k%d D%d: f32 %.2f ms (%.1f TF, err %.1e)  split-bf16 %.2f ms (%.1f TF f32-equivalent, err %.1e)  torch %.2f ms (%.1f TF)" % (
        cin, cout, ks, D, mh, gf / mh, err, ms, gf / ms, errs, mt, gf / mt))
    tot_h += cnt * mh; tot_t += cnt * mt; tot_s = globals().get("tot_s", 0.0) + cnt * ms
print("network (9 convs): f32 %.1f ms, split-bf16 %.1f ms, torch %.1f ms per batch of %d" % (tot_h, tot_s, tot_t, B))
