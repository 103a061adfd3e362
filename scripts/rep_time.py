import sys, time; sys.path.insert(0, "/root/repo")
import torch
from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4, SE3MultiResReprScalar
dev = torch.device("cuda:0")
x = torch.rand(16, 11, 80, 80, 80, device=dev)
for name, M in (("E3", E3MultiResRepr4x4), ("SE3", SE3MultiResReprScalar)):
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        m = M(multiplier=8).to(dev).eval()
        with torch.no_grad():
            for _ in range(2): m(x)
            torch.cuda.synchronize(); t = time.time()
            for _ in range(3): m(x)
            torch.cuda.synchronize()
        print(name, "benchmark", bench, "%.1f ms per batch of 16" % ((time.time() - t) / 3 * 1e3))
        if name == "E3":
            mc = m.to(memory_format=torch.channels_last_3d); xc = x.contiguous(memory_format=torch.channels_last_3d)
            with torch.no_grad():
                for _ in range(2): mc(xc)
                torch.cuda.synchronize(); t = time.time()
                for _ in range(3): mc(xc)
                torch.cuda.synchronize()
            print(name, "channels_last_3d benchmark", bench, "%.1f ms" % ((time.time() - t) / 3 * 1e3))
