import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
x = torch.randn(128, 128, 128)
for nt in (torch.get_num_threads(), 8, 32):
    torch.set_num_threads(nt)
    t = time.time()
    for _ in range(20):
        torch.min(x, dim=2)
    t1 = (time.time() - t) / 20
    t = time.time()
    y = torch.randn(8, 128, 128, 128)
    f = torch.fft.rfftn(y, dim=(1, 2, 3))
    t2 = time.time() - t
    print("threads", nt, "min(dim) %.4f s" % t1, "rfftn 8x128^3 %.3f s" % t2, flush=True)
