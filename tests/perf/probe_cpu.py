import os, time, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hourglass_torch as oh
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
net = oh.build(0)
x = torch.rand(7, 256, 512, 3)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    oh.forward_nhwc(net, x[:1])
    t = time.time(); oh.forward_nhwc(net, x); dt = time.time() - t
    print(th, "threads:", round(dt, 2), "s per frame")
