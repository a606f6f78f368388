"""Development probe of the f32s engine (float32 storage, split half-precision products): error against the exact-fp32 engine and the
torch oracle on the test images, then the forward's wall clock on a bench-sized batch beside the f32 engine's.

    python scripts/probe_f32s.py [views] [dtype ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepfly3d_amd.hourglass import HourglassEngine
from oracle import hourglass_torch as oh

views = int(sys.argv[1]) if len(sys.argv) > 1 else 896
dtypes = sys.argv[2:] or ["f32", "f32s"]
dev = torch.device("cuda:0")
net = oh.build(seed=0)
sd = net.state_dict()
x = torch.rand((2, 256, 512, 3), generator=torch.Generator().manual_seed(0), dtype=torch.float32)
traced = oh.forward_traced(net, x)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


engines = {d: HourglassEngine(sd, dtype=d, device=dev) for d in dtypes}
for d, eng in engines.items():
    worst = (0.0, None)
    for k, (name, hwc) in enumerate(eng.steps(), start=1):
        got = eng.forward_upto(x.to(dev), k).cpu()
        e = rel(got, traced["maxpool"] if tuple(got.shape) != tuple(traced[name].shape) else traced[name])
        if e > worst[0]:
            worst = (e, name)
    hm = eng.forward(x.to(dev)).cpu()
    print(f"{d}: heat-maps vs oracle {rel(hm, traced['score.1']):.3e}; worst plan step {worst[0]:.3e} ({worst[1]})", flush=True)
if "f32" in engines:
    ref = engines["f32"].forward(x.to(dev))
    for d, eng in engines.items():
        if d != "f32":
            print(f"{d} vs f32 engine: {rel(eng.forward(x.to(dev)), ref):.3e}")

img = torch.rand((views, 256, 512, 3), device=dev)
for d, eng in engines.items():
    out = torch.empty((views, 19, 64, 128), device=dev)
    for _ in range(3):
        eng.forward(img, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        eng.forward(img, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{d}: {views} views in {dt * 1e3:.2f} ms = {views / 7 / dt:.1f} frames/s (hourglass only)", flush=True)
