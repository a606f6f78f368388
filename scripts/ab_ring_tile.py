"""Development A/B: network forward time with 8 x 16 (ring_tile=1) vs 16 x 16 (ring_tile=2) ring-bottleneck tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

views = int(sys.argv[1]) if len(sys.argv) > 1 else 896
dev = torch.device("cuda:0")
img = torch.rand((views, 256, 512, 3), device=dev)
sd = synthetic_state_dict(0)
outs = {}
for rt in (1, 2, 1, 2):
    eng = HourglassEngine(sd, dtype="bf16", device=dev, ring_tile=rt)
    for _ in range(2):
        out = eng.forward(img)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = eng.forward(img)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    outs[rt] = out.clone()
    print(f"ring_tile={rt}: {ms:.2f} ms / {views} views = {views / 7 / ms * 1e3:.0f} frames/s")
    del eng
print("bit-identical:", torch.equal(outs[1], outs[2]))
