"""Timing probe for the hourglass engine (development tool): python scripts/probe_hg.py [dtype] [batch] [row_bytes] [iters]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 56
rb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
eng = HourglassEngine(synthetic_state_dict(0), dtype=dtype, device=dev, row_bytes=rb)
img = torch.rand((batch, 256, 512, 3), device=dev)
out = eng.forward(img)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters):
    eng.forward(img, out=out)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
fl, by = eng.work(batch)
print(f"dtype={dtype} batch={batch} rb={rb}: {dt*1e3:.2f} ms/batch  {batch/dt:.1f} views/s  {batch/dt/7:.1f} frames/s  {fl/dt/1e12:.1f} TFLOP/s  {by/dt/1e9:.0f} GB/s(M1)")
