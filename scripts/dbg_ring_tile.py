"""Development: first plan step where ring_tile=2 differs from ring_tile=1, and where in the tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

views = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
img = torch.rand((views, 256, 512, 3), device=dev)
sd = synthetic_state_dict(0)
for fuse in (False, True):
    a = HourglassEngine(sd, dtype="bf16", device=dev, ring_tile=1, fuse_upadd=fuse)
    b = HourglassEngine(sd, dtype="bf16", device=dev, ring_tile=2, fuse_upadd=fuse)
    steps = a.steps()
    bad = 0
    for k in range(1, len(steps) + 1):
        x, y = a.forward_upto(img, k), b.forward_upto(img, k)
        if not torch.equal(x, y):
            d = (x.float() - y.float()).abs()
            idx = torch.nonzero(d > 0)
            print(f"fuse_upadd={fuse} step {k} {steps[k-1]}: {idx.shape[0]} of {d.numel()} differ, max {d.max().item():.3e}")
            print("  shape", tuple(x.shape), "first", idx[:6].tolist(), "last", idx[-3:].tolist())
            for dim in range(idx.shape[1]):
                u = torch.unique(idx[:, dim])
                print(f"  dim {dim}: {u.numel()} distinct, e.g. {u[:24].tolist()}")
            y2 = b.forward_upto(img, k)
            print("  repeat deterministic:", torch.equal(y, y2))
            bad += 1
            if bad >= 2:
                break
    print(f"fuse_upadd={fuse}: {'differences' if bad else 'identical'}")
