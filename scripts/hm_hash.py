"""Development: SHA-256 of the heat-maps (and of every plan step) for a fixed seeded input -- to confirm that a kernel
rewrite which claims the same arithmetic really is bit-identical to the build before it."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda:0")
sd = synthetic_state_dict(0)
for dtype in sys.argv[1:] or ["bf16", "f32"]:
    for (h, w, n) in ((256, 512, 3), (64, 192, 2)):
        img = torch.rand((n, h, w, 3), generator=torch.Generator().manual_seed(h + w), dtype=torch.float32).to(dev)
        eng = HourglassEngine(sd, dtype=dtype, device=dev, height=h, width=w)
        hm = eng.forward(img)
        hs = hashlib.sha256(hm.cpu().numpy().tobytes()).hexdigest()[:16]
        steps = eng.steps()
        parts = []
        for k in range(1, len(steps) + 1):
            if steps[k - 1][0].startswith("score"):
                parts.append(steps[k - 1][0] + ":" + hashlib.sha256(eng.forward_upto(img, k).cpu().numpy().tobytes()).hexdigest()[:8])
        print(dtype, (h, w, n), "heat-maps", hs, " ".join(parts))
