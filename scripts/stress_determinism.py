"""Stress: repeated forwards of both engines at several batch sizes must reproduce their first heat-maps bit for bit (the
weight rings rely on counted vector-memory waits: a wrong count shows up as a rare, timing-dependent difference)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda:0")
sd = synthetic_state_dict(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for dtype in ("f16", "bf16", "f32", "f32s"):
    eng = HourglassEngine(sd, dtype=dtype, device=dev)
    for n in (1, 7, 35, 120):
        img = torch.rand((n, 256, 512, 3), generator=torch.Generator().manual_seed(n), dtype=torch.float32).to(dev)
        ref = eng.forward(img).clone()
        # a second engine's launches on another stream keep the memory system busy while the first repeats
        other = HourglassEngine(sd, dtype="f16" if dtype == "f32" else "f32", device=dev)
        side = torch.cuda.Stream()
        noise = torch.rand((21, 256, 512, 3), device=dev)
        diffs = 0
        for r in range(reps):
            with torch.cuda.stream(side):
                other.forward(noise)
            out = eng.forward(img)
            if not torch.equal(out, ref):
                diffs += 1
        torch.cuda.synchronize()
        print(f"{dtype} n={n}: {diffs} of {reps} repeats differ")
        bad += diffs
        del other
print("network:", "FAILED" if bad else "OK")

# the JPEG front-end under the same kind of load: one batch of camera frames decoded again and again on a side stream while
# the network runs (the parallel Huffman kernel exchanges chunk states through LDS and hands DC offsets to the IDCT through
# the per-file descriptor; the pipeline of inference_folder runs exactly this overlap)
import glob
from deepfly3d_amd import jpeg
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(here, "tests/golden/images/*.jpg")))]
blobs = [base[i % len(base)] for i in range(448)]
ref = jpeg.decode_luma(blobs, 960, 480).clone()
eng = HourglassEngine(sd, dtype="bf16", device=dev)
noise = torch.rand((224, 256, 512, 3), device=dev)
side = torch.cuda.Stream()
diffs = 0
for r in range(reps):
    eng.forward(noise)
    with torch.cuda.stream(side):
        out = jpeg.decode_luma(blobs, 960, 480, check=False)
    side.synchronize()
    if not torch.equal(out, ref):
        diffs += 1
torch.cuda.synchronize()
print(f"jpeg decode of 448 frames beside the network: {diffs} of {reps} repeats differ")
bad += diffs
print("OK" if bad == 0 else f"FAILED: {bad} differing repeats")
sys.exit(1 if bad else 0)
