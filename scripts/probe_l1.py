"""Development probe: per-phase cycle breakdown of bottleneck_l1_kernel (timing build: bash scripts/build_variant.sh;
DF3D_LIB=scratch/variants/libdf3d_hip_timing.so python scripts/probe_l1.py [views])"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd import _native
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

views = int(sys.argv[1]) if len(sys.argv) > 1 else 896
dev = torch.device("cuda:0")
eng = HourglassEngine(synthetic_state_dict(0), dtype="bf16", device=dev)
lib = _native.load()
img = torch.rand((views, 256, 512, 3), device=dev)
names = [n for n, _ in eng.steps()]
buf = (ctypes.c_ulonglong * 8)()
lib.df3d_dbg_ring_cycles.argtypes = [ctypes.c_void_p]
k = names.index("layer1.0.conv3") + 1
eng.forward_upto(img, k); torch.cuda.synchronize()
def cycles_upto(j):
    lib.df3d_dbg_ring_cycles(buf)
    eng.forward_upto(img, j); torch.cuda.synchronize()
    lib.df3d_dbg_ring_cycles(buf)
    return list(buf)
before, after = cycles_upto(k - 1), cycles_upto(k)
own = [a - b for a, b in zip(after, before)]
tiles = views * (128 // 16) * (256 // 16)
labels = ["tile top (masks, wait)", "phase 1 (x arrive, act, MFMA)", "barrier B3", "t1 epilogue", "barrier B1", "phase 2 (+ x requests)", "barrier B2", "phase 3 + epilogues"]
labels = ["tile top", "phase 1", "t1 epilogue", "barrier B1", "phase 2", "barrier B2", "phase 3 + epilogues", "-"]
tot = sum(own)
print(f"layer1: {tiles} tiles, wave-0 cycles per tile {tot / tiles:.0f}")
for l, v in zip(labels, own):
    print(f"   {l:22s} {v / tiles:9.0f} cycles  {100.0 * v / max(tot, 1):5.1f} %")
