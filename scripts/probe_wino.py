"""Development probe: per-phase cycle breakdown of bottleneck_wino_f32_kernel, wave 0 of every workgroup (needs the timing build:
bash scripts/build_variant.sh timing -DDF3D_BT_TIMING; DF3D_LIB=scratch/variants/libdf3d_hip_timing.so python scripts/probe_wino.py [views])"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd import _native
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

views = int(sys.argv[1]) if len(sys.argv) > 1 else 896
dev = torch.device("cuda:0")
eng = HourglassEngine(synthetic_state_dict(0), dtype="f32", device=dev, wino=1)
lib = _native.load()
img = torch.rand((views, 256, 512, 3), device=dev)
steps = eng.steps()
names = [n for n, _ in steps]
buf = (ctypes.c_ulonglong * 12)()
labels = ["prologue / tile entry", "first transform", "phase 2 (16 chunks)", "output transform + residual requests", "t2 crossing", "phase3a K", "epilogue a", "phase3b K", "epilogue b"]
eng.forward(img); torch.cuda.synchronize()
lib.df3d_dbg_ring_cycles.argtypes = [ctypes.c_void_p]


def cycles_upto(k):
    lib.df3d_dbg_ring_cycles(buf)
    eng.forward_upto(img, k); torch.cuda.synchronize()
    lib.df3d_dbg_ring_cycles(buf)
    return list(buf)


for target in ("layer3.0.conv3", "hg.0.hg.3.upadd", "res.0.0.conv3", "hg.0.hg.2.upadd"):
    k = names.index(target) + 1
    before, after = cycles_upto(k - 1), cycles_upto(k)
    own = [a - b for a, b in zip(after, before)]
    hwc = steps[k - 1][1]
    tiles = views * (hwc[0] // 8) * (hwc[1] // 16)
    tot = sum(own)
    print(f"{target} {hwc}: {tiles} tiles, wave-0 ticks per tile {tot / tiles:.0f}  (MFMA cycles per tile: phase 2 65 536, phase 3 2 x 16 384)")
    for l, v in zip(labels, own):
        print(f"   {l:38s} {v / tiles:9.0f}  {100.0 * v / tot:5.1f} %")
