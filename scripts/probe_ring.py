"""Development probe: per-phase cycle breakdown of bottleneck_ring_kernel (needs the timing build:
bash scripts/build_variant.sh timing -DDF3D_BT_TIMING; DF3D_LIB=scratch/variants/libdf3d_hip_timing.so python scripts/probe_ring.py [views])"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepfly3d_amd import _native
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

views = int(sys.argv[1]) if len(sys.argv) > 1 else 896
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
eng = HourglassEngine(synthetic_state_dict(0), dtype=dtype, device=dev)
lib = _native.load()
img = torch.rand((views, 256, 512, 3), device=dev)
steps = eng.steps()
names = [n for n, _ in steps]
buf = (ctypes.c_ulonglong * 12)()
labels = ["prologue", "phase1 K loop", "t1 epilogue", "t2 crossing (W2D) / phase2", "phase3a K", "epilogue a", "phase3b K", "epilogue b", "phase2 loop (W2D)", "-", "-", "-"]
if dtype in ("f32", "f32s"):  # both t1 halves are summed into the phase-1 / phase-2 slots ("phase2" = second-half entry barrier + both phase 2)
    labels = ["prologue", "phase1 K loops (2)", "t1 epilogues (2)", "phase2 (2) + entry", "phase3a K", "epilogue a", "phase3b K", "epilogue b", "-", "-", "-", "-"]
eng.forward(img); torch.cuda.synchronize()
lib.df3d_dbg_ring_cycles.argtypes = [ctypes.c_void_p]
lib.df3d_dbg_ring_cycles(buf)
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
    print(f"{target} {hwc}: {tiles} tiles, wave-0 cycles per tile {tot / tiles:.0f}")
    for l, v in zip(labels, own):
        if l == "-": continue
        print(f"   {l:14s} {v / tiles:9.0f} cycles  {100.0 * v / tot:5.1f} %")
