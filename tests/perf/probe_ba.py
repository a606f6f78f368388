"""BA timing probe: python tests/perf/probe_ba.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deepfly3d_amd.bundle_adjust import bundle_adjust
from deepfly3d_amd.synthetic import synthetic_points2d
from deepfly3d_amd.config import load_calibration
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g3 = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "golden_3d.npz"))
cal = load_calibration()
c = {k: np.stack([cal[i][k] for i in range(7)]) for k in ("R", "tvec", "intr")}
rng = np.random.default_rng(0)
X = np.tile(g3["points3d_wo_procrustes"], (T // 15 + 1, 1, 1))[:T] + rng.normal(0, 0.05, size=(T, 38, 3))
px = synthetic_points2d(X, g3["R"], g3["tvec"], g3["intr"]) * np.array([480.0, 960.0])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    R, t, info = bundle_adjust(px, c["R"], c["tvec"], c["intr"], device="cuda:0", return_info=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"T={T}: {dt*1e3:.1f} ms  nfev={info['nfev']} lsmr={info['lsmr_iters']} cost={info['cost']:.4f}")
if len(sys.argv) > 2:
    from oracle import geometry as og
    t0 = time.time(); og.bundle_adjust_scipy(px, c["R"], c["tvec"], c["intr"]); print("scipy CPU:", time.time() - t0, "s")
