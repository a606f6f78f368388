"""Where a bundle adjustment's wall time goes: python tests/perf/probe_ba_sections.py [frames]
(host sections of deepfly3d_amd.bundle_adjust timed with a device synchronisation behind each; the LSMR calls and every other C-ABI
call counted and timed through a wrapper around the ctypes library)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deepfly3d_amd import bundle_adjust as ba, _native, ops
from deepfly3d_amd.synthetic import synthetic_points2d
from deepfly3d_amd.config import load_calibration

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g3 = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "golden_3d.npz"))
cal = load_calibration()
c = {k: np.stack([cal[i][k] for i in range(7)]) for k in ("R", "tvec", "intr")}
rng = np.random.default_rng(0)
X = np.tile(g3["points3d_wo_procrustes"], (T // 15 + 1, 1, 1))[:T] + rng.normal(0, 0.05, size=(T, 38, 3))
px = synthetic_points2d(X, g3["R"], g3["tvec"], g3["intr"]) * np.array([480.0, 960.0])
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)

calls = collections.defaultdict(lambda: [0, 0.0])
lib = _native.load()


class Timed:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("df3d_"):
            return fn

        def wrapped(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            calls[name][0] += 1
            calls[name][1] += time.perf_counter() - t0
            return r

        return wrapped


for rep in range(3):
    calls.clear()
    with torch.cuda.stream(side):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prob = ba.BAProblemDevice(px, c["intr"], dev)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        P = np.einsum("cij,cjk->cik", c["intr"], np.concatenate([c["R"], c["tvec"][..., None]], axis=-1))
        X0 = ops.triangulate(P, torch.from_numpy(np.ascontiguousarray(px)).to(dev))
        cams = np.concatenate([np.stack([ba._rotvec_from_matrix(c["R"][k]) for k in range(7)]), c["tvec"]], axis=1).ravel()
        sel = prob.ok_dev
        x0 = torch.cat([torch.from_numpy(cams).to(dev), X0.reshape(-1, 3)[sel].reshape(-1)])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        real = ba._Dev.__init__

        def patched(self, prob):
            real(self, prob)
            self.lib = Timed(self.lib)

        ba._Dev.__init__ = patched
        res = ba.solve_trf(prob, x0)
        ba._Dev.__init__ = real
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"T={T}: problem tables {1e3*(t1-t0):.2f} ms | DLT + x0 {1e3*(t2-t1):.2f} ms | solve_trf {1e3*(t3-t2):.2f} ms (nfev {res['nfev']}, lsmr {res['lsmr_iters']})")
    if rep == 2:
        tot = sum(v[1] for v in calls.values())
        for k, (n, t) in sorted(calls.items(), key=lambda kv: -kv[1][1]):
            print(f"    {k:28s} {n:4d} calls {1e3*t:8.2f} ms")
        print(f"    C-ABI calls in all {1e3*tot:.2f} ms; the rest of solve_trf is Python")
