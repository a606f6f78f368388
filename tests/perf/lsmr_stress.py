"""Run-to-run reproducibility of the LSMR forms on a T-frame window: python tests/perf/lsmr_stress.py [frames] [runs] [maxiter]
(every run from the same inputs; reports how many runs differ from the first in the solution bits or in the info vector)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deepfly3d_amd import bundle_adjust as ba, _native, ops
from deepfly3d_amd.synthetic import synthetic_points2d
from deepfly3d_amd.config import load_calibration

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
MAXIT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "golden_3d.npz"))
cal = load_calibration()
c = {k: np.stack([cal[i][k] for i in range(7)]) for k in ("R", "tvec", "intr")}
rng = np.random.default_rng(0)
X = np.tile(g3["points3d_wo_procrustes"], (T // 15 + 1, 1, 1))[:T] + rng.normal(0, 0.05, size=(T, 38, 3))
px = synthetic_points2d(X, g3["R"], g3["tvec"], g3["intr"]) * np.array([480.0, 960.0])
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    prob = ba.BAProblemDevice(px, c["intr"], dev)
    P = np.einsum("cij,cjk->cik", c["intr"], np.concatenate([c["R"], c["tvec"][..., None]], axis=-1))
    X0 = ops.triangulate(P, torch.from_numpy(np.ascontiguousarray(px)).to(dev))
    cams = np.concatenate([np.stack([ba._rotvec_from_matrix(c["R"][k]) for k in range(7)]), c["tvec"]], axis=1).ravel()
    x0 = torch.cat([torch.from_numpy(cams).to(dev), X0.reshape(-1, 3)[prob.ok_dev].reshape(-1)])
    dv = ba._Dev(prob)
    m, n, nobs = prob.m, prob.n, prob.nobs
    f, Jc, Jp, sc, sci, tmp = dv.new(m), dv.new(12 * nobs), dv.new(6 * nobs), dv.new(n), dv.new(n), dv.new(n)
    dv.eval(x0, f, Jc, Jp)
    dv.colsq(Jc, Jp, tmp)
    _native.check(dv.lib.df3d_ba_update_scale(tmp.data_ptr(), sci.data_ptr(), sc.data_ptr(), n, 1, dv.stream()))
    work = dv.new(dv.lib.df3d_ba_lsmr_work_doubles(ctypes.byref(prob.c)))
    xs = dv.new(n)
    first, bad_x, bad_i, seen = None, 0, 0, {}
    for r in range(RUNS):
        info = dv.lsmr(Jc, Jp, sc, f, 0.37, xs, work, maxiter=MAXIT)
        torch.cuda.synchronize()
        got = (xs.cpu().numpy().copy(), np.array(info))
        if first is None:
            first = got
        else:
            bad_x += not np.array_equal(got[0], first[0])
            bad_i += not np.array_equal(got[1], first[1])
        seen[(int(info[0]), int(info[1]), int(info[7]))] = seen.get((int(info[0]), int(info[1]), int(info[7])), 0) + 1
print(f"T={T} nobs={nobs} runs={RUNS} maxiter={MAXIT}: runs whose solution differs from the first: {bad_x}, whose info differs: {bad_i}; (istop, itn, fallback) seen: {seen}")
