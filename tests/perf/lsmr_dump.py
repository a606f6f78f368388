"""Dump the LSMR solution and its info after 16 ... 1000 iterations on the reference sample problem: python tests/perf/lsmr_dump.py OUT.npz
(tests/test_gpu_ba.py runs it once per LSMR form -- DF3D_LSMR_KERNELS = 11 | 2 | 1, the last at several DF3D_LSMR_GRID -- and compares the bits)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deepfly3d_amd import bundle_adjust as ba, _native, ops
from deepfly3d_amd.config import load_calibration
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "golden_2d.npz"))
cal = load_calibration()
c = {k: np.stack([cal[i][k] for i in range(7)]) for k in ("R", "tvec", "intr")}
px = g["points2d"] * np.array([480.0, 960.0])
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)
out = {}
with torch.cuda.stream(side):
    prob = ba.BAProblemDevice(px, c["intr"], dev)
    P = np.einsum("cij,cjk->cik", c["intr"], np.concatenate([c["R"], c["tvec"][..., None]], axis=-1))
    X0 = ops.triangulate(P, torch.from_numpy(np.ascontiguousarray(px)).to(dev))
    cams = np.concatenate([np.stack([ba._rotvec_from_matrix(c["R"][k]) for k in range(7)]), c["tvec"]], axis=1).ravel()
    sel = prob.ok_dev
    x0 = torch.cat([torch.from_numpy(cams).to(dev), X0.reshape(-1, 3)[sel].reshape(-1)])
    dv = ba._Dev(prob)
    m, n, nobs = prob.m, prob.n, prob.nobs
    f, Jc, Jp, g_, sc, sci, tmp = dv.new(m), dv.new(12 * nobs), dv.new(6 * nobs), dv.new(n), dv.new(n), dv.new(n), dv.new(n)
    dv.eval(x0, f, Jc, Jp)
    dv.colsq(Jc, Jp, tmp)
    _native.check(dv.lib.df3d_ba_update_scale(tmp.data_ptr(), sci.data_ptr(), sc.data_ptr(), n, 1, dv.stream()))
    work = dv.new(dv.lib.df3d_ba_lsmr_work_doubles(ctypes.byref(prob.c)))
    xs = dv.new(n)
    for mi in (16, 17, 18, 32, 33, 48, 1000):
        info = dv.lsmr(Jc, Jp, sc, f, 0.37, xs, work, maxiter=mi)
        torch.cuda.synchronize()
        out[f"x{mi}"] = xs.cpu().numpy().copy(); out[f"i{mi}"] = np.array(info)
        print(mi, info[:4])
np.savez(sys.argv[1], **out)
