"""a7: bundle adjustment driver (extrinsics + 3-D points; intrinsics and distortion frozen).

Replaces the solver under pyba's `CameraNetwork.bundle_adjust(update_intrinsic=False, update_distort=False)`
(call site reference df3d/core.py:249), which is scipy's
    least_squares(method='trf', jac_sparsity=..., x_scale='jac', ftol=1e-4)         (SURVEY.md App. A.3).
Parity with the reference requires reproducing that solver's ITERATE SEQUENCE (the problem has a free 7-DoF
gauge and stops early), so this file mirrors the trust-region-reflective loop of scipy's `trf_no_bounds`
with its LSMR 2-D-subspace step -- the scalar control flow (a handful of iterations on 2x2 systems) runs
here on the host, every vector operation runs in libdf3d_hip.so:

    residuals + analytic Jacobian blocks ....... df3d_ba_eval
    column norms for x_scale='jac' ............. df3d_ba_colsq
    J v, J^T u ................................. df3d_ba_matvec / df3d_ba_rmatvec
    damped LSMR on J diag(d) ................... df3d_ba_lsmr  (device vectors, host scalars)
    dot / axpby / elementwise products ......... df3d_vec_*

There is no CPU fallback.
"""
import ctypes
import os

import numpy as np
import torch

from . import _native
from . import ops


def _rotvec_from_matrix(R):
    """Rodrigues vector of a rotation matrix (float64, host; 7 cameras -> parameter packing only)."""
    R = np.asarray(R, np.float64)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.linalg.norm(w)
    c = 0.5 * (np.trace(R) - 1.0)
    theta = np.arctan2(s, c)
    if s > 1e-8:
        return w / (2.0 * s) * theta
    if c > 0:  # near identity
        return 0.5 * w
    # near pi: axis from the symmetric part
    A = 0.5 * (R + np.eye(3))
    k = int(np.argmax(np.diag(A)))
    axis = A[k] / np.sqrt(A[k, k])
    if w @ axis < 0:
        axis = -axis
    return axis * theta


def _matrix_from_rotvec(r):
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    K = K / th
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _visible_pairs(points2d_px):
    """(T, J) mask of the joints seen by at least two cameras (a camera sees a joint iff neither coordinate is 0)."""
    p = np.asarray(points2d_px, np.float64)
    return ((p[..., 0] != 0) & (p[..., 1] != 0)).sum(axis=0) >= 2


class BAProblemDevice:
    """Observation tables on the device (built once per calibration window) + work buffers.

    Round 4: the tables are BUILT on the device too (torch index operations on the uploaded detections: visibility mask, the
    (frame, joint, camera) observation order, the per-point and per-camera groupings) -- on a 1 000-frame window the numpy version of
    this constructor cost 3 ms of host time, a quarter of the whole adjustment."""

    def __init__(self, points2d_px, intr, device, min_views=2):
        """min_views: a joint is a 3-D point when at least this many cameras see it (2: what an adjustment needs; 1 only in tests of the LSMR
        forms on problems the C ABI accepts but this module never builds)."""
        dev = torch.device(device)
        self.device = dev
        if isinstance(points2d_px, torch.Tensor):
            p = points2d_px.to(device=dev, dtype=torch.float64)
        else:
            p = torch.from_numpy(np.ascontiguousarray(points2d_px, dtype=np.float64)).to(dev)
        ncam, T, J, _ = p.shape
        vis = (p[..., 0] != 0) & (p[..., 1] != 0)                 # (ncam, T, J)
        ok = vis.sum(dim=0) >= int(min_views)                     # (T, J): the joints seen by at least two cameras = the 3-D points
        self.ok_dev = ok.reshape(-1)
        slot = torch.cumsum(self.ok_dev.to(torch.int64), 0) - 1   # point index of (t, j) where ok
        # observations in (frame, joint, camera) order: move the camera axis last; nonzero() lists them in that order
        vis_tjc = vis.permute(1, 2, 0) & ok[..., None]
        idx = torch.nonzero(vis_tjc)
        t_i, j_i, c_i = idx[:, 0], idx[:, 1], idx[:, 2]
        self.ncam, self.nobs = ncam, int(idx.shape[0])
        if self.nobs == 0:
            raise ValueError("bundle adjustment needs at least one joint seen by two cameras")
        cam_idx = c_i.to(torch.int32)
        pt_long = slot[t_i * J + j_i]
        self.npts = int(slot[-1].item()) + 1
        pt_idx = pt_long.to(torch.int32)
        obs = p[c_i, t_i, j_i]                                    # (nobs, 2) (row_px, col_px)
        obs_xy = torch.stack([obs[:, 1], obs[:, 0]], dim=1).contiguous()  # x = col_px, y = row_px
        zero = torch.zeros(1, dtype=torch.int64, device=dev)
        pt_start = torch.cat([zero, torch.cumsum(torch.bincount(pt_long, minlength=self.npts), 0)]).to(torch.int32)
        cam_perm = torch.argsort(c_i, stable=True).to(torch.int32)
        cam_start = torch.cat([zero, torch.cumsum(torch.bincount(c_i, minlength=ncam), 0)]).to(torch.int32)
        intr = np.asarray(intr, np.float64)
        intr4 = torch.from_numpy(np.stack([intr[:, 0, 0], intr[:, 1, 1], intr[:, 0, 2], intr[:, 1, 2]], axis=1)).to(dev)
        self._TJ = (T, J)
        self._slot = None
        self.t = dict(intr4=intr4, obs_xy=obs_xy, cam_idx=cam_idx.contiguous(), pt_idx=pt_idx.contiguous(), pt_start=pt_start.contiguous(),
                      cam_perm=cam_perm.contiguous(), cam_start=cam_start.contiguous())
        self.c = _native.BAProblem(
            ncam, self.nobs, self.npts, self.t["intr4"].data_ptr(), self.t["obs_xy"].data_ptr(), self.t["cam_idx"].data_ptr(),
            self.t["pt_idx"].data_ptr(), self.t["pt_start"].data_ptr(), self.t["cam_perm"].data_ptr(), self.t["cam_start"].data_ptr(),
        )
        self.m = 2 * self.nobs
        self.n = 6 * ncam + 3 * self.npts

    @property
    def slot(self):
        """(T, J) int64: index of the 3-D point of (frame, joint), -1 where fewer than two cameras see the joint (host copy, on demand)."""
        if self._slot is None:
            ok = self.ok_dev.cpu().numpy()
            s = np.full(ok.shape, -1, dtype=np.int64)
            s[ok] = np.arange(int(ok.sum()))
            self._slot = s.reshape(self._TJ)
        return self._slot


class _Dev:
    """Device-vector arithmetic through the C ABI (float64)."""

    def __init__(self, prob):
        self.lib = _native.load()
        self.p = prob
        self.dev = prob.device
        self.scratch = torch.empty(4096, dtype=torch.float64, device=self.dev)
        self._res = ctypes.c_double()

    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def new(self, n):
        return torch.empty(n, dtype=torch.float64, device=self.dev)

    def dot(self, a, b):
        _native.check(self.lib.df3d_vec_dot(a.data_ptr(), b.data_ptr(), a.numel(), ctypes.byref(self._res), self.scratch.data_ptr(), self.stream()), "df3d_vec_dot")
        return self._res.value

    def dots(self, *pairs):
        """Several dot products in one launch and ONE synchronising read-back; each is summed exactly as `dot` sums it."""
        k = len(pairs)
        a = (ctypes.c_void_p * k)(*[x.data_ptr() for x, _ in pairs])
        b = (ctypes.c_void_p * k)(*[y.data_ptr() for _, y in pairs])
        n = (ctypes.c_size_t * k)(*[x.numel() for x, _ in pairs])
        out = (ctypes.c_double * k)()
        _native.check(self.lib.df3d_vec_dots(k, a, b, n, out, self.scratch.data_ptr(), self.stream()), "df3d_vec_dots")
        return list(out)

    def absmax(self, a):
        _native.check(self.lib.df3d_vec_absmax(a.data_ptr(), a.numel(), ctypes.byref(self._res), self.scratch.data_ptr(), self.stream()), "df3d_vec_absmax")
        return self._res.value

    def norm(self, a):
        return float(np.sqrt(self.dot(a, a)))

    def axpby(self, a, x, b, y, out):
        _native.check(self.lib.df3d_vec_axpby(a, x.data_ptr(), b, y.data_ptr() if y is not None else None, out.data_ptr(), x.numel(), self.stream()), "df3d_vec_axpby")
        return out

    def mul(self, x, y, out):
        _native.check(self.lib.df3d_vec_mul(x.data_ptr(), y.data_ptr(), out.data_ptr(), x.numel(), self.stream()), "df3d_vec_mul")
        return out

    def eval(self, x, r=None, Jc=None, Jp=None):
        _native.check(
            self.lib.df3d_ba_eval(ctypes.byref(self.p.c), x.data_ptr(), r.data_ptr() if r is not None else None,
                                  Jc.data_ptr() if Jc is not None else None, Jp.data_ptr() if Jp is not None else None, self.stream()),
            "df3d_ba_eval",
        )

    def colsq(self, Jc, Jp, out):
        _native.check(self.lib.df3d_ba_colsq(ctypes.byref(self.p.c), Jc.data_ptr(), Jp.data_ptr(), out.data_ptr(), self.scratch.data_ptr(), self.stream()), "df3d_ba_colsq")
        return out

    def matvec(self, Jc, Jp, d, v, out):
        _native.check(self.lib.df3d_ba_matvec(ctypes.byref(self.p.c), Jc.data_ptr(), Jp.data_ptr(), d.data_ptr() if d is not None else None, v.data_ptr(), out.data_ptr(), self.stream()), "df3d_ba_matvec")
        return out

    def rmatvec(self, Jc, Jp, d, u, out):
        _native.check(self.lib.df3d_ba_rmatvec(ctypes.byref(self.p.c), Jc.data_ptr(), Jp.data_ptr(), d.data_ptr() if d is not None else None, u.data_ptr(), out.data_ptr(), self.scratch.data_ptr(), self.stream()), "df3d_ba_rmatvec")
        return out

    def lsmr(self, Jc, Jp, d, b, damp, x_out, work, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=0, form=_native.LSMR_AUTO):
        info = (ctypes.c_double * 8)()
        _native.check(
            self.lib.df3d_ba_lsmr_form(ctypes.byref(self.p.c), Jc.data_ptr(), Jp.data_ptr(), d.data_ptr(), b.data_ptr(), damp, atol, btol, conlim, maxiter,
                                       x_out.data_ptr(), work.data_ptr(), info, self.stream(), form),
            "df3d_ba_lsmr_form",
        )
        return list(info)


def _solve_trust_region_2d(B, g, Delta):
    """2-D trust-region subproblem (host, 2x2): Newton step if inside, else the boundary minimiser."""
    try:
        L = np.linalg.cholesky(B)
        p = -np.linalg.solve(L.T, np.linalg.solve(L, g))
        if p @ p <= Delta**2:
            return p
    except np.linalg.LinAlgError:
        pass
    a, b, c = B[0, 0] * Delta**2, B[0, 1] * Delta**2, B[1, 1] * Delta**2
    d, f = g[0] * Delta, g[1] * Delta
    t = np.roots(np.array([-b + d, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - d]))
    t = np.real(t[np.isreal(t)])
    p = Delta * np.vstack((2 * t / (1 + t**2), (1 - t**2) / (1 + t**2)))
    value = 0.5 * np.sum(p * (B @ p), axis=0) + g @ p
    return p[:, np.argmin(value)]


def solve_trf(prob, x0, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, lsmr_form=_native.LSMR_AUTO, device_scalars=None):
    """Trust-region-reflective least squares without bounds, LSMR subspace step, x_scale='jac'.
    x0: device float64 [n].  Returns dict(x=device tensor, cost, nfev, njev, status, lsmr_iters, optimality).
    lsmr_form: DF3D_LSMR_* of include/df3d_hip.h (AUTO: one persistent data-local kernel per inner solve; LAUNCHES when the adjustment
    runs beside other work on the device).
    device_scalars (default True; DF3D_TRF_HOST_SCALARS=1 in the environment turns it off): the driver's scalars stay on the device -- one
    read-back per outer iteration and one per trial step (df3d_ba_trf_* of include/df3d_hip.h) instead of seven and one; the same
    arithmetic, the same iterates (tests/test_gpu_ba.py compares the two)."""
    if device_scalars is None:
        device_scalars = os.environ.get("DF3D_TRF_HOST_SCALARS", "0") in ("", "0")
    if device_scalars:
        return _solve_trf_device_scalars(prob, x0, ftol, xtol, gtol, max_nfev, lsmr_form)
    dv = _Dev(prob)
    m, n, nobs = prob.m, prob.n, prob.nobs
    x = x0.clone()
    f = dv.new(m)
    f_new = dv.new(m)
    Jc = dv.new(12 * nobs)
    Jp = dv.new(6 * nobs)
    g = dv.new(n)
    g_h = dv.new(n)
    gn_h = dv.new(n)
    scale = dv.new(n)
    scale_inv = dv.new(n)
    tmp_n = dv.new(n)
    tmp_n2 = dv.new(n)
    s0 = dv.new(n)
    s1 = dv.new(n)
    step_h = dv.new(n)
    x_new = dv.new(n)
    Js0 = dv.new(m)
    Js1 = dv.new(m)
    tmp_m = dv.new(m)
    work = dv.new(dv.lib.df3d_ba_lsmr_work_doubles(ctypes.byref(prob.c)))

    def refresh_scale(first):
        dv.colsq(Jc, Jp, tmp_n)
        _native.check(dv.lib.df3d_ba_update_scale(tmp_n.data_ptr(), scale_inv.data_ptr(), scale.data_ptr(), n, 1 if first else 0, dv.stream()), "df3d_ba_update_scale")

    dv.eval(x, f, Jc, Jp)
    nfev = njev = 1
    cost = 0.5 * dv.dot(f, f)
    dv.rmatvec(Jc, Jp, None, f, g)
    refresh_scale(True)
    dv.mul(x, scale_inv, tmp_n)
    Delta = dv.norm(tmp_n)
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = n * 100
    status = None
    lsmr_iters = []
    lsmr_fallbacks = 0
    g_norm = None
    while True:
        g_norm = dv.absmax(g)
        if g_norm < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        d = scale
        dv.mul(d, g, g_h)
        # Tikhonov term from the 1-D Cauchy model along -g_h
        dv.matvec(Jc, Jp, d, g_h, tmp_m)
        jg2, gh2 = dv.dots((tmp_m, tmp_m), (g_h, g_h))   # (round 4: the driver's scalars come back in groups, one read-back each)
        a = 0.5 * jg2
        b = -gh2
        to_tr = Delta / np.sqrt(gh2)
        cand = [0.0, to_tr]
        if a != 0:
            ext = -0.5 * b / a
            if 0 < ext < to_tr:
                cand.append(ext)
        cand = np.asarray(cand)
        ag_value = np.min(cand * (a * cand + b))
        reg_term = -ag_value / Delta**2
        damp = float(np.sqrt(reg_term))
        info = dv.lsmr(Jc, Jp, d, f, damp, gn_h, work, form=lsmr_form)
        lsmr_iters.append(int(info[1]))
        lsmr_fallbacks += int(info[7]) != 0
        # orthonormal basis S = qr([g_h, gn_h]) (Householder sign convention of LAPACK: R diagonal < 0)
        n0 = np.sqrt(gh2)
        dv.axpby(-1.0 / n0, g_h, 0.0, None, s0)
        r01 = dv.dot(s0, gn_h)
        dv.axpby(1.0, gn_h, -r01, s0, s1)
        n1 = dv.norm(s1)
        dv.axpby(-1.0 / n1, s1, 0.0, None, s1)
        dv.matvec(Jc, Jp, d, s0, Js0)
        dv.matvec(Jc, Jp, d, s1, Js1)
        b00, b01, b11, gs0, gs1 = dv.dots((Js0, Js0), (Js0, Js1), (Js1, Js1), (s0, g_h), (s1, g_h))
        B_S = np.array([[b00, b01], [b01, b11]])
        g_S = np.array([gs0, gs1])
        actual_reduction = -1.0
        cost_new = cost
        while actual_reduction <= 0 and nfev < max_nfev:
            p_S = _solve_trust_region_2d(B_S, g_S, Delta)
            dv.axpby(float(p_S[0]), s0, float(p_S[1]), s1, step_h)
            # predicted reduction = -(0.5 |J_h step|^2 + step . g_h), with J_h step = p0 Js0 + p1 Js1
            dv.axpby(float(p_S[0]), Js0, float(p_S[1]), Js1, tmp_m)
            dv.mul(d, step_h, tmp_n2)  # step
            dv.axpby(1.0, x, 1.0, tmp_n2, x_new)
            dv.eval(x_new, f_new, None, None)
            nfev += 1
            jp2, sg, sh2, ff, st2, xx = dv.dots((tmp_m, tmp_m), (step_h, g_h), (step_h, step_h), (f_new, f_new), (tmp_n2, tmp_n2), (x, x))
            predicted_reduction = -(0.5 * jp2 + sg)
            step_h_norm = float(np.sqrt(sh2))
            cost_new = 0.5 * ff
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual_reduction = cost - cost_new
            if predicted_reduction > 0:
                ratio = actual_reduction / predicted_reduction
            elif predicted_reduction == actual_reduction == 0:
                ratio = 1
            else:
                ratio = 0
            Delta_new = Delta
            if ratio < 0.25:
                Delta_new = 0.25 * step_h_norm
            elif ratio > 0.75 and step_h_norm > 0.95 * Delta:
                Delta_new = 2.0 * Delta
            step_norm = float(np.sqrt(st2))
            ftol_ok = actual_reduction < ftol * cost and ratio > 0.25
            xtol_ok = step_norm < xtol * (xtol + float(np.sqrt(xx)))
            if ftol_ok and xtol_ok:
                status = 4
            elif ftol_ok:
                status = 2
            elif xtol_ok:
                status = 3
            if status is not None:
                break
            Delta = Delta_new
        if actual_reduction > 0:
            x, x_new = x_new, x
            f, f_new = f_new, f
            cost = cost_new
            dv.eval(x, None, Jc, Jp)
            njev += 1
            dv.rmatvec(Jc, Jp, None, f, g)
            refresh_scale(False)
    if status is None:
        status = 0
    return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=status, lsmr_iters=lsmr_iters, optimality=g_norm, lsmr_fallbacks=lsmr_fallbacks)


def _solve_trf_device_scalars(prob, x0, ftol, xtol, gtol, max_nfev, lsmr_form):
    """solve_trf with the scalars of an outer iteration left on the device (see there): the host keeps the trust radius, the 2x2
    subproblem and the acceptance logic -- everything that decides what is enqueued next."""
    dv = _Dev(prob)
    lib, P, st = dv.lib, ctypes.byref(prob.c), dv.stream()
    m, n, nobs = prob.m, prob.n, prob.nobs
    x = x0.clone()
    f, f_new, Js0, Js1, tmp_m = (dv.new(m) for _ in range(5))
    Jc, Jp = dv.new(12 * nobs), dv.new(6 * nobs)
    g, g_h, gn_h, scale, scale_inv, tmp_n, step, s0, s1, step_h, x_new = (dv.new(n) for _ in range(11))
    work = dv.new(lib.df3d_ba_lsmr_work_doubles(P))
    scratch = dv.scratch

    def linearize(xv, fv, eval_f, first):
        _native.check(lib.df3d_ba_trf_linearize(P, xv.data_ptr(), fv.data_ptr(), 1 if eval_f else 0, Jc.data_ptr(), Jp.data_ptr(), g.data_ptr(), tmp_n.data_ptr(),
                                                scale_inv.data_ptr(), scale.data_ptr(), 1 if first else 0, scratch.data_ptr(), st), "df3d_ba_trf_linearize")

    linearize(x, f, True, True)
    nfev = njev = 1
    dv.mul(x, scale_inv, tmp_n)
    ff0, d2 = dv.dots((f, f), (tmp_n, tmp_n))
    cost = 0.5 * ff0
    Delta = float(np.sqrt(d2))
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = n * 100
    status = None
    lsmr_iters = []
    lsmr_fallbacks = 0
    g_norm = None
    sub = (ctypes.c_double * 19)()
    tri = (ctypes.c_double * 6)()
    while True:
        if status is not None or nfev == max_nfev:
            g_norm = dv.absmax(g)
            if g_norm < gtol:
                status = 1
            break
        _native.check(lib.df3d_ba_trf_subspace(P, Jc.data_ptr(), Jp.data_ptr(), scale.data_ptr(), g.data_ptr(), f.data_ptr(), Delta, g_h.data_ptr(), gn_h.data_ptr(),
                                               s0.data_ptr(), s1.data_ptr(), Js0.data_ptr(), Js1.data_ptr(), tmp_m.data_ptr(), work.data_ptr(), scratch.data_ptr(),
                                               sub, st, lsmr_form), "df3d_ba_trf_subspace")
        g_norm = sub[0]
        if g_norm < gtol:   # (what was enqueued behind |g|_inf is not looked at)
            status = 1
            break
        lsmr_iters.append(int(sub[12]))
        lsmr_fallbacks += int(sub[18]) != 0
        B_S = np.array([[sub[6], sub[7]], [sub[7], sub[8]]])
        g_S = np.array([sub[9], sub[10]])
        actual_reduction = -1.0
        cost_new = cost
        while actual_reduction <= 0 and nfev < max_nfev:
            p_S = _solve_trust_region_2d(B_S, g_S, Delta)
            _native.check(lib.df3d_ba_trf_trial(P, float(p_S[0]), float(p_S[1]), s0.data_ptr(), s1.data_ptr(), Js0.data_ptr(), Js1.data_ptr(), scale.data_ptr(),
                                                x.data_ptr(), g_h.data_ptr(), step_h.data_ptr(), tmp_m.data_ptr(), step.data_ptr(), x_new.data_ptr(), f_new.data_ptr(),
                                                scratch.data_ptr(), tri, st), "df3d_ba_trf_trial")
            nfev += 1
            jp2, sg, sh2, ff, st2, xx = tri
            predicted_reduction = -(0.5 * jp2 + sg)
            step_h_norm = float(np.sqrt(sh2))
            cost_new = 0.5 * ff
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual_reduction = cost - cost_new
            if predicted_reduction > 0:
                ratio = actual_reduction / predicted_reduction
            elif predicted_reduction == actual_reduction == 0:
                ratio = 1
            else:
                ratio = 0
            Delta_new = Delta
            if ratio < 0.25:
                Delta_new = 0.25 * step_h_norm
            elif ratio > 0.75 and step_h_norm > 0.95 * Delta:
                Delta_new = 2.0 * Delta
            step_norm = float(np.sqrt(st2))
            ftol_ok = actual_reduction < ftol * cost and ratio > 0.25
            xtol_ok = step_norm < xtol * (xtol + float(np.sqrt(xx)))
            if ftol_ok and xtol_ok:
                status = 4
            elif ftol_ok:
                status = 2
            elif xtol_ok:
                status = 3
            if status is not None:
                break
            Delta = Delta_new
        if actual_reduction > 0:
            x, x_new = x_new, x
            f, f_new = f_new, f
            cost = cost_new
            linearize(x, f, False, False)
            njev += 1
    if status is None:
        status = 0
    return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=status, lsmr_iters=lsmr_iters, optimality=g_norm, lsmr_fallbacks=lsmr_fallbacks)


def reprojection_error(points2d_px, points3d, R, tvec, intr, device="cuda:0"):
    """Mean pixel distance between the observations and the re-projected 3-D joints, over every observation of a
    joint seen by >= 2 cameras (what `CameraNetwork.reprojection_error()` prints, call site reference
    df3d/core.py:250).  The residuals are df3d_ba_eval's (the bundle adjustment's own cost terms), the mean is a
    fixed-order device reduction (df3d_vec_pairnorm_sum)."""
    _native.require_gpu()
    dev = torch.device(device)
    if not np.any(_visible_pairs(points2d_px)):
        # a recording without a single joint seen by two cameras: the reference only prints this number (core.py:250)
        from . import logger

        logger.warning("reprojection error: no joint is seen by two cameras, nothing to average")
        return float("nan")
    with torch.cuda.device(dev):
        prob = BAProblemDevice(points2d_px, intr, dev)
        dv = _Dev(prob)
        ncam = prob.ncam
        cams = np.concatenate([np.stack([_rotvec_from_matrix(np.asarray(R[c], np.float64)) for c in range(ncam)]), np.asarray(tvec, np.float64)], axis=1).ravel()
        X = torch.from_numpy(np.ascontiguousarray(points3d, dtype=np.float64)).to(dev).reshape(-1, 3)[prob.ok_dev].reshape(-1)
        x = torch.cat([torch.from_numpy(cams).to(dev), X])
        r = dv.new(prob.m)
        dv.eval(x, r, None, None)
        _native.check(dv.lib.df3d_vec_pairnorm_sum(r.data_ptr(), prob.nobs, ctypes.byref(dv._res), dv.scratch.data_ptr(), dv.stream()), "df3d_vec_pairnorm_sum")
        return dv._res.value / prob.nobs


_side_streams = {}


def bundle_adjust(points2d_px, R, tvec, intr, device="cuda:0", return_info=False, concurrent=False):
    """See _bundle_adjust; runs with `device` as the current HIP device (kernels launch on the current device).
    concurrent=True: the adjustment shares the device with other work (a window's re-calibration beside the frame pipeline): its inner
    solves take the launch-based LSMR form, which needs no co-resident workgroups; the default -- the CLI's one adjustment per folder
    (reference df3d/core.py:249) -- takes one persistent kernel per solve.  When the
    caller's current stream is the legacy default stream the solve runs on a private stream ordered behind it: the LSMR
    chunks are replayed from a HIP graph, and a graph cannot be recorded on the default stream."""
    _native.require_gpu()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        cur = torch.cuda.current_stream(dev)
        if cur.cuda_stream != 0:
            return _bundle_adjust(points2d_px, R, tvec, intr, device, return_info, concurrent)
        side = _side_streams.get(str(dev))
        if side is None:
            side = _side_streams[str(dev)] = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = _bundle_adjust(points2d_px, R, tvec, intr, device, return_info, concurrent)
        cur.wait_stream(side)
        return out


def _bundle_adjust(points2d_px, R, tvec, intr, device, return_info, concurrent=False):
    """points2d_px (ncam, T, J, 2) float64 (row_px, col_px); R (ncam,3,3), tvec (ncam,3), intr (ncam,3,3).
    Returns adjusted (R, tvec) as float64 numpy arrays (+ solver info)."""
    _native.require_gpu()
    R = np.asarray(R, np.float64)
    tvec = np.asarray(tvec, np.float64)
    intr = np.asarray(intr, np.float64)
    ncam = R.shape[0]
    dev = torch.device(device)
    px_dev = torch.from_numpy(np.ascontiguousarray(points2d_px, dtype=np.float64)).to(dev)
    prob = BAProblemDevice(px_dev, intr, dev)
    # initial points: DLT with the initial calibration (HIP kernel)
    P = np.einsum("cij,cjk->cik", intr, np.concatenate([R, tvec[..., None]], axis=-1))
    X0 = ops.triangulate(P, px_dev)
    cams = np.concatenate([np.stack([_rotvec_from_matrix(R[c]) for c in range(ncam)]), tvec], axis=1).ravel()
    x0 = torch.cat([torch.from_numpy(cams).to(dev), X0.reshape(-1, 3)[prob.ok_dev].reshape(-1)])
    res = solve_trf(prob, x0, lsmr_form=_native.LSMR_LAUNCHES if concurrent else _native.LSMR_AUTO)
    cams_new = res["x"][: 6 * ncam].cpu().numpy().reshape(ncam, 6)
    R_new = np.stack([_matrix_from_rotvec(cams_new[c, :3]) for c in range(ncam)])
    t_new = cams_new[:, 3:].copy()
    if return_info:
        return R_new, t_new, res
    return R_new, t_new
