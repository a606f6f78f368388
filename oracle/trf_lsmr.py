"""CPU oracle for the bundle-adjustment SOLVER: a numpy restatement of the trust-region-reflective
iteration (no bounds) with the LSMR 2-D subspace step that scipy.optimize.least_squares runs for the
reference's configuration  method='trf', jac_sparsity=..., x_scale='jac', ftol=1e-4
(the configuration pyba.bundle_adjust uses; solver trace in reference
notebook/run_df3d.ipynb:74).   TEST INFRASTRUCTURE ONLY.

Differences from oracle/geometry.py:bundle_adjust_scipy (which calls scipy itself and is pinned to the
reference's golden cameras):  the Jacobian here is ANALYTIC (block form: 2x6 camera + 2x3 point per
observation) instead of scipy's 2-point finite differences.  tests/test_oracle_golden.py checks that
both land within the reference's own tolerances of the golden result, which pins this restatement.
The device driver (deepfly3d_amd/bundle_adjust.py + csrc/ba.hip) mirrors THIS file step for step.
"""
import numpy as np

from . import geometry as g

EPS = np.finfo(float).eps


# ------------------------------------------------------------------------------------------------
# analytic residual + Jacobian blocks
# ------------------------------------------------------------------------------------------------
def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rotation_and_dfactor(rvec):
    """R(r) and the 3x3 factor M with  d(R v)/dr = -R [v]x M   (Gallego & Yezzi 2015, eq. 8):
    M = (r r^T + (R^T - I) [r]x) / |r|^2,  M -> I as |r| -> 0."""
    th2 = float(rvec @ rvec)
    R = g.matrix_from_rotvec(rvec)
    if th2 < 1e-24:
        return R, np.eye(3)
    M = (np.outer(rvec, rvec) + (R.T - np.eye(3)) @ _skew(rvec)) / th2
    return R, M


def eval_blocks(x, ncam, intr, cam_idx, pt_idx, obs_xy):
    """r (2n,), Jc (n, 2, 6), Jp (n, 2, 3)."""
    cams = x[: ncam * 6].reshape(ncam, 6)
    pts = x[ncam * 6 :].reshape(-1, 3)
    RM = [rotation_and_dfactor(cams[c, :3]) for c in range(ncam)]
    R = np.stack([rm[0] for rm in RM])[cam_idx]  # (n,3,3)
    M = np.stack([rm[1] for rm in RM])[cam_idx]
    X = pts[pt_idx]
    Xc = np.einsum("nij,nj->ni", R, X) + cams[cam_idx, 3:]
    fx, fy = intr[cam_idx, 0, 0], intr[cam_idx, 1, 1]
    cx, cy = intr[cam_idx, 0, 2], intr[cam_idx, 1, 2]
    iz = 1.0 / Xc[:, 2]
    u = fx * Xc[:, 0] * iz + cx
    v = fy * Xc[:, 1] * iz + cy
    r = np.stack([u - obs_xy[:, 0], v - obs_xy[:, 1]], axis=1).ravel()
    n = cam_idx.size
    dpi = np.zeros((n, 2, 3))
    dpi[:, 0, 0] = fx * iz
    dpi[:, 0, 2] = -fx * Xc[:, 0] * iz * iz
    dpi[:, 1, 1] = fy * iz
    dpi[:, 1, 2] = -fy * Xc[:, 1] * iz * iz
    Xx = np.zeros((n, 3, 3))
    Xx[:, 0, 1], Xx[:, 0, 2] = -X[:, 2], X[:, 1]
    Xx[:, 1, 0], Xx[:, 1, 2] = X[:, 2], -X[:, 0]
    Xx[:, 2, 0], Xx[:, 2, 1] = -X[:, 1], X[:, 0]
    dXc_dr = -np.einsum("nij,njk,nkl->nil", R, Xx, M)
    Jc = np.concatenate([np.einsum("nij,njk->nik", dpi, dXc_dr), dpi], axis=2)
    Jp = np.einsum("nij,njk->nik", dpi, R)
    return r, Jc, Jp


class BlockJacobian:
    """J = [Jc | Jp] in block form with the matvecs the solver needs."""

    def __init__(self, ncam, npts, cam_idx, pt_idx, Jc, Jp):
        self.ncam, self.npts, self.cam_idx, self.pt_idx, self.Jc, self.Jp = ncam, npts, cam_idx, pt_idx, Jc, Jp
        self.shape = (2 * cam_idx.size, 6 * ncam + 3 * npts)

    def matvec(self, v):
        vc = v[: 6 * self.ncam].reshape(self.ncam, 6)[self.cam_idx]
        vp = v[6 * self.ncam :].reshape(self.npts, 3)[self.pt_idx]
        return (np.einsum("nij,nj->ni", self.Jc, vc) + np.einsum("nij,nj->ni", self.Jp, vp)).ravel()

    def rmatvec(self, u):
        u2 = u.reshape(-1, 2)
        wc = np.zeros((self.ncam, 6))
        np.add.at(wc, self.cam_idx, np.einsum("nij,ni->nj", self.Jc, u2))
        wp = np.zeros((self.npts, 3))
        np.add.at(wp, self.pt_idx, np.einsum("nij,ni->nj", self.Jp, u2))
        return np.concatenate([wc.ravel(), wp.ravel()])

    def colsq(self):
        wc = np.zeros((self.ncam, 6))
        np.add.at(wc, self.cam_idx, (self.Jc**2).sum(axis=1))
        wp = np.zeros((self.npts, 3))
        np.add.at(wp, self.pt_idx, (self.Jp**2).sum(axis=1))
        return np.concatenate([wc.ravel(), wp.ravel()])


# ------------------------------------------------------------------------------------------------
# LSMR (Fong & Saunders 2011) on  A = J diag(d)  with damping, as scipy runs it inside TRF
# ------------------------------------------------------------------------------------------------
def sym_ortho(a, b):
    if b == 0:
        return np.sign(a), 0.0, abs(a)
    if a == 0:
        return 0.0, np.sign(b), abs(b)
    if abs(b) > abs(a):
        tau = a / b
        s = np.sign(b) / np.sqrt(1 + tau * tau)
        c = s * tau
        r = b / s
    else:
        tau = b / a
        c = np.sign(a) / np.sqrt(1 + tau * tau)
        s = c * tau
        r = a / c
    return c, s, r


def lsmr(matvec, rmatvec, b, m, n, damp=0.0, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=None):
    if maxiter is None:
        maxiter = min(m, n)
    u = b.copy()
    normb = np.linalg.norm(b)
    x = np.zeros(n)
    beta = normb
    if beta > 0:
        u = u / beta
        v = rmatvec(u)
        alpha = np.linalg.norm(v)
    else:
        v = np.zeros(n)
        alpha = 0.0
    if alpha > 0:
        v = v / alpha
    itn = 0
    zetabar = alpha * beta
    alphabar = alpha
    rho = rhobar = cbar = 1.0
    sbar = 0.0
    h = v.copy()
    hbar = np.zeros(n)
    betadd, betad = beta, 0.0
    rhodold = 1.0
    tautildeold = thetatilde = zeta = d = 0.0
    normA2 = alpha * alpha
    maxrbar, minrbar = 0.0, 1e100
    normA, condA, normx = np.sqrt(normA2), 1.0, 0.0
    istop = 0
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    normr = beta
    normar = alpha * beta
    if normar == 0 or normb == 0:
        return x, istop, itn, normr, normar, normA, condA, normx
    while itn < maxiter:
        itn += 1
        u = matvec(v) - alpha * u
        beta = np.linalg.norm(u)
        if beta > 0:
            u = u / beta
            v = rmatvec(u) - beta * v
            alpha = np.linalg.norm(v)
            if alpha > 0:
                v = v / alpha
        chat, shat, alphahat = sym_ortho(alphabar, damp)
        rhoold = rho
        c, s, rho = sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold, zetaold = rhobar, zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        hbar = h - (thetabar * rho / (rhoold * rhobarold)) * hbar
        x = x + (zeta / (rho * rhobar)) * hbar
        h = v - (thetanew / rho) * h
        betaacute = chat * betadd
        betacheck = -shat * betadd
        betahat = c * betaacute
        betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck * betacheck
        normr = np.sqrt(d + (betad - taud) ** 2 + betadd * betadd)
        normA2 = normA2 + beta * beta
        normA = np.sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)
        normar = abs(zetabar)
        normx = np.linalg.norm(x)
        test1 = normr / normb
        test2 = normar / (normA * normr) if (normA * normr) != 0 else np.inf
        test3 = 1.0 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= maxiter:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop > 0:
            break
    return x, istop, itn, normr, normar, normA, condA, normx


# ------------------------------------------------------------------------------------------------
# 2-D trust-region subproblem and the TRF outer loop (no bounds, linear loss)
# ------------------------------------------------------------------------------------------------
def solve_trust_region_2d(B, gS, Delta):
    """min 0.5 p^T B p + g^T p  s.t. |p| <= Delta,  B 2x2 symmetric."""
    try:
        L = np.linalg.cholesky(B)
        p = -np.linalg.solve(L.T, np.linalg.solve(L, gS))
        if p @ p <= Delta**2:
            return p, True
    except np.linalg.LinAlgError:
        pass
    a, b, c = B[0, 0] * Delta**2, B[0, 1] * Delta**2, B[1, 1] * Delta**2
    d, f = gS[0] * Delta, gS[1] * Delta
    t = np.roots(np.array([-b + d, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - d]))
    t = np.real(t[np.isreal(t)])
    p = Delta * np.vstack((2 * t / (1 + t**2), (1 - t**2) / (1 + t**2)))
    value = 0.5 * np.sum(p * (B @ p), axis=0) + gS @ p
    return p[:, np.argmin(value)], False


def trf_lsmr(evaluate, x0, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, trace=None):
    """evaluate(x, want_jac) -> (f, J or None) with J exposing matvec / rmatvec / colsq / shape."""
    x = x0.copy()
    f, J = evaluate(x, True)
    nfev = njev = 1
    m, n = J.shape
    cost = 0.5 * (f @ f)
    gvec = J.rmatvec(f)
    scale_inv = np.sqrt(J.colsq())
    scale_inv[scale_inv == 0] = 1
    scale = 1.0 / scale_inv
    Delta = np.linalg.norm(x0 * scale_inv)
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = x0.size * 100
    status = None
    lsmr_iters = []
    while True:
        g_norm = np.linalg.norm(gvec, ord=np.inf)
        if g_norm < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        d = scale
        g_h = d * gvec
        mv = lambda v: J.matvec(d * v)  # noqa: E731
        rmv = lambda u: d * J.rmatvec(u)  # noqa: E731
        # Tikhonov term from the 1-D Cauchy model along -g_h
        Jg = mv(-g_h)
        a = 0.5 * (Jg @ Jg)
        b = -(g_h @ g_h)
        to_tr = Delta / np.linalg.norm(g_h)
        cand = [0.0, to_tr]
        if a != 0:
            ext = -0.5 * b / a
            if 0 < ext < to_tr:
                cand.append(ext)
        cand = np.asarray(cand)
        ag_value = np.min(cand * (a * cand + b))
        reg_term = -ag_value / Delta**2
        damp = np.sqrt(reg_term)
        gn_h, istop, itn = lsmr(mv, rmv, f, m, n, damp=damp)[:3]
        lsmr_iters.append(itn)
        S = np.vstack((g_h, gn_h)).T
        S, _ = np.linalg.qr(S, mode="reduced")
        JS = np.stack([mv(S[:, 0]), mv(S[:, 1])], axis=1)
        B_S = JS.T @ JS
        g_S = S.T @ g_h
        actual_reduction = -1.0
        while actual_reduction <= 0 and nfev < max_nfev:
            p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
            step_h = S @ p_S
            Js = mv(step_h)
            predicted_reduction = -(0.5 * (Js @ Js) + step_h @ g_h)
            step = d * step_h
            x_new = x + step
            f_new, _ = evaluate(x_new, False)
            nfev += 1
            step_h_norm = np.linalg.norm(step_h)
            if not np.all(np.isfinite(f_new)):
                Delta = 0.25 * step_h_norm
                continue
            cost_new = 0.5 * (f_new @ f_new)
            actual_reduction = cost - cost_new
            # trust-radius update
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
            step_norm = np.linalg.norm(step)
            ftol_ok = actual_reduction < ftol * cost and ratio > 0.25
            xtol_ok = step_norm < xtol * (xtol + np.linalg.norm(x))
            if ftol_ok and xtol_ok:
                status = 4
            elif ftol_ok:
                status = 2
            elif xtol_ok:
                status = 3
            if status is not None:
                break
            Delta = Delta_new
        if trace is not None:
            trace.append(dict(nfev=nfev, cost=cost, Delta=Delta, lsmr_itn=itn, damp=damp))
        if actual_reduction > 0:
            x = x_new
            cost = cost_new
            f, J = evaluate(x, True)
            njev += 1
            gvec = J.rmatvec(f)
            scale_inv = np.maximum(np.sqrt(J.colsq()), scale_inv)
            scale = 1.0 / scale_inv
    if status is None:
        status = 0
    return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=status, lsmr_iters=lsmr_iters, optimality=g_norm)


def bundle_adjust(points2d_px, R, tvec, intr, return_info=False):
    """Same problem set-up as geometry.bundle_adjust_scipy, solved by the restated TRF+LSMR."""
    R = np.asarray(R, np.float64)
    tvec = np.asarray(tvec, np.float64)
    intr = np.asarray(intr, np.float64)
    ncam = R.shape[0]
    P = g.projection_matrices(R, tvec, intr)
    pts0 = g.triangulate_dlt_batched(points2d_px, P)
    cam_idx, pt_idx, obs_xy, slot = g.build_observations(points2d_px)
    x0 = g.ba_pack(R, tvec, pts0, slot)
    npts = int((slot >= 0).sum())

    def evaluate(x, want_jac):
        if not want_jac:
            return g.ba_residuals(x, ncam, intr, cam_idx, pt_idx, obs_xy), None
        r, Jc, Jp = eval_blocks(x, ncam, intr, cam_idx, pt_idx, obs_xy)
        return r, BlockJacobian(ncam, npts, cam_idx, pt_idx, Jc, Jp)

    res = trf_lsmr(evaluate, x0)
    cams = res["x"][: ncam * 6].reshape(ncam, 6)
    R_new = g.matrix_from_rotvec(cams[:, :3])
    t_new = cams[:, 3:].copy()
    if return_info:
        return R_new, t_new, res
    return R_new, t_new
