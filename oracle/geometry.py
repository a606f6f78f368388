"""CPU oracle for the geometry half of the DeepFly3D hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (deepfly3d_amd/) never does.

The arithmetic of this path lives in two third-party packages that are NOT vendored in the
reference checkout: `nely-pyba >= 0.13` (reference setup.py:30) and `nely-df2d >= 0.14`
(reference setup.py:31).  What is restated here is therefore anchored on the reference's own call
sites and on its committed golden pickles (tests/data/reference_df3d/*.pkl, data/*.pkl), which
tests/test_oracle_golden.py reproduces:

  heatmap_argmax            reference README.md:404 ("argmax_{h,w} H ... H[h,w] for the confidence")
  relayout_19_to_38         reference df3d/core.py:187-203     (pinned by tests/golden/relayout_*.npz,
                                                                made by executing the reference lines)
  pixels_from_normalised    reference df3d/core.py:247          (points2d * image_shape[::-1])
  triangulate_dlt           call site df3d/core.py:355 (pyba CameraNetwork.triangulate); pinned by
                            golden points3d_wo_procrustes to ~1e-13 given the golden cameras
  bundle_adjust_scipy       call site df3d/core.py:249 (pyba bundle_adjust(update_intrinsic=False,
                            update_distort=False)); scipy.optimize.least_squares(method='trf',
                            jac_sparsity, x_scale='jac', ftol=1e-4) -- the solver trace is visible in
                            reference notebook/run_df3d.ipynb:74; pinned by golden cameras/points
  procrustes_separate       reference df3d/procrustes.py:51-151,154-263 and plot_util.py:85-91;
                            pinned by tests/golden/procrustes_*.npz (made by executing the reference)
"""
import numpy as np

NUM_CAMERAS = 7
NUM_PREDICT = 19  # network outputs per view   (reference df3d/config.py:36)
NUM_JOINTS = 38  # skeleton joints            (reference df3d/skeleton_fly.py:16-55)


# --------------------------------------------------------------------------------------------
# a3: heat-map -> point + confidence
# --------------------------------------------------------------------------------------------
def heatmap_argmax(hm):
    """hm: (N, J, H, W) float32 -> points (N, J, 2) float32 as (row/H, col/W), conf (N, J) float32.

    Hard arg-max; ties resolve to the first index in row-major order (np.argmax semantics)."""
    hm = np.asarray(hm, dtype=np.float32)
    n, j, h, w = hm.shape
    flat = hm.reshape(n, j, h * w)
    idx = flat.argmax(axis=-1)
    conf = np.take_along_axis(flat, idx[..., None], axis=-1)[..., 0]
    pts = np.stack([(idx // w).astype(np.float32) / np.float32(h), (idx % w).astype(np.float32) / np.float32(w)], axis=-1)
    return pts.astype(np.float32), conf.astype(np.float32)


# --------------------------------------------------------------------------------------------
# a4: 19 -> 38 joint re-layout and un-flip   (reference df3d/core.py:187-203)
# --------------------------------------------------------------------------------------------
def relayout_19_to_38(points2d, camera_ordering):
    """points2d: (7, T, 19, 2) normalised (row, col) -> (7, T, 38, 2) float64."""
    o = np.asarray(camera_ordering)
    p = np.asarray(points2d)
    out = np.zeros((p.shape[0], p.shape[1], p.shape[2] * 2, 2))
    out[o[:3], :, :NUM_PREDICT] = p[o[:3]]
    out[o[4:], :, NUM_PREDICT:] = p[o[4:]]
    out[o[2], :, 15:] = 0
    out[o[4], :, NUM_PREDICT + 15 :] = 0
    for cidx in (4, 5, 6):
        out[o[cidx], ..., 1] = 1 - out[o[cidx], ..., 1]
    return out


def pixels_from_normalised(points2d, image_shape):
    """reference df3d/core.py:247 -- image_shape is [W, H]; result is (row_px, col_px)."""
    return np.asarray(points2d, dtype=np.float64) * np.asarray(image_shape[::-1], dtype=np.float64)


# --------------------------------------------------------------------------------------------
# a5/a6: cameras and DLT triangulation
# --------------------------------------------------------------------------------------------
def projection_matrices(R, tvec, intr):
    """P_c = K_c [R_c | t_c], (7, 3, 4) float64, pixel units, no normalisation."""
    Rt = np.concatenate([np.asarray(R, np.float64), np.asarray(tvec, np.float64)[..., None]], axis=-1)
    return np.einsum("cij,cjk->cik", np.asarray(intr, np.float64), Rt)


def visibility(points2d_px):
    """A camera contributes to (t, j) iff NEITHER coordinate is 0."""
    p = np.asarray(points2d_px)
    return (p[..., 0] != 0) & (p[..., 1] != 0)


def triangulate_dlt(points2d_px, P):
    """points2d_px: (7, T, J, 2) as (row_px, col_px); P: (7, 3, 4) -> (T, J, 3) float64.

    Per (t, j): rows x*P[2]-P[0], y*P[2]-P[1] with x = col_px, y = row_px for every visible camera
    (>= 2 needed); X = last right-singular vector of A, de-homogenised; otherwise 0."""
    p = np.asarray(points2d_px, np.float64)
    ncam, T, J, _ = p.shape
    vis = visibility(p)
    out = np.zeros((T, J, 3))
    for t in range(T):
        for j in range(J):
            cams = np.nonzero(vis[:, t, j])[0]
            if cams.size < 2:
                continue
            rows = []
            for c in cams:
                x, y = p[c, t, j, 1], p[c, t, j, 0]
                rows.append(x * P[c, 2] - P[c, 0])
                rows.append(y * P[c, 2] - P[c, 1])
            A = np.asarray(rows)
            X = np.linalg.svd(A)[2][-1]
            out[t, j] = X[:3] / X[3]
    return out


def triangulate_dlt_batched(points2d_px, P):
    """Same result as triangulate_dlt (zero-row padding leaves the right-singular vectors intact),
    vectorised so the CPU baseline can be timed on 1k+ frames."""
    p = np.asarray(points2d_px, np.float64)
    ncam, T, J, _ = p.shape
    vis = visibility(p)  # (7, T, J)
    x = p[..., 1][..., None]
    y = p[..., 0][..., None]
    r0 = x * P[:, None, None, 2, :] - P[:, None, None, 0, :]
    r1 = y * P[:, None, None, 2, :] - P[:, None, None, 1, :]
    A = np.stack([r0, r1], axis=-2) * vis[..., None, None]  # (7, T, J, 2, 4)
    A = np.moveaxis(A, 0, 2).reshape(T, J, 2 * ncam, 4)
    Vt = np.linalg.svd(A)[2]
    X = Vt[..., -1, :]
    ok = vis.sum(axis=0) >= 2
    with np.errstate(divide="ignore", invalid="ignore"):
        out = X[..., :3] / X[..., 3:4]
    return np.where(ok[..., None], out, 0.0)


# --------------------------------------------------------------------------------------------
# a7: bundle adjustment (extrinsics + points; intrinsics and distortion frozen)
# --------------------------------------------------------------------------------------------
def rotvec_from_matrix(R):
    from scipy.spatial.transform import Rotation

    return Rotation.from_matrix(np.asarray(R, np.float64)).as_rotvec()


def matrix_from_rotvec(rvec):
    """Rodrigues formula (equivalent to cv2.Rodrigues / scipy Rotation.from_rotvec)."""
    rvec = np.asarray(rvec, np.float64)
    single = rvec.ndim == 1
    r = np.atleast_2d(rvec)
    th = np.linalg.norm(r, axis=1)
    out = np.empty((r.shape[0], 3, 3))
    for i in range(r.shape[0]):
        if th[i] < 1e-12:
            k = r[i]
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            out[i] = np.eye(3) + Kx
        else:
            k = r[i] / th[i]
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            out[i] = np.eye(3) + np.sin(th[i]) * Kx + (1 - np.cos(th[i])) * (Kx @ Kx)
    return out[0] if single else out


def build_observations(points2d_px):
    """Observation table in (frame, joint, camera) order; points with >= 2 views only.

    Returns cam_idx (n,), pt_idx (n,), obs_xy (n, 2) as (x = col_px, y = row_px), and the
    (t, j) -> point slot map (T, J) with -1 for untriangulated joints."""
    p = np.asarray(points2d_px, np.float64)
    ncam, T, J, _ = p.shape
    vis = visibility(p)
    nviews = vis.sum(axis=0)
    slot = np.full((T, J), -1, dtype=np.int64)
    ok = nviews >= 2
    slot[ok] = np.arange(int(ok.sum()))
    cam_idx, pt_idx, obs = [], [], []
    for t in range(T):
        for j in range(J):
            if slot[t, j] < 0:
                continue
            for c in range(ncam):
                if vis[c, t, j]:
                    cam_idx.append(c)
                    pt_idx.append(slot[t, j])
                    obs.append((p[c, t, j, 1], p[c, t, j, 0]))
    return np.asarray(cam_idx, np.int64), np.asarray(pt_idx, np.int64), np.asarray(obs, np.float64).reshape(-1, 2), slot


def ba_pack(R, tvec, points3d, slot):
    rv = np.stack([rotvec_from_matrix(R[c]) for c in range(len(R))])
    cams = np.concatenate([rv, np.asarray(tvec, np.float64)], axis=1).ravel()
    pts = np.asarray(points3d, np.float64)[slot >= 0]
    return np.concatenate([cams, pts.ravel()])


def ba_residuals(x, ncam, intr, cam_idx, pt_idx, obs_xy):
    """r = pi(K_c (R(rvec_c) X_p + t_c)) - obs, interleaved (x0, y0, x1, y1, ...)."""
    cams = x[: ncam * 6].reshape(ncam, 6)
    pts = x[ncam * 6 :].reshape(-1, 3)
    Rm = matrix_from_rotvec(cams[:, :3])
    Xc = np.einsum("nij,nj->ni", Rm[cam_idx], pts[pt_idx]) + cams[cam_idx, 3:]
    fx, fy = intr[cam_idx, 0, 0], intr[cam_idx, 1, 1]
    cx, cy = intr[cam_idx, 0, 2], intr[cam_idx, 1, 2]
    u = fx * Xc[:, 0] / Xc[:, 2] + cx
    v = fy * Xc[:, 1] / Xc[:, 2] + cy
    return np.stack([u - obs_xy[:, 0], v - obs_xy[:, 1]], axis=1).ravel()


def ba_sparsity(ncam, npts, cam_idx, pt_idx):
    from scipy.sparse import lil_matrix

    n = cam_idx.size
    A = lil_matrix((2 * n, ncam * 6 + npts * 3), dtype=int)
    i = np.arange(n)
    for s in range(6):
        A[2 * i, cam_idx * 6 + s] = 1
        A[2 * i + 1, cam_idx * 6 + s] = 1
    for s in range(3):
        A[2 * i, ncam * 6 + pt_idx * 3 + s] = 1
        A[2 * i + 1, ncam * 6 + pt_idx * 3 + s] = 1
    return A


def bundle_adjust_scipy(points2d_px, R, tvec, intr, return_info=False):
    """The reference's solver configuration (see module docstring).  Returns adjusted (R, tvec)."""
    from scipy.optimize import least_squares

    R = np.asarray(R, np.float64)
    tvec = np.asarray(tvec, np.float64)
    intr = np.asarray(intr, np.float64)
    ncam = R.shape[0]
    P = projection_matrices(R, tvec, intr)
    pts0 = triangulate_dlt_batched(points2d_px, P)
    cam_idx, pt_idx, obs_xy, slot = build_observations(points2d_px)
    x0 = ba_pack(R, tvec, pts0, slot)
    npts = int((slot >= 0).sum())
    A = ba_sparsity(ncam, npts, cam_idx, pt_idx)
    res = least_squares(
        ba_residuals,
        x0,
        jac_sparsity=A,
        method="trf",
        x_scale="jac",
        ftol=1e-4,
        args=(ncam, intr, cam_idx, pt_idx, obs_xy),
    )
    cams = res.x[: ncam * 6].reshape(ncam, 6)
    R_new = matrix_from_rotvec(cams[:, :3])
    t_new = cams[:, 3:].copy()
    if return_info:
        return R_new, t_new, res
    return R_new, t_new


def reprojection_error(points2d_px, points3d, R, tvec, intr):
    """Mean Euclidean pixel distance over visible observations of triangulated joints."""
    cam_idx, pt_idx, obs_xy, slot = build_observations(points2d_px)
    x = ba_pack(R, tvec, points3d, slot)
    r = ba_residuals(x, len(R), np.asarray(intr, np.float64), cam_idx, pt_idx, obs_xy).reshape(-1, 2)
    return float(np.mean(np.linalg.norm(r, axis=1)))


# --------------------------------------------------------------------------------------------
# a9: procrustes   (reference df3d/procrustes.py:51-263, plot_util.py:85-91)
# --------------------------------------------------------------------------------------------
# joints used for the rigid fit: BODY_COXA / COXA_FEMUR among the first 19 of a side
# (reference skeleton_fly.py tracked_points[0:19] -> indices 0,1,5,6,10,11)
_FIT_JOINTS = [0, 1, 5, 6, 10, 11]


def _bone_lengths(side):
    """side: (T, 19, 3) -> (T, 12): 3 limbs x 4 consecutive-joint distances."""
    limbs = side[:, :15].reshape(side.shape[0], 3, 5, 3)
    return np.linalg.norm(limbs[:, :, 1:] - limbs[:, :, :-1], axis=-1).reshape(side.shape[0], 12)


def _rigid_fit(X, Y):
    """MATLAB-style procrustes without scaling: rotation T and translation c mapping Y onto X."""
    muX, muY = X.mean(0), Y.mean(0)
    X0, Y0 = X - muX, Y - muY
    normX, normY = np.sqrt((X0**2).sum()), np.sqrt((Y0**2).sum())
    X0 = X0 / normX
    Y0 = Y0 / normY
    U, s, Vt = np.linalg.svd(X0.T @ Y0, full_matrices=False)
    Tm = Vt.T @ U.T
    c = muX - muY @ Tm
    return Tm, c


def _procrustes_side(pts, template):
    s = np.median(np.median(_bone_lengths(template), axis=0) / np.median(_bone_lengths(pts), axis=0))
    pts = pts - np.median(pts.reshape(-1, 3), axis=0)
    pts = pts * s
    Tm, c = _rigid_fit(np.median(template[:, _FIT_JOINTS], axis=0), np.median(pts[:, _FIT_JOINTS], axis=0))
    return pts @ Tm + c


def procrustes_separate(points3d, template_points3d):
    pts = np.asarray(points3d, np.float64)
    tmpl = np.asarray(template_points3d, np.float64)
    out = np.zeros_like(pts)
    out[:, :19] = _procrustes_side(pts[:, :19].copy(), tmpl[:, :19])
    out[:, 19:38] = _procrustes_side(pts[:, 19:38].copy(), tmpl[:, 19:38])
    return out
