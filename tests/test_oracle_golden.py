"""CPU tests: the oracle against the reference's golden vectors (tests/golden/*.npz, written by
tests/golden/make_golden.py from the reference's own pickles and by executing the reference's own modules).
These PIN the oracle; the -m gpu tests then compare the HIP path with the pinned oracle."""
import numpy as np
import pytest

from oracle import geometry as og
from oracle import trf_lsmr as ot


def _load(golden_dir, name):
    return np.load(f"{golden_dir}/{name}.npz")


def test_golden_schema_and_grid(golden_dir):
    g2, g3 = _load(golden_dir, "golden_2d"), _load(golden_dir, "golden_3d")
    assert list(g3["key_order"]) == ["0", "1", "2", "3", "4", "5", "6", "points3d", "points2d", "points3d_wo_procrustes", "camera_ordering", "heatmap_confidence"]
    assert list(g3["cam_key_order"]) == ["R", "tvec", "distort", "intr"]
    assert g2["points2d"].shape == (7, 15, 38, 2) and g2["heatmap_confidence"].shape == (7, 15, 19, 1)
    assert g3["camera_ordering"].dtype == np.int64 and np.array_equal(g3["camera_ordering"], np.arange(7))
    # hard arg-max: every golden point sits on the 64 x 128 heat-map grid
    grid = g2["points2d"] * np.array([64.0, 128.0])
    assert np.array_equal(grid, np.round(grid))
    # confidences are float32-exact values stored as float64
    c = g2["heatmap_confidence"]
    assert np.array_equal(c, c.astype(np.float32).astype(np.float64))


@pytest.mark.parametrize("tag", ["id", "rev", "clc"])
def test_relayout_matches_reference_execution(golden_dir, tag):
    r = _load(golden_dir, f"relayout_{tag}")
    out = og.relayout_19_to_38(r["in_points2d"], r["camera_ordering"])
    assert np.array_equal(out, r["out_points2d"])
    # captured call arguments of the reference: flips = ordering[4:]
    assert np.array_equal(r["camera_ids_to_flip"], r["camera_ordering"][4:])


def test_golden_points2d_layout_properties(golden_dir):
    """The 38-joint layout of the golden 2-D result: front camera all zero, antenna/stripes zero for
    ordering[2] / ordering[4], 'unseen' of left cameras encoded as (0, 1)."""
    p = _load(golden_dir, "golden_2d")["points2d"]
    assert np.all(p[3] == 0)
    assert np.all(p[2, :, 15:] == 0) and np.all(p[0, :, 19:] == 0) and np.all(p[1, :, 19:] == 0)
    for cam in (4, 5, 6):
        assert np.all(p[cam, :, :19, 0] == 0) and np.all(p[cam, :, :19, 1] == 1)
    assert np.all(p[4, :, 34:, 0] == 0) and np.all(p[4, :, 34:, 1] == 1)


def test_triangulation_pinned_by_golden(golden_dir):
    g2, g3 = _load(golden_dir, "golden_2d"), _load(golden_dir, "golden_3d")
    px = og.pixels_from_normalised(g2["points2d"], [960, 480])
    P = og.projection_matrices(g3["R"], g3["tvec"], g3["intr"])
    X = og.triangulate_dlt(px, P)
    assert np.abs(X - g3["points3d_wo_procrustes"]).max() < 1e-12
    assert np.abs(og.triangulate_dlt_batched(px, P) - X).max() < 1e-12


def test_procrustes_pinned_by_reference_execution(golden_dir):
    tmpl = _load(golden_dir, "template")["points3d"]
    g3 = _load(golden_dir, "golden_3d")
    assert np.abs(og.procrustes_separate(g3["points3d_wo_procrustes"], tmpl) - g3["points3d"]).max() < 1e-12
    for name in ("procrustes_golden", "procrustes_jitter"):
        d = _load(golden_dir, name)
        assert np.abs(og.procrustes_separate(d["inp"], tmpl) - d["out"]).max() < 1e-12


def test_bundle_adjust_pinned_by_golden(golden_dir):
    """reference tests/test_df3d.py:198-244 (test_calibration): cameras atol 1e-4, 3-D atol 1e-5 -- for the scipy
    configuration AND for the restated TRF+LSMR with analytic Jacobian the device driver mirrors."""
    c, g2, g3 = _load(golden_dir, "calib"), _load(golden_dir, "golden_2d"), _load(golden_dir, "golden_3d")
    tmpl = _load(golden_dir, "template")["points3d"]
    px = og.pixels_from_normalised(g2["points2d"], [960, 480])
    for solver in (og.bundle_adjust_scipy, ot.bundle_adjust):
        R, t, info = solver(px, c["R"], c["tvec"], c["intr"], return_info=True)
        nfev = info.nfev if hasattr(info, "nfev") else info["nfev"]
        assert nfev == 4
        assert np.abs(R - g3["R"]).max() < 1e-4 and np.abs(t - g3["tvec"]).max() < 1e-4
        X = og.triangulate_dlt(px, og.projection_matrices(R, t, c["intr"]))
        assert np.abs(X - g3["points3d_wo_procrustes"]).max() < 1e-5
        assert np.abs(og.procrustes_separate(X, tmpl) - g3["points3d"]).max() < 1e-5
    # golden intrinsics / distortion are the initial ones (frozen), front camera untouched
    assert np.array_equal(g3["intr"], c["intr"]) and np.array_equal(g3["distort"], c["distort"])
    assert np.abs(g3["R"][3] - c["R"][3]).max() < 1e-12


def test_reference_run_to_run_noise_floor(golden_dir):
    a, b = _load(golden_dir, "golden_3d"), _load(golden_dir, "golden_3d_run2")
    assert np.abs(a["tvec"] - b["tvec"]).max() < 1e-5 and np.abs(a["points3d"] - b["points3d"]).max() < 1e-6


def test_analytic_jacobian_matches_finite_differences(golden_dir):
    c, g2 = _load(golden_dir, "calib"), _load(golden_dir, "golden_2d")
    px = og.pixels_from_normalised(g2["points2d"], [960, 480])
    cam_idx, pt_idx, obs, slot = og.build_observations(px)
    X0 = og.triangulate_dlt_batched(px, og.projection_matrices(c["R"], c["tvec"], c["intr"]))
    x0 = og.ba_pack(c["R"], c["tvec"], X0, slot)
    r, Jc, Jp = ot.eval_blocks(x0, 7, c["intr"], cam_idx, pt_idx, obs)
    J = ot.BlockJacobian(7, int((slot >= 0).sum()), cam_idx, pt_idx, Jc, Jp)
    assert np.abs(r - og.ba_residuals(x0, 7, c["intr"], cam_idx, pt_idx, obs)).max() < 1e-10
    for k in (0, 2, 4, 7, 40, 42, 500):
        e = np.zeros(x0.size)
        h = 1e-6 * max(1.0, abs(x0[k]))
        e[k] = h
        fd = (og.ba_residuals(x0 + e, 7, c["intr"], cam_idx, pt_idx, obs) - og.ba_residuals(x0 - e, 7, c["intr"], cam_idx, pt_idx, obs)) / (2 * h)
        assert np.abs(fd - J.matvec(e / h)).max() < 1e-6 * max(1.0, np.abs(fd).max())


def test_argmax_oracle_semantics():
    hm = np.zeros((2, 3, 8, 16), dtype=np.float32)
    hm[0, 0, 3, 5] = 2.0
    hm[0, 1, 3, 5] = 2.0
    hm[0, 1, 1, 2] = 2.0  # tie -> first in row-major order
    pts, conf = og.heatmap_argmax(hm)
    assert pts.dtype == np.float32 and conf.dtype == np.float32
    assert pts[0, 0].tolist() == [3 / 8, 5 / 16] and pts[0, 1].tolist() == [1 / 8, 2 / 16]
    assert pts[1, 2].tolist() == [0.0, 0.0] and conf[0, 0] == 2.0


def test_pose_chain_pinned_by_reference_execution(golden_dir):
    """normalize_pose_3d / filter_batch restatements are bit-identical to the reference's own functions."""
    from oracle import postprocess as pp

    tmpl = np.load(f"{golden_dir}/template.npz")["points3d"]
    for name in ("pose_chain_golden", "pose_chain_jitter"):
        d = np.load(f"{golden_dir}/{name}.npz")
        p, n, f = pp.pose_chain(d["inp"], tmpl)
        assert np.abs(p - d["procrustes"]).max() < 1e-12
        assert np.array_equal(pp.normalize_pose_3d(d["procrustes"]), d["normalized"])
        assert np.array_equal(pp.oneeuro_filter(d["normalized"]), d["filtered"])
        assert np.abs(f - d["filtered"]).max() < 1e-12
    d = np.load(f"{golden_dir}/oneeuro_random.npz")
    assert np.array_equal(pp.oneeuro_filter(d["inp"][:120]), d["out"][:120])


def test_jpeg_oracle_pinned_by_libjpeg(golden_dir):
    """The C restatement of the baseline luma decode equals libjpeg-turbo (through Pillow) bit for bit on the
    reference's own test JPEGs and on encoder-made files (tables, sampling, restart intervals, odd sizes)."""
    import glob
    import io

    from PIL import Image

    from oracle import jpeg as oj

    def pil_luma(blob):
        im = Image.open(io.BytesIO(blob))
        if im.mode != "L":
            im.draft("L", im.size)
        return np.asarray(im)

    for p in sorted(glob.glob(f"{golden_dir}/images/*.jpg")):
        b = open(p, "rb").read()
        y, coef, nsym = oj.decode_luma(b, with_coefficients=True)
        assert np.array_equal(y, pil_luma(b)) and coef.shape == (60, 120, 64) and 100000 < nsym < 200000
    rng = np.random.default_rng(0)
    for h, w in [(8, 8), (17, 33), (100, 75), (1, 1)]:
        for q in (30, 95, 100):
            for kw in ({}, {"optimize": True}, {"subsampling": 0}, {"subsampling": 2}, {"restart_marker_blocks": 5}):
                for c in (None, 3):
                    if c is None and "subsampling" in kw:
                        continue
                    img = (rng.random((h, w) + ((c,) if c else ())) * 255).astype(np.uint8)
                    buf = io.BytesIO()
                    Image.fromarray(img).save(buf, "JPEG", quality=q, **kw)
                    assert np.array_equal(oj.decode_luma(buf.getvalue()), pil_luma(buf.getvalue()))
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", progressive=True)
    assert oj.status(buf.getvalue()) == 3 and oj.status(b"nope") == 2


def test_hourglass_oracle_has_the_surveyed_architecture():
    """No reference vector can exercise the network (nely-df2d and its weights are not in the checkout: parity
    unpinned), but its SHAPE is pinned: the torch restatement has the counts SURVEY.md App. B derives from the reference's
    constants (df3d/config.py:18,33,36): 102 convolutions, 6.73 M parameters, 19 heat-maps at 1/4 resolution from the last of 2 stacks."""
    import torch

    from oracle import hourglass_torch as oh

    net = oh.build(seed=0)
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 102
    assert sum(p.numel() for p in net.parameters()) == 6733222
    with torch.no_grad():
        out = net(torch.zeros(1, 3, 64, 128))
    assert tuple(out.shape) == (1, 19, 16, 32)  # the last stack's heat-maps (what df2d's inference consumes)
    assert net.num_stacks == 2 if hasattr(net, "num_stacks") else True
