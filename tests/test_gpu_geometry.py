"""-m gpu parity tests for the non-network kernels, through the C ABI, against the CPU oracle and the
reference's golden vectors.  Bit-exact for index work (arg-max, re-layout); float64 tolerances stated per test."""
import numpy as np
import pytest
import torch

from oracle import geometry as og

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(f"{golden_dir}/{name}.npz")


# ---------------------------------------------------------------- a3 arg-max ---------------------
@pytest.mark.parametrize("n,joints,h,w", [(1, 19, 64, 128), (13, 19, 64, 128), (3, 5, 8, 16), (2, 19, 128, 256)])
def test_argmax_bit_exact_random(native_lib, cuda, n, joints, h, w):
    from deepfly3d_amd import ops

    g = torch.Generator().manual_seed(n * 1000 + h)
    hm = torch.randn((n, joints, h, w), generator=g, dtype=torch.float32)
    pts, conf = ops.heatmap_argmax(hm.to(cuda))
    rp, rc = og.heatmap_argmax(hm.numpy())
    assert np.array_equal(pts.cpu().numpy(), rp)
    assert np.array_equal(conf.cpu().numpy(), rc)


def test_argmax_ties_and_edges(native_lib, cuda):
    from deepfly3d_amd import ops

    hm = torch.zeros((6, 19, 64, 128), dtype=torch.float32)
    hm[0] = 0.0  # all equal -> index 0
    hm[1, :, 63, 127] = 1.0  # last element
    hm[2, :, 10, 5] = 2.0
    hm[2, :, 40, 100] = 2.0  # tie: first (10, 5) wins
    hm[3] = -1.0
    hm[3, :, 0, 64] = -0.5  # negative values, peak in lane-crossing position
    hm[4] = float("-inf")  # -inf plane -> index 0, conf -inf
    hm[5, :, 31, 77] = 3.0
    hm[5, :, 31, 78] = 3.0  # adjacent tie inside one 16-byte chunk
    pts, conf = ops.heatmap_argmax(hm.to(cuda))
    rp, rc = og.heatmap_argmax(hm.numpy())
    assert np.array_equal(pts.cpu().numpy(), rp)
    assert np.array_equal(conf.cpu().numpy(), rc)
    assert pts[2, 0].tolist() == [10 / 64, 5 / 128]


def test_argmax_empty_and_bad_args(native_lib, cuda):
    from deepfly3d_amd import _native, ops

    pts, conf = ops.heatmap_argmax(torch.empty((0, 19, 64, 128), dtype=torch.float32, device=cuda))
    assert pts.shape == (0, 19, 2) and conf.shape == (0, 19)
    with pytest.raises(_native.NativeLibraryError):
        ops.heatmap_argmax(torch.zeros((1, 2, 3, 5), dtype=torch.float32, device=cuda))
    with pytest.raises(ValueError):
        ops.heatmap_argmax(torch.zeros((1, 2, 4, 4), dtype=torch.float64, device=cuda))


def test_argmax_values_are_grid_points(native_lib, cuda):
    """Reference property: points * (64, 128) are exact integers; conf is the f32 peak."""
    from deepfly3d_amd import ops

    hm = torch.rand((50, 19, 64, 128), generator=torch.Generator().manual_seed(3))
    pts, conf = ops.heatmap_argmax(hm.to(cuda))
    grid = pts.cpu().numpy().astype(np.float64) * np.array([64.0, 128.0])
    assert np.array_equal(grid, np.round(grid))
    assert np.array_equal(conf.cpu().numpy(), hm.amax(dim=(2, 3)).numpy())


# ---------------------------------------------------------------- a4 re-layout -------------------
def test_argmax_counts_planes_with_non_finite_values(native_lib, cuda):
    """df3d_heatmap_argmax_checked: the counter is incremented once per plane that holds an infinity or a NaN ANYWHERE (not only at the peak: a
    NaN never wins the arg-max, so the confidences alone would hide it), clean planes leave it alone, points and confidences are those of the
    unchecked call."""
    from deepfly3d_amd import ops

    g = torch.Generator().manual_seed(4)
    hm = torch.rand((6, 19, 64, 128), generator=g, dtype=torch.float32).to(cuda)
    flag = torch.zeros(1, dtype=torch.int32, device=cuda)
    pts, conf = ops.heatmap_argmax(hm, nonfinite=flag)
    assert int(flag.item()) == 0
    ref_pts, ref_conf = ops.heatmap_argmax(hm)
    assert torch.equal(pts, ref_pts) and torch.equal(conf, ref_conf)
    hm[0, 3, 10, 77] = float("nan")      # off the peak, a lane's second quad
    hm[2, 0, 63, 127] = float("inf")     # the plane's last value
    hm[2, 18, 0, 0] = float("-inf")      # the plane's first value
    hm[5, 7, 31, 64] = float("nan")
    hm[5, 7, 31, 65] = float("inf")      # two in one plane: counted once
    pts, conf = ops.heatmap_argmax(hm, nonfinite=flag)
    assert int(flag.item()) == 4
    assert bool(torch.isfinite(conf[0, 3])) and float(conf[2, 0]) == float("inf")
    ops.heatmap_argmax(hm[1:2], nonfinite=flag)   # a clean slice adds nothing
    assert int(flag.item()) == 4
    with pytest.raises(ValueError):
        ops.heatmap_argmax(hm, nonfinite=torch.zeros(1, dtype=torch.int64, device=cuda))


@pytest.mark.parametrize("tag", ["id", "rev", "clc"])
def test_relayout_matches_reference_vectors(native_lib, cuda, golden_dir, tag):
    from deepfly3d_amd import ops

    r = _load(golden_dir, f"relayout_{tag}")
    out = ops.relayout_19_to_38(torch.from_numpy(r["in_points2d"]).to(cuda), r["camera_ordering"])
    assert out.dtype == torch.float64
    assert np.array_equal(out.cpu().numpy(), r["out_points2d"])  # bit-exact vs the reference's own lines


def test_relayout_rejects_bad_ordering(native_lib, cuda):
    from deepfly3d_amd import _native, ops

    p = torch.zeros((7, 2, 19, 2), dtype=torch.float32, device=cuda)
    with pytest.raises(_native.NativeLibraryError):
        ops.relayout_19_to_38(p, [0, 1, 2, 3, 4, 5, 5])
    with pytest.raises(_native.NativeLibraryError):
        ops.relayout_19_to_38(p, [0, 1, 2, 3, 4, 5, 7])


# ---------------------------------------------------------------- a6 triangulation ---------------
def test_triangulate_golden(native_lib, cuda, golden_dir):
    """Golden 2-D points + golden cameras -> golden points3d_wo_procrustes (reference tolerance 1e-5 mm;
    this kernel: <= 1e-9 mm)."""
    from deepfly3d_amd import ops

    g2, g3 = _load(golden_dir, "golden_2d"), _load(golden_dir, "golden_3d")
    px = og.pixels_from_normalised(g2["points2d"], [960, 480])
    P = og.projection_matrices(g3["R"], g3["tvec"], g3["intr"])
    X = ops.triangulate(P, torch.from_numpy(px).to(cuda)).cpu().numpy()
    assert np.abs(X - g3["points3d_wo_procrustes"]).max() < 1e-9
    assert np.abs(X - og.triangulate_dlt(px, P)).max() < 1e-9
    # untriangulated joints are exactly zero
    vis = og.visibility(px).sum(axis=0) < 2
    assert np.all(X[vis] == 0.0)


def test_triangulate_random_views_and_edge_cases(native_lib, cuda, golden_dir):
    from deepfly3d_amd import ops

    c = _load(golden_dir, "calib")
    P = og.projection_matrices(c["R"], c["tvec"], c["intr"])
    rng = np.random.default_rng(5)
    T, J = 40, 38
    X = rng.normal(0, 1.0, size=(T, J, 3))
    Xh = np.concatenate([X, np.ones((T, J, 1))], axis=-1)
    proj = np.einsum("cij,tkj->ctki", P, Xh)
    uv = proj[..., :2] / proj[..., 2:3]
    px = np.stack([uv[..., 1], uv[..., 0]], axis=-1)  # (row, col)
    px += rng.normal(0, 0.5, size=px.shape)  # pixel noise
    vis = rng.random((7, T, J)) < 0.6
    vis[:, 0, 0] = False  # nobody sees it
    vis[:, 0, 1] = False
    vis[3, 0, 1] = True  # a single view
    vis[:, 0, 2] = True  # all seven views
    px = px * vis[..., None]
    got = ops.triangulate(P, torch.from_numpy(px).to(cuda)).cpu().numpy()
    ref = og.triangulate_dlt(px, P)
    assert np.abs(got - ref).max() < 1e-8
    assert np.all(got[0, 0] == 0) and np.all(got[0, 1] == 0)


def test_triangulate_large_matches_batched_oracle(native_lib, cuda, golden_dir):
    """BASELINE config size (1 000 frames x 38 joints): kernel == batched numpy SVD oracle."""
    from deepfly3d_amd import ops

    g2, g3 = _load(golden_dir, "golden_2d"), _load(golden_dir, "golden_3d")
    px = og.pixels_from_normalised(np.tile(g2["points2d"], (1, 67, 1, 1))[:, :1000], [960, 480])
    P = og.projection_matrices(g3["R"], g3["tvec"], g3["intr"])
    got = ops.triangulate(P, torch.from_numpy(px).to(cuda)).cpu().numpy()
    assert np.abs(got - og.triangulate_dlt_batched(px, P)).max() < 1e-9
    assert got.shape == (1000, 38, 3)
