"""-m gpu parity tests for a9 (Procrustes) and the `Core.get_points3d` chain (median-centre / axis swap / One-Euro
filter) through the C ABI, against the reference-executed vectors and the CPU oracle.

Bars: exact order statistics and the One-Euro recurrence are BIT-EXACT; Procrustes (a 3x3 SVD whose summation
order differs from LAPACK's) within 1e-11 mm of the reference's output (the reference's own bar is 1e-5)."""
import numpy as np
import pytest
import torch

from oracle import geometry as og
from oracle import postprocess as pp

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).cuda()


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 255, 256, 257, 1000, 4097])
def test_column_median_is_exact(native_lib, cuda, n):
    from deepfly3d_amd import ops

    rng = np.random.default_rng(n)
    cols = rng.normal(0, 3, size=(7, n))
    cols[1] = np.round(cols[1])  # many ties
    cols[2] = -np.abs(cols[2])  # all negative
    cols[3] = 0.0
    cols[4, ::2] = -0.0
    cols[5] *= 1e-300  # denormal-range magnitudes
    cols[6] = np.sort(cols[6])
    got = ops.column_median(_dev(cols)).cpu().numpy()
    assert np.array_equal(got, np.median(cols, axis=1))


def test_column_median_bad_args(native_lib, cuda):
    from deepfly3d_amd import _native

    with pytest.raises(_native.NativeLibraryError):
        _native.check(native_lib.df3d_column_median(None, 1, 4, 4, None, None), "df3d_column_median")
    x = _dev(np.zeros((1, 4)))
    with pytest.raises(_native.NativeLibraryError):
        _native.check(native_lib.df3d_column_median(x.data_ptr(), 1, 0, 0, x.data_ptr(), None), "df3d_column_median")


@pytest.mark.parametrize("name", ["procrustes_golden", "procrustes_jitter", "pose_chain_jitter"])
def test_procrustes_matches_reference_vectors(native_lib, cuda, golden_dir, name):
    from deepfly3d_amd.procrustes import procrustes_separate

    d = np.load(f"{golden_dir}/{name}.npz")
    inp = d["inp"].copy()
    out = procrustes_separate(inp)
    want = d["out"] if "out" in d else d["procrustes"]
    err = np.abs(out - want).max()
    print(f"{name}: device Procrustes vs reference output: {err:.2e}")
    assert err < 1e-11
    assert np.array_equal(inp, d["inp"])  # unlike the reference's in-place centring, the input is left intact


def test_procrustes_long_sequence_against_oracle(native_lib, cuda, golden_dir):
    """20 001 frames (odd) and 20 000 (even): medians over a long sequence, custom template."""
    from deepfly3d_amd.procrustes import procrustes_separate

    g3 = np.load(f"{golden_dir}/golden_3d.npz")
    tmpl = np.load(f"{golden_dir}/template.npz")["points3d"]
    rng = np.random.default_rng(5)
    base = np.tile(g3["points3d_wo_procrustes"], (1334, 1, 1))[:20001]
    X = base * 1.7 + rng.normal(0, 0.05, size=base.shape) + np.array([0.3, -1.0, 2.0])
    for T in (20001, 20000):
        got = procrustes_separate(X[:T], template=tmpl)
        want = og.procrustes_separate(X[:T], tmpl)
        assert np.abs(got - want).max() < 1e-10


def test_pose_normalize_bit_exact(native_lib, cuda, golden_dir):
    from deepfly3d_amd import ops

    for name in ("pose_chain_golden", "pose_chain_jitter"):
        d = np.load(f"{golden_dir}/{name}.npz")
        got = ops.pose_normalize(_dev(d["procrustes"]), rotate=True).cpu().numpy()
        assert np.array_equal(got, d["normalized"])
        plain = ops.pose_normalize(_dev(d["procrustes"]), rotate=False).cpu().numpy()
        assert np.array_equal(plain, pp.normalize_pose_3d(d["procrustes"], rotate=False))


def test_pose_normalize_long_sequences_take_the_multi_workgroup_median(native_lib, cuda):
    """More than 65 536 values per axis: the medians come from the histogram / select kernel pairs (many workgroups per
    column) -- same exact order statistics: odd and even counts, ties, against numpy; and against the 3-double work buffer of
    round 1's callers, which keeps the one-workgroup kernel."""
    import ctypes

    from deepfly3d_amd import _native, ops

    lib = _native.load()
    rng = np.random.default_rng(17)
    for T in (20001, 20000):
        X = rng.normal(0, 1, size=(T, 38, 3))
        X[::7] = np.round(X[::7], 1)   # plenty of exact ties
        X[:, :, 2] = -np.abs(X[:, :, 2]) * 1e-3   # one axis with all keys in one top-level bin
        got = ops.pose_normalize(_dev(X), rotate=False).cpu().numpy()
        assert np.array_equal(got, X - np.median(X.reshape(-1, 3), axis=0))
        x = _dev(X)
        out, work = torch.empty_like(x), torch.empty(3, dtype=torch.float64, device=x.device)
        _native.check(lib.df3d_pose_normalize(x.data_ptr(), T, 38, 0, out.data_ptr(), work.data_ptr(), 3, None), "df3d_pose_normalize")
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), got)


def test_oneeuro_bit_exact(native_lib, cuda, golden_dir):
    from deepfly3d_amd import ops

    d = np.load(f"{golden_dir}/oneeuro_random.npz")
    got = ops.oneeuro_filter(_dev(d["inp"])).cpu().numpy()
    assert np.array_equal(got, d["out"])
    for name in ("pose_chain_golden", "pose_chain_jitter"):
        c = np.load(f"{golden_dir}/{name}.npz")
        assert np.array_equal(ops.oneeuro_filter(_dev(c["normalized"])).cpu().numpy(), c["filtered"])
    # single frame, and zero frames
    one = ops.oneeuro_filter(_dev(d["inp"][:1])).cpu().numpy()
    assert np.array_equal(one, d["inp"][:1])
    assert ops.oneeuro_filter(_dev(np.zeros((0, 38, 3)))).shape == (0, 38, 3)
    # other parameters / stamps, against the oracle on a long series (2 000 frames)
    rng = np.random.default_rng(11)
    walk = np.cumsum(rng.normal(0, 0.1, size=(2000, 5, 3)), axis=0)
    got = ops.oneeuro_filter(_dev(walk), freq=30.0, mincutoff=0.5, beta=0.7, dcutoff=2.0).cpu().numpy()
    assert np.array_equal(got, pp.oneeuro_filter(walk, freq=30.0, mincutoff=0.5, beta=0.7, dcutoff=2.0))


def test_video_pose_chain(native_lib, cuda, golden_dir):
    """Core.get_points3d's chain end to end against the reference-executed vectors."""
    from deepfly3d_amd.procrustes import video_pose

    for name in ("pose_chain_golden", "pose_chain_jitter"):
        d = np.load(f"{golden_dir}/{name}.npz")
        got = video_pose(d["inp"])
        err = np.abs(got - d["filtered"]).max()
        print(f"{name}: get_points3d chain vs reference: {err:.2e}")
        assert err < 1e-11
