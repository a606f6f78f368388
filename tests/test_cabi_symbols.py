"""CPU tests: the C-ABI library builds for gfx950, loads, and exports every symbol include/df3d_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "df3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(df3d_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    syms = _declared_symbols()
    for needed in ("df3d_hg_forward", "df3d_heatmap_argmax", "df3d_triangulate", "df3d_ba_eval", "df3d_ba_lsmr", "df3d_relayout_19_to_38", "df3d_preprocess_u8"):
        assert needed in syms


def test_library_exports_every_declared_symbol(native_lib):
    missing = [s for s in _declared_symbols() if not hasattr(native_lib, s)]
    assert not missing, f"libdf3d_hip.so lacks {missing}"


def test_python_prototypes_cover_the_header(native_lib):
    from deepfly3d_amd import _native

    assert sorted(_native.PROTOTYPES) == _declared_symbols()


def test_library_is_gfx950_code_object():
    from deepfly3d_amd import _native

    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_argument_validation_without_gpu(native_lib):
    """Entry points validate arguments before touching the device, so this runs on CPU."""
    from deepfly3d_amd import _native

    assert native_lib.df3d_version() == _native.ABI_VERSION == 610   # DF3D_ABI_VERSION of include/df3d_hip.h
    rc = native_lib.df3d_heatmap_argmax(None, 1, 19, 3, 5, None, None, None)  # 15 pixels: not a multiple of 4
    assert rc == -1 and b"multiple of 4" in native_lib.df3d_last_error()
    order = (ctypes.c_int * 7)(0, 1, 2, 3, 4, 5, 5)
    rc = native_lib.df3d_relayout_19_to_38(ctypes.c_void_p(16), order, 2, ctypes.c_void_p(16), None)
    assert rc == -1 and b"permutation" in native_lib.df3d_last_error()
    h = ctypes.c_void_p()
    assert native_lib.df3d_hg_create(7, 2, ctypes.byref(h)) == -1
    assert native_lib.df3d_hg_create(_native.DF3D_DTYPE_F32, 2, ctypes.byref(h)) == 0
    # forward before weights are set is a state error, never a silent success
    rc = native_lib.df3d_hg_forward(h, ctypes.c_void_p(256), 1, ctypes.c_void_p(256), ctypes.c_void_p(256), 1 << 40, None)
    assert rc == -3 and b"weights" in native_lib.df3d_last_error()
    assert native_lib.df3d_hg_set_option(h, b"fuse_upadd", 3) == -1 and native_lib.df3d_hg_set_option(h, b"no_such_knob", 1) == -1
    native_lib.df3d_hg_destroy(h)
    # front-end and sequence tail
    p16 = ctypes.c_void_p(4096)
    assert native_lib.df3d_jpeg_work_bytes(4, 960, 480, 300000) > 4 * 921600 and native_lib.df3d_jpeg_work_bytes(-1, 960, 480, 0) == 0
    rc = native_lib.df3d_jpeg_decode_luma(ctypes.c_void_p(4100), p16, p16, 1, 1000, 1000, 960, 480, p16, p16, None, p16, 1 << 30, 0, None)
    assert rc == -1 and b"16-byte aligned" in native_lib.df3d_last_error()
    rc = native_lib.df3d_jpeg_decode_luma(p16, p16, p16, 1, 1000, 1000, 960, 480, p16, p16, None, p16, 16, 0, None)
    assert rc == -1 and b"work buffer too small" in native_lib.df3d_last_error()
    assert native_lib.df3d_jpeg_decode_luma(None, None, None, 0, 0, 0, 960, 480, None, None, None, None, 0, 0, None) == 0  # empty batch
    assert native_lib.df3d_column_median(p16, 1, 0, 0, p16, None) == -1
    assert native_lib.df3d_procrustes_work_doubles(1000) >= 60 * 1000
    two = (ctypes.c_double * 24)()
    six = (ctypes.c_double * 36)()
    assert native_lib.df3d_procrustes(p16, 0, two, six, p16, p16, 1 << 20, None) == -1
    assert native_lib.df3d_procrustes(p16, 10, two, six, p16, p16, 8, None) == -1 and b"work buffer" in native_lib.df3d_last_error()
    assert native_lib.df3d_pose_normalize(p16, 10, 38, 1, p16, p16, 2, None) == -1
    assert native_lib.df3d_oneeuro_filter(p16, 10, 114, 0.0, 0.1, 2.0, 1.0, 1, 0.1, p16, None) == -1
    assert native_lib.df3d_oneeuro_filter(None, 0, 114, 100.0, 0.1, 2.0, 1.0, 1, 0.1, None, None) == 0  # no frames
    # the trust-region driver's entries (ABI 610): a null problem / null vectors / a non-positive radius are argument errors, before any launch
    out = (ctypes.c_double * 19)()
    prob = _native.BAProblem()
    args = [p16] * 5
    assert native_lib.df3d_ba_trf_subspace(None, *args, 1.0, *([p16] * 9), out, None, 0) == -1
    assert native_lib.df3d_ba_trf_subspace(ctypes.byref(prob), *args, 1.0, *([p16] * 9), out, None, 0) == -1   # an empty problem
    assert native_lib.df3d_ba_trf_trial(None, 0.5, 0.5, *([p16] * 13), out, None) == -1
    assert native_lib.df3d_ba_trf_linearize(None, p16, p16, 1, *([p16] * 6), 1, p16, None) == -1


def test_native_file_reader(native_lib, tmp_path):
    """df3d_read_files (host helper of the JPEG front-end): layout, padding, size query, error reporting."""
    import numpy as np

    from deepfly3d_amd import _native

    rng = np.random.default_rng(3)
    blobs = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in (1, 15, 16, 17, 70001, 0, 4096)]
    paths = []
    for i, b in enumerate(blobs):
        paths.append(str(tmp_path / f"f{i}.bin"))
        open(paths[-1], "wb").write(b)
    n = len(paths)
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    starts, sizes, total = np.empty(n, np.uint32), np.empty(n, np.uint32), ctypes.c_size_t()
    assert native_lib.df3d_read_files(arr, n, None, 0, starts.ctypes.data, sizes.ctypes.data, ctypes.byref(total), 3) == _native.DF3D_ENOSPC
    assert total.value == sum((len(b) + 15) // 16 * 16 for b in blobs)
    buf = np.full(total.value + 16, 0xAB, np.uint8)
    assert native_lib.df3d_read_files(arr, n, buf.ctypes.data, buf.size - 1, starts.ctypes.data, sizes.ctypes.data, ctypes.byref(total), 3) == _native.DF3D_ENOSPC
    assert native_lib.df3d_read_files(arr, n, buf.ctypes.data, buf.size, starts.ctypes.data, sizes.ctypes.data, ctypes.byref(total), 3) == 0
    off = 0
    for b, s, z in zip(blobs, starts, sizes):
        assert s == off and s % 16 == 0 and z == len(b) and bytes(buf[s : s + z]) == b
        pad = (len(b) + 15) // 16 * 16
        assert not buf[s + z : s + pad].any()  # padding zeroed
        off += pad
    assert not buf[total.value :].any()
    arr2 = (ctypes.c_char_p * 2)(os.fsencode(paths[0]), os.fsencode(str(tmp_path / "missing.jpg")))
    assert native_lib.df3d_read_files(arr2, 2, buf.ctypes.data, buf.size, starts.ctypes.data, sizes.ctypes.data, ctypes.byref(total), 8) == _native.DF3D_EIO
    assert b"missing.jpg" in native_lib.df3d_last_error()
    arr3 = (ctypes.c_char_p * 1)(os.fsencode(str(tmp_path)))  # a directory is not a frame
    assert native_lib.df3d_read_files(arr3, 1, buf.ctypes.data, buf.size, starts.ctypes.data, sizes.ctypes.data, ctypes.byref(total), 2) == _native.DF3D_EIO
    assert native_lib.df3d_read_files(None, 0, None, 0, None, None, ctypes.byref(total), 1) == 0 and total.value == 0


def test_engine_plan_accounting(native_lib):
    """FLOPs / bytes of the engine's own plan equal the SURVEY.md 8d figures (35.993 GFLOP, 647.3 MB per view fp32)."""
    from deepfly3d_amd import _native

    for dtype, mb in ((_native.DF3D_DTYPE_F32, 647.33), (_native.DF3D_DTYPE_BF16, 323.67)):
        h = ctypes.c_void_p()
        assert native_lib.df3d_hg_create(dtype, 2, ctypes.byref(h)) == 0
        fl, by = ctypes.c_double(), ctypes.c_double()
        assert native_lib.df3d_hg_work(h, 1, ctypes.byref(fl), ctypes.byref(by)) == 0
        assert abs(fl.value / 1e9 - 35.9934) < 1e-3
        assert abs(by.value / 1e6 - mb) < 0.05
        # 25 fused bottlenecks, 2 fused heads, 8 max-pools folded into bottleneck epilogues, 8 upsample + add passes
        # folded into the consuming bottlenecks (the M1 byte model above still counts them), the ninth max-pool (input of the
        # second stack) written by the ring bottleneck that reads that tensor
        assert native_lib.df3d_hg_num_steps(h) == 119 - 2 * 23 - 6 - 4 - 8 - 8 - 1
        assert native_lib.df3d_hg_set_option(h, b"fuse", 0) == 0 and native_lib.df3d_hg_num_steps(h) == 119
        native_lib.df3d_hg_destroy(h)


def test_f32s_engine_shares_the_f32_plan_and_needs_its_weight_copy(native_lib):
    """DF3D_DTYPE_F32S (split half-precision products on float32 tensors) is the F32 engine as far as a caller's buffers go: the same plan, the
    same workspace and work accounting; its lowp buffer = a float32-sized pre-split copy of the blob + the F32 engine's streams; set_weights
    refuses to run without it (argument checks come before any device work, so this runs on CPU)."""
    from deepfly3d_amd import _native

    assert _native.DF3D_DTYPE_F32S == 3
    hs = {}
    for dt in (_native.DF3D_DTYPE_F32, _native.DF3D_DTYPE_F32S):
        h = ctypes.c_void_p()
        assert native_lib.df3d_hg_create(dt, 2, ctypes.byref(h)) == 0
        hs[dt] = h
    f32, f32s = hs[_native.DF3D_DTYPE_F32], hs[_native.DF3D_DTYPE_F32S]
    assert native_lib.df3d_hg_num_steps(f32) == native_lib.df3d_hg_num_steps(f32s)
    assert native_lib.df3d_hg_blob_floats(f32) == native_lib.df3d_hg_blob_floats(f32s)
    for n in (1, 7, 896):
        assert native_lib.df3d_hg_workspace_bytes(f32, n) == native_lib.df3d_hg_workspace_bytes(f32s, n)
    copy = (native_lib.df3d_hg_blob_floats(f32) * 4 + 255) & ~255
    # the exact-fp32 engine's default plan adds, per identity-skip bottleneck (23; layer2, with its skip convolution: 256 KiB of 1x1 weights; layer1: 256 + 64 KiB), the Winograd-domain weights of its 3x3 (1 MiB) and W3 with
    # permuted rows (128 KiB): option "wino" (round 6); without it the two engines' streams are the same
    with_wino = native_lib.df3d_hg_lowp_bytes(f32)
    assert native_lib.df3d_hg_set_option(f32, b"wino", 0) == 0
    assert with_wino - native_lib.df3d_hg_lowp_bytes(f32) == 23 * ((1 << 20) + (128 << 10) + (128 << 10)) + (1 << 20) + (256 << 10) + (256 << 10) + (64 << 10)   # per identity block: U, W3 (permuted rows), W1 for the LDS-resident conv1
    assert native_lib.df3d_hg_lowp_bytes(f32s) == copy + native_lib.df3d_hg_lowp_bytes(f32)
    rc = native_lib.df3d_hg_set_weights(f32s, ctypes.c_void_p(256), None, None)
    assert rc == -1 and b"f32s" in native_lib.df3d_last_error()
    for h in hs.values():
        native_lib.df3d_hg_destroy(h)


def test_no_gpu_fails_loudly(native_lib):
    import torch

    from deepfly3d_amd import _native

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeLibraryError):
        _native.require_gpu()
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.synthetic import synthetic_state_dict

    with pytest.raises(_native.NativeLibraryError):
        HourglassEngine(synthetic_state_dict(0))
