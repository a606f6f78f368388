"""Thin host wrappers over the C ABI for the non-network kernels (a3 arg-max, a4 re-layout, a6 triangulation).

Inputs/outputs are torch CUDA tensors (device memory + stream plumbing only); every computation is a
libdf3d_hip.so kernel.  No CPU fallback: a missing library or GPU raises `_native.NativeLibraryError`.
"""
import ctypes

import numpy as np
import torch

from . import _native


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _on_tensor_device(fn):
    """Kernels are launched on the calling thread's CURRENT HIP device: make that the device of the first tensor
    argument, so a process that drives several GPUs (or one that never called torch.cuda.set_device) stays correct."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kw)
        return fn(*args, **kw)

    return wrapper


def _need(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous {dtype} CUDA tensor")


@_on_tensor_device
def heatmap_argmax(heatmaps, nonfinite=None):
    """heatmaps [n, J, H, W] float32 (cuda) -> (points [n, J, 2] float32 (row/H, col/W), conf [n, J] float32).
    `nonfinite`: an int32 CUDA tensor of one element (zeroed by its owner) that is incremented once per plane holding an infinity or a NaN --
    the overflow guard of the reduced-precision engines (HourglassEngine.check_finite reads it)."""
    lib = _native.load()
    _need(heatmaps, torch.float32, "heatmaps")
    n, j, h, w = heatmaps.shape
    pts = torch.empty((n, j, 2), dtype=torch.float32, device=heatmaps.device)
    conf = torch.empty((n, j), dtype=torch.float32, device=heatmaps.device)
    if nonfinite is not None and not (nonfinite.is_cuda and nonfinite.dtype == torch.int32 and nonfinite.numel() == 1 and nonfinite.device == heatmaps.device):
        raise ValueError("nonfinite must be a one-element int32 CUDA tensor on the heat-maps' device")
    _native.check(
        lib.df3d_heatmap_argmax_checked(heatmaps.data_ptr(), n, j, h, w, pts.data_ptr(), conf.data_ptr(),
                                        nonfinite.data_ptr() if nonfinite is not None else None, _stream(heatmaps)),
        "df3d_heatmap_argmax_checked",
    )
    return pts, conf


@_on_tensor_device
def relayout_19_to_38(points19, camera_ordering):
    """points19 [7, T, 19, 2] float32 (cuda) -> [7, T, 38, 2] float64 (reference df3d/core.py:187-203)."""
    lib = _native.load()
    _need(points19, torch.float32, "points19")
    if points19.shape[0] != 7 or points19.shape[2] != 19 or points19.shape[3] != 2:
        raise ValueError("points19 must be [7, T, 19, 2]")
    T = points19.shape[1]
    order = (ctypes.c_int * 7)(*[int(c) for c in camera_ordering])
    out = torch.empty((7, T, 38, 2), dtype=torch.float64, device=points19.device)
    _native.check(lib.df3d_relayout_19_to_38(points19.data_ptr(), order, T, out.data_ptr(), _stream(points19)), "df3d_relayout_19_to_38")
    return out


@_on_tensor_device
def triangulate(P, points2d_px):
    """P [ncam, 3, 4] float64 (numpy or tensor), points2d_px [ncam, T, J, 2] float64 cuda (row_px, col_px)
    -> X [T, J, 3] float64 cuda; zeros where fewer than two cameras see the joint."""
    lib = _native.load()
    _need(points2d_px, torch.float64, "points2d_px")
    ncam, T, J, two = points2d_px.shape
    if two != 2:
        raise ValueError("points2d_px must be [ncam, T, J, 2]")
    Ph = np.ascontiguousarray(P.detach().cpu().numpy() if isinstance(P, torch.Tensor) else P, dtype=np.float64)
    if Ph.shape != (ncam, 3, 4):
        raise ValueError("P must be [ncam, 3, 4]")
    X = torch.empty((T, J, 3), dtype=torch.float64, device=points2d_px.device)
    _native.check(
        lib.df3d_triangulate(Ph.ctypes.data_as(ctypes.c_void_p), points2d_px.data_ptr(), ncam, T, J, X.data_ptr(), _stream(points2d_px)),
        "df3d_triangulate",
    )
    return X


@_on_tensor_device
def column_median(cols):
    """cols [ncols, n] float64 cuda -> [ncols] exact medians (numpy.median semantics)."""
    lib = _native.load()
    _need(cols, torch.float64, "cols")
    ncols, n = cols.shape
    out = torch.empty((ncols,), dtype=torch.float64, device=cols.device)
    _native.check(lib.df3d_column_median(cols.data_ptr(), ncols, n, n, out.data_ptr(), _stream(cols)), "df3d_column_median")
    return out


@_on_tensor_device
def procrustes(points3d, tmpl_seg_med, tmpl_fit_med):
    """points3d [T, 38, 3] float64 cuda -> registered copy (a9).  tmpl_* are the template's host constants
    (deepfly3d_amd.procrustes.template_constants)."""
    lib = _native.load()
    _need(points3d, torch.float64, "points3d")
    if points3d.dim() != 3 or tuple(points3d.shape[1:]) != (38, 3):
        raise ValueError("points3d must be [T, 38, 3]")
    T = points3d.shape[0]
    seg = np.ascontiguousarray(tmpl_seg_med, dtype=np.float64)
    fit = np.ascontiguousarray(tmpl_fit_med, dtype=np.float64)
    if seg.shape != (2, 12) or fit.shape != (2, 6, 3):
        raise ValueError("template constants must be [2, 12] and [2, 6, 3]")
    need = lib.df3d_procrustes_work_doubles(T)
    work = torch.empty((need,), dtype=torch.float64, device=points3d.device)
    out = torch.empty_like(points3d)
    dp = ctypes.POINTER(ctypes.c_double)
    _native.check(
        lib.df3d_procrustes(points3d.data_ptr(), T, seg.ctypes.data_as(dp), fit.ctypes.data_as(dp), out.data_ptr(), work.data_ptr(), need,
                            _stream(points3d)),
        "df3d_procrustes",
    )
    return out


@_on_tensor_device
def pose_normalize(points3d, rotate=True):
    """[T, J, 3] float64 cuda -> minus the per-axis median of all points, optionally (x, y, z) -> (x, -z, -y)."""
    lib = _native.load()
    _need(points3d, torch.float64, "points3d")
    T, J, three = points3d.shape
    if three != 3:
        raise ValueError("points3d must be [T, J, 3]")
    work = torch.empty((1024,), dtype=torch.float64, device=points3d.device)   # medians + scratch of the long-column median
    out = torch.empty_like(points3d)
    _native.check(lib.df3d_pose_normalize(points3d.data_ptr(), T, J, 1 if rotate else 0, out.data_ptr(), work.data_ptr(), work.numel(), _stream(points3d)),
                  "df3d_pose_normalize")
    return out


@_on_tensor_device
def oneeuro_filter(series, freq=100.0, mincutoff=0.1, beta=2.0, dcutoff=1.0, first_stamp=1, stamp_step=0.1):
    """series [T, ...] float64 cuda: every trailing element is one channel filtered along T (reference
    df3d/signal_util.py:69-100 defaults)."""
    lib = _native.load()
    _need(series, torch.float64, "series")
    T = series.shape[0]
    nch = int(series[0].numel()) if T else 1
    out = torch.empty_like(series)
    _native.check(
        lib.df3d_oneeuro_filter(series.data_ptr(), T, nch, freq, mincutoff, beta, dcutoff, first_stamp, stamp_step, out.data_ptr(), _stream(series)),
        "df3d_oneeuro_filter",
    )
    return out
