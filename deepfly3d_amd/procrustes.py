"""a9: per-side Procrustes registration of the triangulated pose to the template, and the `get_points3d` chain.

Replaces `df3d.procrustes.procrustes_seperate` (reference df3d/procrustes.py:51-151, its MATLAB-style rigid fit
:154-263 and the median-centring of plot_util.py:85-91), `normalize_pose_3d` (plot_util.py:85-91) and
`filter_batch` (signal_util.py:69-100).  The transforms are SEQUENCE-GLOBAL (medians over all frames), so in
multi-GPU runs they are applied once on rank 0 after the gather.  All arithmetic runs in
libdf3d_hip.so (csrc/pose3d.hip); the only host work is reducing the constant 15-frame template to the 60 numbers
the kernels need (parameter preprocessing, cached).
"""
import numpy as np
import torch

from . import _native, ops
from .config import BODY_COXA, COXA_FEMUR, TRACKED_SIDE, load_procrustes_template

_FIT = [j for j, k in enumerate(TRACKED_SIDE) if k in (BODY_COXA, COXA_FEMUR)]  # side joints 0,1,5,6,10,11
_template_cache = {}


def template_constants(template=None):
    """(seg_med [2, 12], fit_med [2, 6, 3]) of a template pose [F, 38, 3]: per side the median over its frames of
    the 12 leg-segment lengths and of the 6 fit joints."""
    key = None if template is None else id(template)
    if key in _template_cache:
        return _template_cache[key]
    tmpl = load_procrustes_template() if template is None else np.asarray(template, np.float64)
    seg = np.empty((2, 12))
    fit = np.empty((2, 6, 3))
    for s, lo in enumerate((0, 19)):
        side = tmpl[:, lo : lo + 19]
        legs = side[:, :15].reshape(side.shape[0], 3, 5, 3)
        seg[s] = np.median(np.linalg.norm(np.diff(legs, axis=2), axis=-1).reshape(side.shape[0], -1), axis=0)
        fit[s] = np.median(side[:, _FIT], axis=0)
    if template is None:
        _template_cache[key] = (seg, fit)
    return seg, fit


def _to_device(points3d, device):
    _native.require_gpu()
    if isinstance(points3d, torch.Tensor) and points3d.is_cuda:
        return points3d.to(torch.float64).contiguous()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    return torch.as_tensor(np.ascontiguousarray(points3d, dtype=np.float64)).to(dev)


def procrustes_separate(points3d, template=None, device=None, return_tensor=False):
    """points3d (T, 38, 3) float64 -> registered copy; joints 0-18 and 19-37 are aligned independently."""
    x = _to_device(points3d, device)
    seg, fit = template_constants(template)
    out = ops.procrustes(x, seg, fit)
    return out if return_tensor else out.cpu().numpy()


def video_pose(points3d_wo_procrustes, template=None, device=None):
    """The pose `Core.get_points3d` returns (reference df3d/core.py:332-343): Procrustes -> median-centred and
    axis-swapped -> One-Euro filtered, all on the device."""
    x = _to_device(points3d_wo_procrustes, device)
    seg, fit = template_constants(template)
    p = ops.procrustes(x, seg, fit)
    n = ops.pose_normalize(p, rotate=True)
    return ops.oneeuro_filter(n).cpu().numpy()
