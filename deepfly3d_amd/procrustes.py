"""a9: per-side Procrustes registration of the triangulated pose to the template.

Replaces `df3d.procrustes.procrustes_seperate` (reference df3d/procrustes.py:51-151, its MATLAB-style rigid
fit :154-263 and the median-centring of plot_util.py:85-91).  The transform is SEQUENCE-GLOBAL (medians over
all frames), so in multi-GPU runs it is applied once on rank 0 after the gather.  It touches
T x 38 x 3 float64 numbers and three 3x3 SVDs; SURVEY.md 8(a9) scopes it as host float64 work.
"""
import numpy as np

from .config import BODY_COXA, COXA_FEMUR, TRACKED_SIDE, load_procrustes_template

_FIT = [j for j, k in enumerate(TRACKED_SIDE) if k in (BODY_COXA, COXA_FEMUR)]  # joints 0,1,5,6,10,11


def _limb_segment_lengths(side):
    """(T, 19, 3) -> (T, 12): the 4 segment lengths of each of the 3 legs."""
    legs = side[:, :15].reshape(side.shape[0], 3, 5, 3)
    return np.linalg.norm(np.diff(legs, axis=2), axis=-1).reshape(side.shape[0], -1)


def _rigid_transform(target, source):
    """Rotation T and offset c with source @ T + c ~= target (no scaling, reflection allowed = 'best')."""
    mu_t, mu_s = target.mean(axis=0), source.mean(axis=0)
    t0, s0 = target - mu_t, source - mu_s
    t0 = t0 / np.sqrt((t0**2).sum())
    s0 = s0 / np.sqrt((s0**2).sum())
    U, _, Vt = np.linalg.svd(t0.T @ s0, full_matrices=False)
    rot = Vt.T @ U.T
    return rot, mu_t - mu_s @ rot


def _register_side(pts, template):
    ratio = np.median(_limb_segment_lengths(template), axis=0) / np.median(_limb_segment_lengths(pts), axis=0)
    pts = (pts - np.median(pts.reshape(-1, 3), axis=0)) * np.median(ratio)
    rot, off = _rigid_transform(np.median(template[:, _FIT], axis=0), np.median(pts[:, _FIT], axis=0))
    return pts @ rot + off


def procrustes_separate(points3d, template=None):
    """points3d (T, 38, 3) float64 -> registered copy; joints 0-18 and 19-37 are aligned independently."""
    pts = np.asarray(points3d, dtype=np.float64)
    tmpl = load_procrustes_template() if template is None else np.asarray(template, np.float64)
    out = np.zeros_like(pts)
    for lo in (0, 19):
        out[:, lo : lo + 19] = _register_side(pts[:, lo : lo + 19].copy(), tmpl[:, lo : lo + 19])
    return out
