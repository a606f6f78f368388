"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the `Core.get_points3d` chain behind the video pose (reference df3d/core.py:332-343):

  normalize_pose_3d   reference df3d/plot_util.py:85-91 (+ rotate_points3d :10-18): subtract the median of all
                      T*J points per axis, then (x, y, z) -> (x, -z, -y)
  oneeuro_filter      reference df3d/signal_util.py:5-67 (LowPassFilter / OneEuroFilter) driven as in filter_batch
                      (:69-100): per joint and axis one filter, freq 100 Hz initially, mincutoff 0.1, beta 2.0,
                      dcutoff 1.0, time stamps (i+1)*0.1 s -- the filter re-estimates its sampling frequency
                      from consecutive stamps, so from the second sample on freq = 1/(t_i - t_{i-1}) ~ 10 Hz
  pose_chain          procrustes -> normalize -> filter

Pinned by tests/golden/pose_chain_*.npz and oneeuro_random.npz, produced by executing the reference's own
modules (tests/golden/make_golden_post.py).  All arithmetic is float64 in the reference's operation order, so
the restatement is expected to be bit-identical to it.
"""
import math

import numpy as np

ONEEURO = {"freq": 100.0, "mincutoff": 0.1, "beta": 2.0, "dcutoff": 1.0}


def normalize_pose_3d(points3d, rotate=True):
    p = np.array(points3d, dtype=np.float64)
    p -= np.median(p.reshape(-1, 3), axis=0)
    if rotate:
        p = np.stack([p[..., 0], -p[..., 2], -p[..., 1]], axis=-1)
    return p


def _alpha(freq, cutoff):
    te = 1.0 / freq
    tau = 1.0 / (2 * math.pi * cutoff)
    return 1.0 / (1.0 + tau / te)


def oneeuro_filter(pts, freq=ONEEURO["freq"], mincutoff=ONEEURO["mincutoff"], beta=ONEEURO["beta"], dcutoff=ONEEURO["dcutoff"]):
    """pts (T, J, 3) -> filtered copy; channel (j, a) is an independent scalar recurrence over T."""
    x = np.asarray(pts, dtype=np.float64)
    T = x.shape[0]
    flat = x.reshape(T, -1)
    out = np.empty_like(flat)
    for ch in range(flat.shape[1]):
        f = float(freq)
        last_t = None
        prev_raw = None  # last raw sample          (LowPassFilter.lastValue of the value filter)
        s_x = None  # smoothed value
        s_dx = None  # smoothed derivative
        for i in range(T):
            t = (i + 1) * 0.1
            v = float(flat[i, ch])
            if last_t and t:
                f = 1.0 / (t - last_t)
            last_t = t
            dx = 0.0 if prev_raw is None else (v - prev_raw) * f
            a_d = _alpha(f, dcutoff)
            s_dx = dx if s_dx is None else a_d * dx + (1.0 - a_d) * s_dx
            cutoff = mincutoff + beta * math.fabs(s_dx)
            a = _alpha(f, cutoff)
            s_x = v if s_x is None else a * v + (1.0 - a) * s_x
            prev_raw = v
            out[i, ch] = s_x
    return out.reshape(x.shape)


def pose_chain(points3d_wo, template_points3d):
    from .geometry import procrustes_separate

    p = procrustes_separate(points3d_wo, template_points3d)
    n = normalize_pose_3d(p, rotate=True)
    return p, n, oneeuro_filter(n)
