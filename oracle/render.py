"""TEST INFRASTRUCTURE: numpy restatement of the video-frame kernels (csrc/render.hip), f4.

The drawing RULE is this build's own (the reference draws with matplotlib / cv2 on the host, df3d/video.py:21-108, core.py:298-319; its
pixels are not a parity target -- nothing reads them back); what is pinned here is that the device kernels implement exactly the rule
they document: float64 distance tests, later table entries win, joints over bones.  Only tests/ import this module.
"""
import math

import numpy as np


def _seg_dist2(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    len2 = dx * dx + dy * dy
    t = ((px - ax) * dx + (py - ay) * dy) / len2 if len2 > 0.0 else np.zeros_like(px)
    t = np.clip(t, 0.0, 1.0)
    qx, qy = ax + t * dx, ay + t * dy
    return (px - qx) * (px - qx) + (py - qy) * (py - qy)


def pose2d_grid(luma, points_px, bones, joint_rgb, radius, line_width):
    """luma [6, H, W] uint8, points_px [6, J, 2] (row, col) -> [2 H, 3 W, 3] uint8."""
    _, H, W = luma.shape
    out = np.zeros((2 * H, 3 * W, 3), np.uint8)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    hw2, rad2 = (0.5 * line_width) ** 2, float(radius) ** 2
    for slot in range(6):
        img = np.repeat(luma[slot][:, :, None], 3, axis=2).copy()
        pts = np.asarray(points_px[slot], np.float64)
        seen = (pts[:, 0] != 0.0) & (pts[:, 1] != 0.0)
        for a, b in bones:
            if seen[a] and seen[b]:
                hit = _seg_dist2(xx, yy, pts[a, 1], pts[a, 0], pts[b, 1], pts[b, 0]) <= hw2
                img[hit] = joint_rgb[a]
        for j in range(len(pts)):
            if seen[j]:
                dx, dy = xx - pts[j, 1], yy - pts[j, 0]
                img[dx * dx + dy * dy <= rad2] = joint_rgb[j]
        r, c = divmod(slot, 3)
        out[r * H:(r + 1) * H, c * W:(c + 1) * W] = img
    return out


def pose3d_panels(points3d, bones, joint_rgb, azimuth_deg3, elevation_deg, lim, size, line_width):
    """points3d [J, 3] -> [size, 3 size, 3] uint8 (orthographic views, black background)."""
    out = np.zeros((size, 3 * size, 3), np.uint8)
    k = 3.14159265358979323846 / 180.0
    ce, se = math.cos(elevation_deg * k), math.sin(elevation_deg * k)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    hw2 = (0.5 * line_width) ** 2
    P = np.asarray(points3d, np.float64)
    for panel in range(3):
        ca, sa = math.cos(azimuth_deg3[panel] * k), math.sin(azimuth_deg3[panel] * k)
        u = -sa * P[:, 0] + ca * P[:, 1]
        v = -se * ca * P[:, 0] - se * sa * P[:, 1] + ce * P[:, 2]
        sx = (u / lim * 0.5 + 0.5) * float(size - 1)
        sy = (0.5 - v / lim * 0.5) * float(size - 1)
        img = np.zeros((size, size, 3), np.uint8)
        for a, b in bones:
            img[_seg_dist2(xx, yy, sx[a], sy[a], sx[b], sy[b]) <= hw2] = joint_rgb[a]
        out[:, panel * size:(panel + 1) * size] = img
    return out


def resize_rgb(img, out_h, out_w):
    """bilinear, pixel centres at half integers, clamped; round half up."""
    ih, iw = img.shape[:2]
    fy = (np.arange(out_h, dtype=np.float64) + 0.5) * float(ih) / float(out_h) - 0.5
    fx = (np.arange(out_w, dtype=np.float64) + 0.5) * float(iw) / float(out_w) - 0.5
    cy, cx = np.clip(fy, 0.0, float(ih - 1)), np.clip(fx, 0.0, float(iw - 1))
    y0, x0 = np.floor(cy).astype(int), np.floor(cx).astype(int)
    y1, x1 = np.minimum(y0 + 1, ih - 1), np.minimum(x0 + 1, iw - 1)
    wy, wx = (cy - y0)[:, None, None], (cx - x0)[None, :, None]
    f = img.astype(np.float64)
    v00, v01, v10, v11 = f[y0][:, x0], f[y0][:, x1], f[y1][:, x0], f[y1][:, x1]
    top = v00 + wx * (v01 - v00)
    bot = v10 + wx * (v11 - v10)
    return np.floor(top + wy * (bot - top) + 0.5).astype(np.uint8)
