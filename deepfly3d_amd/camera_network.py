"""a5 / a6 / a7 / a8: the `pyba.CameraNetwork`-shaped object `df3d.core.Core` drives
(call sites reference df3d/core.py:120-126, 165, 246-250, 339, 355-360, 396).

Holds per-camera R, tvec, intr, distort and the 2-D detections in PIXELS, (row, col) order
(= normalised * [H, W], reference df3d/core.py:247).  `triangulate()` and `bundle_adjust()` run on the
MI355X through libdf3d_hip.so; nothing here computes on the CPU beyond packing a few 3x4 matrices.
"""
import os

import numpy as np
import torch

from . import _native, ops
from . import bundle_adjust as _ba
from .config import config


class Camera:
    def __init__(self, cam_id, points2d, R=None, tvec=None, intr=None, distort=None, image_path=None):
        self.cam_id = cam_id
        self.points2d = points2d  # (T, J, 2) pixels, (row, col)
        self.R = None if R is None else np.array(R, dtype=np.float64)
        self.tvec = None if tvec is None else np.array(tvec, dtype=np.float64).reshape(3)
        self.intr = None if intr is None else np.array(intr, dtype=np.float64)
        self.distort = np.zeros(5) if distort is None else np.array(distort, dtype=np.float64).reshape(-1)
        self.image_path = image_path

    def __getitem__(self, img_id):
        return self.points2d[img_id]

    def has_calibration(self):
        return self.R is not None and self.tvec is not None and self.intr is not None

    @property
    def P(self):
        return self.intr @ np.concatenate([self.R, self.tvec[:, None]], axis=1)

    def is_empty(self):
        return self.points2d is None or not np.any(self.points2d)

    def get_image(self, img_id):
        """The camera's image as an ndarray (IO plumbing; reference core.py:323)."""
        from PIL import Image

        path = self.image_path.format(cam_id=self.cam_id, img_id=img_id)
        if not os.path.exists(path):
            path = self.image_path.format(cam_id=self.cam_id, img_id=f"{img_id:06d}")
        return np.asarray(Image.open(path))

    def plot_2d(self, img_id, points2d=None, bones=None, colors=None, image=None, radius=6, width=3):
        """The camera's image with the 2-D pose drawn on it, as an RGB ndarray (visualisation only, host side; call site
        reference df3d/core.py:317-319).  points2d: (J, 2) pixels (row, col), default this camera's detections of the
        image; joints at 0 are unseen and skipped.  bones: [[a, b], ...]; colors: one RGB triple per limb."""
        from PIL import Image, ImageDraw

        from .config import LIMB_COLORS, limb_of_joint, skeleton_bones

        pts = np.asarray(self.points2d[img_id] if points2d is None else points2d, dtype=np.float64)
        bones = skeleton_bones() if bones is None else bones
        colors = LIMB_COLORS if colors is None else colors
        img = self.get_image(img_id) if image is None else np.asarray(image)
        canvas = Image.fromarray(img).convert("RGB")
        draw = ImageDraw.Draw(canvas)
        seen = lambda j: j < len(pts) and pts[j, 0] != 0 and pts[j, 1] != 0  # noqa: E731
        colour = lambda j: tuple(int(v) for v in colors[limb_of_joint(j) % len(colors)])  # noqa: E731
        for a, b in bones:
            if seen(a) and seen(b):
                draw.line([(pts[a, 1], pts[a, 0]), (pts[b, 1], pts[b, 0])], fill=colour(a), width=width)
        for j in range(len(pts)):
            if seen(j):
                draw.ellipse([pts[j, 1] - radius, pts[j, 0] - radius, pts[j, 1] + radius, pts[j, 0] + radius], fill=colour(j))
        return np.asarray(canvas)

    def summarize(self):
        # key order as in the reference's golden pickles
        return {"R": self.R, "tvec": self.tvec, "distort": self.distort, "intr": self.intr}


class CameraNetwork:
    def __init__(self, points2d, calib=None, image_path=None, colors=None, bones=None, device=None):
        """points2d: (ncam, T, J, 2) float64 pixels (row, col).  calib: {cam_id: {R, tvec, intr, distort}};
        extra non-camera keys (e.g. a whole df3d_result dict, reference core.py:120-126) are ignored."""
        self._points2d = np.ascontiguousarray(points2d, dtype=np.float64)
        self.image_path = image_path
        self.colors, self.bones = colors, bones
        self.device = device
        ncam = self._points2d.shape[0]
        self.cam_list = []
        lookup = {}
        if calib is not None:
            for k, v in calib.items():
                if isinstance(k, (int, np.integer)) and isinstance(v, dict) and "R" in v:
                    lookup[int(k)] = v
        for c in range(ncam):
            cal = lookup.get(c, {})
            self.cam_list.append(Camera(c, self._points2d[c], cal.get("R"), cal.get("tvec"), cal.get("intr"), cal.get("distort"), image_path))
        self.points3d = None

    def __getitem__(self, cam_id):
        return self.cam_list[cam_id]

    @property
    def points2d(self):
        return self._points2d

    def has_calibration(self):
        return all(c.has_calibration() for c in self.cam_list)

    def _device(self):
        _native.require_gpu()
        return torch.device(self.device if self.device is not None else f"cuda:{torch.cuda.current_device()}")

    def _stack(self):
        R = np.stack([c.R for c in self.cam_list])
        t = np.stack([c.tvec for c in self.cam_list])
        K = np.stack([c.intr for c in self.cam_list])
        return R, t, K

    def triangulate(self):
        """Multi-view DLT of every (frame, joint) seen by >= 2 cameras (HIP kernel, float64)."""
        dev = self._device()
        P = np.stack([c.P for c in self.cam_list])
        px = torch.from_numpy(self._points2d).to(dev)
        self.points3d = ops.triangulate(P, px).cpu().numpy()
        return self.points3d

    def bundle_adjust(self, update_intrinsic=False, update_distort=False):
        """Adjust camera extrinsics (+ internal 3-D points); intrinsics / distortion stay frozen -- the only
        configuration the reference uses (core.py:249)."""
        if update_intrinsic or update_distort:
            raise NotImplementedError("only update_intrinsic=False, update_distort=False is supported (the reference's call)")
        R, t, K = self._stack()
        # pyba bounds the problem (`max_num_images`, 1 000 by its default as far as recalled -- pyba is not in the
        # reference checkout, SURVEY.md App. A.3 "unpinned: frame sub-sampling"); recordings up to that length -- every
        # golden vector, every 1 000-frame window -- use all frames.  Longer ones are cut to an evenly strided,
        # deterministic subset, which also bounds the Jacobian (12 * nobs doubles) instead of growing with T.
        pts = self._ba_subset()
        R_new, t_new, info = _ba.bundle_adjust(pts, R, t, K, device=self._device(), return_info=True)
        for c, cam in enumerate(self.cam_list):
            cam.R, cam.tvec = R_new[c], t_new[c]
        self.ba_info = {k: v for k, v in info.items() if k != "x"}
        self.triangulate()
        return self.ba_info

    def _ba_subset(self, points3d=None):
        """The frames the bundle adjustment (and its reported error) look at: all of them up to config["ba_max_images"],
        an evenly strided subset beyond.  Returns points2d[, points3d] of that subset."""
        T = self._points2d.shape[1]
        cap = int(config.get("ba_max_images") or 0)
        if cap <= 0 or T <= cap:
            return self._points2d if points3d is None else (self._points2d, points3d)
        stride = -(-T // cap)
        if not getattr(self, "_warned_subset", False):
            self._warned_subset = True
            from . import logger

            logger.warning(
                f"bundle adjustment: {T} frames > ba_max_images = {cap}: using every {stride}th frame ({len(range(0, T, stride))} frames). "
                "pyba draws a RANDOM subset here (not in the reference checkout, unpinned): calibrations of recordings longer than "
                f"{cap} frames are not comparable number for number with the reference's.")
        p2 = np.ascontiguousarray(self._points2d[:, ::stride])
        return p2 if points3d is None else (p2, np.ascontiguousarray(points3d[::stride]))

    def reprojection_error(self):
        """Mean pixel distance between observations and re-projected triangulated joints (device residual kernel of
        the bundle adjustment + a device reduction).  pyba's exact definition is not in the reference checkout; this
        one is the mean Euclidean distance over the observations of joints seen by >= 2 cameras, and the reference
        only prints the value (core.py:250)."""
        if self.points3d is None:
            self.triangulate()
        R, t, K = self._stack()
        # on the frames the bundle adjustment itself used (a 100 000-frame recording would otherwise upload tens of millions of
        # observation rows to print one number)
        p2, p3 = self._ba_subset(self.points3d)
        return _ba.reprojection_error(p2, p3, R, t, K, device=self._device())

    def summarize(self):
        out = {c.cam_id: c.summarize() for c in self.cam_list}
        out["points3d"] = self.points3d
        out["points2d"] = self._points2d
        return out
