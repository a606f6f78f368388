"""The per-frame hot path on device-resident frames: 7 views -> heat-maps -> arg-max/confidence -> 38-joint
layout -> multi-view DLT (fixed calibration), with no host round trip between the stages.

This is the loop `df3d-cli` runs per recording (reference df3d/cli.py:292-300 -> core.py:170-203, 349-369),
restructured for a GPU: frames are independent, so a rank owns a contiguous frame range and processes it in
batches of `frames_per_batch` frames (x 7 views).  Multi-GPU sharding + the single gather live in
deepfly3d_amd/distributed.py; Procrustes (sequence-global) runs after the gather.
"""
import ctypes

import numpy as np
import torch

from . import _native, ops
from .config import config


class FramePipeline:
    def __init__(self, engine, R, tvec, intr, camera_ordering=(0, 1, 2, 3, 4, 5, 6), image_shape=(960, 480)):
        """engine: HourglassEngine; R/tvec/intr: (7,3,3)/(7,3)/(7,3,3) calibration already in camera-id order
        (i.e. after the reference's `calib_reordered`, core.py:240-242); image_shape = [W, H] of the camera frames."""
        self.engine = engine
        self.lib = _native.load()
        self.device = engine.device
        self.order = [int(c) for c in camera_ordering]
        self.P = np.ascontiguousarray(np.einsum("cij,cjk->cik", np.asarray(intr, np.float64),
                                                np.concatenate([np.asarray(R, np.float64), np.asarray(tvec, np.float64)[..., None]], axis=-1)))
        self.W, self.H = float(image_shape[0]), float(image_shape[1])
        self._order_c = (ctypes.c_int * 7)(*self.order)

    def run_batch(self, frames, out_points2d, out_conf, out_points3d, t0):
        """frames: [F, 7, 256, 512, 3] float32 cuda (frame-major).  Writes rows t0..t0+F of
        out_points2d [7, T, 38, 2] f64, out_conf [7, T, 19] f32 and out_points3d [T, 38, 3] f64 (cuda)."""
        F = frames.shape[0]
        views = frames.reshape(F * 7, *frames.shape[2:])
        hm = self.engine.forward(views)
        pts, conf = ops.heatmap_argmax(hm, nonfinite=self.engine.nonfinite_planes)  # [F*7, 19, 2], [F*7, 19]; counts planes with inf / NaN
        # (frame, camera) -> (camera, frame): a strided copy of F*7*19*3 numbers (data movement only)
        pts_ct = pts.reshape(F, 7, 19, 2).transpose(0, 1).contiguous()
        out_conf[:, t0 : t0 + F] = conf.reshape(F, 7, 19).transpose(0, 1)
        p38 = ops.relayout_19_to_38(pts_ct, self.order)  # [7, F, 38, 2] f64
        out_points2d[:, t0 : t0 + F] = p38
        X = torch.empty((F, 38, 3), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _native.check(
                self.lib.df3d_triangulate_scaled(self.P.ctypes.data_as(ctypes.c_void_p), p38.data_ptr(), self.H, self.W, 7, F, 38, X.data_ptr(), stream),
                "df3d_triangulate_scaled",
            )
        out_points3d[t0 : t0 + F] = X

    def canary(self, exact_engine, frames, what="the first frame"):
        """A reduced-precision engine against the exact one (same weights) on the 7 views of `frames[0]` (HourglassEngine.canary): call once per
        run, before its results count."""
        views = frames[:1].reshape(7, *frames.shape[2:])
        return self.engine.canary(exact_engine, lambda e: e.forward(views), what=what)

    def check_finite(self, what="this frame range"):
        """Once per run (one 4-byte read-back): refuse results of a reduced-precision engine that overflowed (HourglassEngine.check_finite)."""
        self.engine.check_finite(what)

    def allocate_outputs(self, T):
        dev = self.device
        return (
            torch.zeros((7, T, config["num_joints"], 2), dtype=torch.float64, device=dev),
            torch.zeros((7, T, config["num_predict"]), dtype=torch.float32, device=dev),
            torch.zeros((T, config["num_joints"], 3), dtype=torch.float64, device=dev),
        )

    def run(self, frames, frames_per_batch=8):
        """frames [T, 7, 256, 512, 3] -> (points2d [7,T,38,2] f64, conf [7,T,19] f32, points3d [T,38,3] f64), cuda."""
        T = frames.shape[0]
        outs = self.allocate_outputs(T)
        for t0 in range(0, T, frames_per_batch):
            self.run_batch(frames[t0 : t0 + frames_per_batch], *outs, t0)
        self.check_finite()
        return outs
