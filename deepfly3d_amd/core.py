"""`Core`: the reference's orchestration surface (df3d.core.Core, reference df3d/core.py:62-544) on the
MI355X back-end.  Same constructor, methods, attributes, exceptions and result-pickle schema for the hot path:

    core = Core(input_folder, output_folder=None, num_images_max=None, camera_ordering=[0..6])
    core.pose2d_estimation(batch_size=8, disable_pin_memory=False)     # reference :170-203
    core.save()                                                        # reference :349-369
    core.calibrate_calc(min_img_id, max_img_id)                        # reference :229-250
    core.save()

GUI-only methods (interactive correction, error navigation, plotting) are out of scope;
the correction store and `corrected_points2d*` are kept so that stored corrections reach the triangulation.
"""
import glob
import os
import pickle
import re
from typing import List, Optional

import numpy as np
import torch

from . import _native, logger, ops
from .camera_network import CameraNetwork
from .config import config, load_calibration
from .db import PoseDB
from .inference import inference_folder
from .os_util import camera_videos, extract_frames, get_max_img_id, parse_frame_rate, parse_vid_name, probe_frame_rate
from .procrustes import procrustes_separate, video_pose

_KNOWN_ORDERINGS = [
    (r"/CLC/", [0, 6, 5, 4, 3, 2, 1]),
    (r"/FA/", [6, 5, 4, 3, 2, 1, 0]),
    (r"/SG/", [6, 5, 4, 3, 2, 1, 0]),
    (r"Laura", [0, 6, 5, 4, 3, 2, 1]),
    (r"AYMANNS_Florian", [6, 5, 4, 3, 2, 1, 0]),
    (r"sample/test", [0, 1, 2, 3, 4, 5, 6]),
    (r"/JB/", [6, 5, 4, 3, 2, 1, 0]),
]


def find_default_camera_ordering(input_folder):
    """Lab-specific defaults keyed on the folder path (reference df3d/core.py:24-59)."""
    folder = str(input_folder)
    for pattern, order in _KNOWN_ORDERINGS:
        if re.search(pattern, folder):
            logger.debug(f"Default camera ordering found: {order}")
            return np.array(order)
    raise NotImplementedError(
        f"Cannot find camera ordering for folder {folder}. Please set your camera ordering using the --order flag. "
        "Example usage is df3d-cli /your/path/images/ --order 0 1 2 3 4 5 6"
    )


def relayout_points2d(points19, camera_ordering, device=None):
    """(7, T, 19, 2) network output -> (7, T, 38, 2) float64 skeleton layout, on the device
    (df3d_relayout_19_to_38; reference df3d/core.py:187-203)."""
    _native.require_gpu()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    t = torch.as_tensor(np.ascontiguousarray(points19, dtype=np.float32)).to(dev)
    return ops.relayout_19_to_38(t, camera_ordering).cpu().numpy()


class Core:
    """Main interface to the 2-D and 3-D pose estimation (same surface as reference df3d/core.py:62)."""

    def __init__(self, input_folder: str, output_folder: Optional[str] = None, num_images_max: Optional[int] = None,
                 camera_ordering: List[int] = [0, 1, 2, 3, 4, 5, 6], dtype: str = "f32", device=None):
        from . import distributed as dd

        self.dtype, self.device = dtype, device
        rank, world = dd.current()
        self.is_primary = rank == 0  # multi-GPU: rank 0 alone expands videos, calibrates and writes results

        def everyone_waits():
            if world > 1:
                torch.distributed.barrier()

        self.input_folder = input_folder
        self.output_folder = output_folder if output_folder is not None else self.input_folder + "_df3d"
        if self.is_primary:
            self.expand_videos()
        everyone_waits()  # the frames exist for every rank from here on
        self.fps = self.get_fps()

        # frame range: ids 0 .. max_img_id, optionally cut to the first num_images_max
        self.num_images_max = num_images_max or 0
        last = get_max_img_id(self.input_folder)
        self.num_images = min(self.num_images_max, last + 1) if self.num_images_max > 0 else last + 1
        self.max_img_id = self.num_images - 1

        self._image_path = os.path.join(self.input_folder, "camera_{cam_id}_img_{img_id}.jpg")
        self.image_shape = self._resolve_image_shape(self._image_path.format(cam_id=0, img_id=0))

        if self.is_primary:
            self.db = PoseDB(self.output_folder)  # creates pose_corr_*.pkl on first use, like the reference
        everyone_waits()
        if not self.is_primary:
            self.db = PoseDB(self.output_folder)  # written by rank 0 above: loaded, not re-created
        self.camera_ordering = self.setup_camera_ordering(camera_ordering)
        self.camNet = self.points2d = self.points3d = self.conf = None
        self._points2d_shard = None  # multi-GPU: this rank's frames of points2d (device tensor, normalised), kept for the sharded DLT
        if os.path.exists(self.save_path):
            self._resume(self.save_path)

    @staticmethod
    def _resolve_image_shape(first_image):
        """[W, H] of the recording: read from the first frame, cross-checked with a configured shape, remembered in
        the config (reference df3d/core.py:84-100: same precedence, same ValueErrors)."""
        configured = config.get("image_shape")
        if os.path.exists(first_image):
            from PIL import Image

            with Image.open(first_image) as im:
                actual = list(im.size)
            if configured is not None and actual != configured:
                raise ValueError(f"Actual image shape {actual} does not match config.py image shape {configured}")
            config["image_shape"] = actual
            return actual
        if configured is None:
            raise ValueError(f"Image shape not specified in config and could not be read from {first_image}")
        return configured

    def _resume(self, result_file):
        """Re-open an earlier result: its poses and the cameras it holds (reference df3d/core.py:109-126)."""
        with open(result_file, "rb") as f:
            earlier = pickle.load(f)
        self.points2d, self.conf = earlier["points2d"], earlier["heatmap_confidence"]
        self.points3d = earlier.get("points3d", self.points3d)
        pixels = earlier["points2d"] * self.image_shape[::-1]
        self.camNet = CameraNetwork(pixels, calib=earlier, image_path=self._image_path, device=self.device)

    # -- properties -----------------------------------------------------------------------------------
    @staticmethod
    def _as_directory(path, create=False):
        if create:
            os.makedirs(path, exist_ok=True)
        path = os.path.abspath(path).rstrip("/")
        assert os.path.isdir(path), f"Not a directory {path}"
        return path

    @property
    def input_folder(self):
        return self._input_folder

    @input_folder.setter
    def input_folder(self, value):
        self._input_folder = self._as_directory(value)

    @property
    def output_folder(self):
        return self._output_folder

    @output_folder.setter
    def output_folder(self, value):
        self._output_folder = self._as_directory(value, create=True)

    @property
    def number_of_joints(self):
        return config["num_joints"]

    @property
    def has_pose(self):
        return True

    @property
    def has_calibration(self):
        return self.camNet.has_calibration()

    @property
    def save_path(self):
        flat = self.input_folder.replace("/", "_")
        return os.path.join(self.output_folder, f"df3d_result_{flat}.pkl")

    # -- hot path -------------------------------------------------------------------------------------
    def pose2d_estimation(self, batch_size: int = 8, disable_pin_memory: bool = False):
        """2-D pose on every frame of every camera, then the 19 -> 38 joint layout (reference :170-203).

        Under `torch.distributed` (one process per GPU) every rank processes a contiguous range of frames and ONE
        gather brings the results to rank 0, which alone goes on to calibrate and save."""
        from . import distributed as dd

        flip = [cam for idx, cam in enumerate(self.camera_ordering) if idx > 3]
        rank, world = dd.current()
        t0, t1 = dd.shard_range(self.num_images, world, rank)
        points19, conf = inference_folder(
            folder=self.input_folder, camera_ids_to_flip=flip, return_heatmap=False, return_confidence=True,
            max_img_id=self.max_img_id, batch_size=batch_size, disable_pin_memory=disable_pin_memory, dtype=self.dtype, device=self.device,
            frame_range=(t0, t1), as_device_tensors=True,
        )
        # 19 -> 38 layout on the device, then (N > 1) ONE gather of the device tensors: no host round trip before it
        points2d = ops.relayout_19_to_38(points19.contiguous(), self.camera_ordering)
        self._points2d_shard = points2d if world > 1 else None
        if dd.collective_needed(world):
            gathered = dd.gather_packed([(points2d, 1), (conf, 1)], self.num_images)
            self.is_primary = rank == 0
            if gathered is None:
                self.points2d = self.conf = None
                return
            points2d, conf = gathered
        self.points2d, self.conf = points2d.cpu().numpy(), conf.cpu().numpy()

    def calibrate_calc(self, min_img_id, max_img_id):
        """Bundle adjustment from the shipped initial calibration (reference :229-250; like the reference the
        image-id range is accepted and unused).  Multi-GPU: rank 0 solves, EVERY rank calls this and leaves it together --
        a failure on rank 0 (the solver, the disk) is raised on all ranks (`distributed.agree`)."""
        from . import distributed as dd

        error = None
        if self.is_primary:
            try:
                calib = load_calibration()
                reordered = {int(cidx): calib[idx] for idx, cidx in enumerate(self.camera_ordering)}
                self.camNet = CameraNetwork(self.points2d * self.image_shape[::-1], calib=reordered, image_path=self._image_path, device=self.device)
                self.camNet.bundle_adjust(update_intrinsic=False, update_distort=False)
                print(f"Reprojection error is {self.camNet.reprojection_error()}")
            except Exception as e:  # noqa: BLE001  (re-raised by agree, on every rank)
                error = e
        dd.agree(error, "calibrate_calc")

    def get_points3d(self):
        """Pose for the 3-D video: array[image_id][joint_id] = (x, y, z) after Procrustes, median-centring + axis swap
        and the One-Euro temporal filter (reference :332-343), computed on the device."""
        return video_pose(np.copy(self.camNet.points3d), device=self.device)

    def save_corrections(self):
        """Write the manual corrections to the output folder (reference :345-347)."""
        self.db.dump()

    def _triangulate_sharded(self):
        """Multi-GPU form of the triangulation inside save(): the cameras are fixed by now (loaded from an earlier result, or
        adjusted by calibrate_calc on rank 0), so rank 0 broadcasts them -- one record of a flag + the seven 3 x 4 projection
        matrices -- every rank triangulates ITS frame range on its own GPU, and one gather brings points3d to rank 0
        (Procrustes is sequence-global and stays there).  Every rank calls this; returns [T, 38, 3] on rank 0, None on the
        others or when rank 0 holds no calibration.  Bit-identical to CameraNetwork.triangulate() on the gathered points."""
        import torch.distributed as dist

        from . import distributed as dd

        rank, world = dd.current()
        dev = dd.local_device() if self.device is None else torch.device(self.device)
        rec = torch.zeros(1 + 7 * 12, dtype=torch.float64)
        if rank == 0 and self.camNet is not None and self.camNet.has_calibration():
            rec[0] = 1.0
            rec[1:] = torch.from_numpy(np.stack([c.P for c in self.camNet.cam_list]).reshape(-1))
        wire = dd._wire_tensor(rec.to(dev), None)
        dist.broadcast(wire, src=0)
        rec = wire.cpu()
        if rec[0].item() == 0.0:
            return None
        t0, t1 = dd.shard_range(self.num_images, world, rank)
        scale = torch.tensor([float(v) for v in self.image_shape[::-1]], dtype=torch.float64, device=dev)
        # This rank's frames, in pixels.  A rank that holds a camera network (rank 0 after calibrate_calc; every rank after
        # reopening a result) reads them from IT, as CameraNetwork.triangulate() does single-GPU -- corrections written in place by
        # corrected_points2d_matrix() are then part of the triangulation; the other ranks use the shard their own inference left.
        px, bad = None, None
        if self.camNet is not None and self.camNet.points2d is not None and self.camNet.points2d.shape[1] == self.num_images:
            px = torch.from_numpy(np.ascontiguousarray(self.camNet.points2d[:, t0:t1])).to(dev)
        elif self._points2d_shard is not None:
            px = self._points2d_shard.to(dev) * scale
        elif self.points2d is not None:
            px = torch.from_numpy(np.ascontiguousarray(self.points2d[:, t0:t1])).to(dev) * scale
        if px is None or px.shape[1] != t1 - t0:
            # e.g. a reopened result whose length is not this run's num_images: NOT silently zeros (round-3 advisor finding) --
            # the collective below still runs well-formed, then every rank raises
            bad = ValueError(f"rank {rank}: {0 if px is None else px.shape[1]} frames of 2-D points for the frame range [{t0}, {t1}) of {self.num_images}")
            px = torch.zeros((7, t1 - t0, config["num_joints"], 2), dtype=torch.float64, device=dev)
        px = px.contiguous()
        # Manual corrections live in rank 0's camera network only (written in place by corrected_points2d_matrix()); the other
        # ranks triangulate from their raw inference shard.  Rank 0 therefore broadcasts the rows its correction store names, with
        # the values ITS table holds for them -- [cam, frame, J x 2 pixels] per row, two small broadcasts -- and every rank writes
        # the rows of its own frame range over its shard: the sharded result is then CameraNetwork.triangulate()'s, corrections
        # included, whichever rank owns the frame (round-4 advisor finding).
        J = config["num_joints"]
        rows = np.zeros((0, 2 + 2 * J))
        if rank == 0 and self.camNet is not None and self.camNet.points2d is not None and self.camNet.points2d.shape[1] == self.num_images:
            try:   # (a malformed correction entry must not leave the peers alone in the broadcasts below: it travels to agree())
                listed = [(int(c), int(i)) for c, per_image in self.db.manual_corrections().items() for i in per_image if c < config["num_cameras"] and i < self.num_images]
                if listed:
                    cams, frames = np.array([c for c, _ in listed]), np.array([i for _, i in listed])
                    rows = np.concatenate([cams[:, None], frames[:, None], self.camNet.points2d[cams, frames].reshape(len(listed), 2 * J)], axis=1).astype(np.float64)
            except Exception as e:  # noqa: BLE001
                bad, rows = bad or e, np.zeros((0, 2 + 2 * J))
        count = dd._wire_tensor(torch.tensor([rows.shape[0]], dtype=torch.int64).to(dev), None)
        dist.broadcast(count, src=0)
        if int(count.cpu()[0]):
            wire = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float64)) if rank == 0 else torch.zeros((int(count.cpu()[0]), 2 + 2 * J), dtype=torch.float64)
            wire = dd._wire_tensor(wire.to(dev), None)
            dist.broadcast(wire, src=0)
            table = wire.cpu()
            mine = (table[:, 1] >= t0) & (table[:, 1] < t1)
            if bool(mine.any()):   # one indexed assignment (rows of one (camera, frame) are unique: the store is a dict of dicts)
                table = table[mine]
                px[table[:, 0].long().to(px.device), (table[:, 1].long() - t0).to(px.device)] = table[:, 2:].reshape(-1, J, 2).to(px.device)
        X = ops.triangulate(rec[1:].reshape(7, 3, 4).numpy(), px) if t1 > t0 else torch.zeros((0, config["num_joints"], 3), dtype=torch.float64, device=dev)
        logger.debug(f"rank {rank} of {world}: triangulated frames [{t0}, {t1}) on {dev}")
        full = dd.gather_frames(X, 0, self.num_images)
        dd.agree(bad, "the sharded triangulation")
        return None if full is None else full.cpu().numpy()

    def save(self):
        """Write df3d_result_*.pkl with the reference's schema and key order (reference :349-369)."""
        from . import distributed as dd

        pts3d_sharded = self._triangulate_sharded() if dd.current()[1] > 1 else None   # a collective: every rank takes part
        error = None
        if self.is_primary:
            try:
                self._write_result(pts3d_sharded)
            except Exception as e:  # noqa: BLE001  (ENOSPC, a failing Procrustes, ...: re-raised by agree, on every rank)
                error = e
        dd.agree(error, "save")   # rank 0 failing here must not leave its peers in the NEXT step's collectives alone

    def _write_result(self, pts3d_sharded=None):
        result = {"points2d": np.copy(self.points2d)}
        if self.camNet is not None and self.camNet.has_calibration():
            if pts3d_sharded is not None:
                self.camNet.points3d = pts3d_sharded
            else:
                self.camNet.triangulate()
            pts3d = self.camNet.points3d
            result["points3d_wo_procrustes"] = pts3d
            result["points3d"] = procrustes_separate(pts3d, device=self.device)
            result = {**self.camNet.summarize(), **result}
        else:
            logger.debug("Triangulation skipped.")
        result["camera_ordering"] = self.camera_ordering
        result["heatmap_confidence"] = self.conf
        with open(self.save_path, "wb") as f:
            pickle.dump(result, f)
        print(f"Saved results at: {self.save_path}")

    # -- helpers --------------------------------------------------------------------------------------
    def _correction_for(self, corrections, cam_id, img_id):
        return corrections.get(cam_id, {}).get(img_id)

    def corrected_points2d(self, cam_id, img_id):
        """Joints of one image in pixels: the manual correction when one is stored, else the estimate
        (reference df3d/core.py:374-385)."""
        estimate = self.camNet.cam_list[cam_id][img_id].copy()
        fix = self._correction_for(self.db.manual_corrections(), cam_id, img_id)
        if fix is not None:
            estimate[:] = fix
        return estimate

    def corrected_points2d_matrix(self):
        """results[cam_id][img_id][joint_id] = (row, col), stored corrections written over the camera network's
        estimates -- in place, like the reference (df3d/core.py:387-401)."""
        corrections = self.db.manual_corrections()
        everything = self.camNet.points2d
        for cam_id, per_image in corrections.items():
            for img_id, fix in per_image.items():
                if cam_id < config["num_cameras"] and img_id < self.num_images:
                    everything[cam_id, img_id, :] = fix
        return everything

    def setup_camera_ordering(self, camera_ordering) -> np.ndarray:
        order = find_default_camera_ordering(self.input_folder) if camera_ordering is None else camera_ordering
        return np.array(order)

    def plot_2d(self, cam_id, img_id, with_corrections=False, smooth=False, joints=[]):
        """Image `img_id` of camera `cam_id` with its 2-D pose drawn on it, as an ndarray (reference df3d/core.py:298-319).
        Host-side drawing; `smooth` (temporal smoothing for videos) is accepted and ignored, `joints` restricts the drawing
        to the listed joint ids."""
        from .config import skeleton_bones

        pts = self.corrected_points2d(cam_id, img_id) if with_corrections else np.array(self.camNet.cam_list[cam_id][img_id], dtype=np.float64)
        if len(joints):
            keep = np.zeros(len(pts), dtype=bool)
            keep[list(joints)] = True
            pts = np.where(keep[:, None], pts, 0.0)
        return self.camNet[cam_id].plot_2d(img_id, points2d=pts, bones=skeleton_bones())

    def get_image(self, cam_id, img_id):
        return self.camNet.cam_list[cam_id].get_image(img_id)

    def get_fps(self):
        """Frame rate of the camera videos (ffprobe on each of them; a warning when they differ, the first one wins),
        None without videos, without ffprobe or when an answer cannot be parsed (reference df3d/core.py:403-428)."""
        rates = []
        for video in camera_videos(self.input_folder):
            cmd, answer = probe_frame_rate(video)
            if answer is None:
                logger.warning(f"Command failed: {' '.join(cmd)}")
                return None
            rate = parse_frame_rate(answer)
            if rate is None:
                logger.warning(f'Could not parse framerate "{answer}" returned by ffprobe, so setting fps to None.')
                return None
            rates.append(rate)
        if len(set(rates)) > 1:
            logger.warning(f"The camera videos have different frame rates {rates}; using {rates[0]}.")
        return rates[0] if rates else None

    def expand_videos(self):
        """camera_x.mp4 -> camera_x_img_y.jpg through ffmpeg for cameras whose frames are not there yet."""
        for video in camera_videos(self.input_folder):
            cam_id = parse_vid_name(os.path.basename(video))
            first_frames = (os.path.join(self.input_folder, f"camera_{cam_id}_img_{n}.jpg") for n in ("0", "000000"))
            if not any(os.path.exists(f) for f in first_frames):
                extract_frames(video, self.input_folder, cam_id)

    def delete_images(self):
        """Remove the expanded frames of every camera that still has its .mp4.  Multi-GPU: every rank must have finished
        reading its shard first (the gather is not a barrier for the senders), and only rank 0 deletes."""
        from . import distributed as dd

        if dd.current()[1] > 1:
            torch.distributed.barrier()
        if not self.is_primary:
            return
        for video in camera_videos(self.input_folder, any_id=False):
            cam_id = parse_vid_name(os.path.basename(video))
            for frame in glob.glob(os.path.join(self.input_folder, f"camera_{cam_id}_img_*.jpg")):
                try:
                    os.remove(frame)
                except FileNotFoundError:
                    pass
