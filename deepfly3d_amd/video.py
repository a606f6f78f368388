"""f4: the pose videos of `df3d-cli --video-2d / --video-3d` (reference df3d/video.py:21-108, called from df3d/cli.py:308-321).

The reference builds every frame on the host: six `Core.plot_2d` images (matplotlib / cv2 drawing, one image at a time) stacked into a
2 x 3 grid -- cameras 0, 1, 2 over 4, 5, 6 -- and, for the 3-D video, a third row of three matplotlib 3-D plots; cv2.VideoWriter
encodes them.  Here a frame is drawn by ONE kernel launch on the GPU (csrc/render.hip: `df3d_render_pose2d_grid`,
`df3d_render_pose3d_panels`, `df3d_resize_rgb`) from camera frames decoded on the GPU (csrc/jpeg.hip); the host only moves finished RGB
frames to the encoder:

    ffmpeg on PATH   -> raw RGB frames piped into `ffmpeg ... -c:v mpeg4`, the reference's file name  video_pose2d_<folder>.mp4
    no ffmpeg        -> Motion-JPEG in an AVI container written here (Pillow encodes the frames): same stem, .avi -- this image has
                        neither cv2 nor ffmpeg, and an MP4 muxer is outside the hot path

Same layout, same file naming, same frame rate rule (--output-fps, else the camera videos' rate, else 30) as the reference; the
drawing rule (joint discs over bone segments over the grey image; orthographic 3-D panels) is this build's own and is pinned by
oracle/render.py.  Visualisation only: no result file depends on these pixels.
"""
import ctypes
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import torch

from . import _native, logger
from .config import LIMB_COLORS, config, limb_of_joint, skeleton_bones

DEFAULT_FPS = 30                     # reference video.py:19
GRID_CAMERAS = (0, 1, 2, 4, 5, 6)    # reference video.py:35-36: the front camera (3) is not shown
PANEL_CAMERAS = (4, 5, 6)            # reference video.py:66
PANEL_SIZE = 200                     # img3d_aspect (2, 2) x img3d_dpi 100 (reference video.py:14-15)
SMALL_2D = (100, 200)                # img2d_aspect (2, 1) x 100: (height, width) of one camera image in the 3-D video
JOINT_RADIUS, BONE_WIDTH = 6.0, 3.0  # as Camera.plot_2d draws them (camera_network.py)
PANEL_LINE_WIDTH, PANEL_LIM, PANEL_ELEV = 1.5, 2.0, 30.0   # reference video.py:150-155 (thickness 1.5, lim 2), matplotlib's default elevation


def _tables(num_joints):
    bones = np.asarray([b for b in skeleton_bones() if max(b) < num_joints], dtype=np.int32).reshape(-1, 2)
    rgb = np.asarray([LIMB_COLORS[limb_of_joint(j) % len(LIMB_COLORS)] for j in range(num_joints)], dtype=np.uint8)
    return np.ascontiguousarray(bones), np.ascontiguousarray(rgb)


def panel_azimuths(cameras=PANEL_CAMERAS):
    """matplotlib view_init azimuth per camera (reference plot_util.py:48-51)."""
    return [(-60.0 + 30.0 * c) if c < 3 else (-60.0 + 45.0 * c) for c in cameras]


def merge_stripes(points3d):
    """The stripe joints of the two body sides are the same physical points: both get their mean (reference plot_util.py:62-72)."""
    p = np.array(points3d, dtype=np.float64, copy=True)
    half = p.shape[-2] // 2
    for j in (16, 17, 18):
        if j + half < p.shape[-2]:
            p[..., j, :] = 0.5 * (p[..., j, :] + p[..., j + half, :])
            p[..., j + half, :] = p[..., j, :]
    return p


class FrameRenderer:
    """Device-side drawing of video frames through the C ABI."""

    def __init__(self, height, width, num_joints, device):
        _native.require_gpu()
        self.lib = _native.load()
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.h, self.w, self.j = height, width, num_joints
        self.bones, self.rgb = _tables(num_joints)
        self._bones_p = self.bones.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        self._rgb_p = self.rgb.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def grid2d(self, luma6, points6, out=None):
        """luma6 [6, H, W] uint8 cuda, points6 [6, J, 2] float64 cuda (row_px, col_px) -> [2 H, 3 W, 3] uint8 cuda."""
        if out is None:
            out = torch.empty((2 * self.h, 3 * self.w, 3), dtype=torch.uint8, device=self.dev)
        with torch.cuda.device(self.dev):
            _native.check(self.lib.df3d_render_pose2d_grid(luma6.data_ptr(), self.h, self.w, points6.data_ptr(), self.j, self._bones_p, len(self.bones),
                                                           self._rgb_p, JOINT_RADIUS, BONE_WIDTH, out.data_ptr(), self._stream()), "df3d_render_pose2d_grid")
        return out

    def panels3d(self, points3d, out=None, size=PANEL_SIZE):
        """points3d [J, 3] float64 cuda -> [size, 3 size, 3] uint8 cuda."""
        if out is None:
            out = torch.empty((size, 3 * size, 3), dtype=torch.uint8, device=self.dev)
        az = (ctypes.c_double * 3)(*panel_azimuths())
        with torch.cuda.device(self.dev):
            _native.check(self.lib.df3d_render_pose3d_panels(points3d.data_ptr(), self.j, self._bones_p, len(self.bones), self._rgb_p, az, PANEL_ELEV,
                                                             PANEL_LIM, size, PANEL_LINE_WIDTH, out.data_ptr(), self._stream()), "df3d_render_pose3d_panels")
        return out

    def resize(self, img, out):
        """img [h, w, 3] uint8 cuda -> out (a [oh, ow, 3] view of a wider image is fine: row pitch taken from its stride)."""
        assert img.stride(1) == 3 and out.stride(1) == 3 and img.stride(2) == 1 and out.stride(2) == 1
        with torch.cuda.device(self.dev):
            _native.check(self.lib.df3d_resize_rgb(img.data_ptr(), img.shape[0], img.shape[1], img.stride(0) // 3, out.data_ptr(), out.shape[0], out.shape[1],
                                                   out.stride(0) // 3, self._stream()), "df3d_resize_rgb")
        return out


# ---- encoders ------------------------------------------------------------------------------------------------------------------------
class FfmpegWriter:
    """Raw RGB frames into an ffmpeg child process (mpeg4 in .mp4, like the reference's cv2 'mp4v')."""

    def __init__(self, path, width, height, fps):
        import tempfile

        self.path = path
        self.log = tempfile.TemporaryFile()   # ffmpeg's stderr: a file, so a chatty child can never block on a full pipe
        self.proc = subprocess.Popen(
            ["ffmpeg", "-y", "-loglevel", "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{width}x{height}", "-r", str(fps), "-i", "-",
             "-c:v", "mpeg4", "-q:v", "3", "-pix_fmt", "yuv420p", path], stdin=subprocess.PIPE, stderr=self.log)

    def _reason(self):
        self.log.seek(0)
        return self.log.read()[-2000:].decode(errors="replace").strip()

    def write(self, frame):
        try:
            self.proc.stdin.write(frame.tobytes())
        except BrokenPipeError:
            self.proc.wait()
            raise RuntimeError(f"ffmpeg exited (code {self.proc.returncode}) while writing {self.path}: {self._reason()}") from None

    def close(self):
        """Always reaps the child; raises for a failed encode only when no other exception is already in flight (a `finally:
        writer.close()` must not mask the error that got us here)."""
        in_flight = sys.exc_info()[1] is not None
        try:
            self.proc.stdin.close()
        except (BrokenPipeError, OSError):
            pass
        code = self.proc.wait()
        reason = self._reason()
        self.log.close()
        if code != 0 and not in_flight:
            raise RuntimeError(f"ffmpeg failed (code {code}) writing {self.path}: {reason}")


class MjpegAviWriter:
    """Motion-JPEG in a RIFF/AVI container: what can be written with this image's tools (Pillow) when ffmpeg is absent."""

    def __init__(self, path, width, height, fps, quality=90):
        self.path, self.w, self.h, self.fps, self.q = path, width, height, float(fps), quality
        self.f = open(path, "wb")
        self.index = []
        self.limit = (1 << 32) - (1 << 24)   # a plain RIFF/AVI carries 32-bit sizes and idx1 offsets: stop 16 MiB short of 4 GiB
        self.f.write(b"\0" * 224)   # RIFF + hdrl, patched in close()
        self.movi_at = self.f.tell()
        self.f.write(b"LIST\0\0\0\0movi")

    def write(self, frame):
        import io

        from PIL import Image

        buf = io.BytesIO()
        Image.fromarray(frame).save(buf, format="JPEG", quality=self.q)
        data = buf.getvalue()
        pad = len(data) & 1
        if self.f.tell() + 8 + len(data) + pad + 16 * (len(self.index) + 1) + 8 > self.limit:
            raise RuntimeError(f"{self.path}: the Motion-JPEG fallback writes a plain RIFF/AVI, which ends at 4 GiB ({len(self.index)} frames written so far); "
                               "install ffmpeg for the .mp4 path, or render fewer frames (-n)")
        self.index.append((self.f.tell() - self.movi_at - 8, len(data)))
        self.f.write(b"00dc" + struct.pack("<I", len(data)) + data + b"\0" * pad)

    def close(self):
        n = len(self.index)
        movi_end = self.f.tell()
        idx = b"".join(b"00dc" + struct.pack("<III", 0x10, off, size) for off, size in self.index)
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        end = self.f.tell()
        usec = int(round(1e6 / self.fps))
        biggest = max((s for _, s in self.index), default=0)
        avih = struct.pack("<IIIIIIIIII4I", usec, int(biggest * self.fps), 0, 0x10, n, 0, 1, biggest, self.w, self.h, 0, 0, 0, 0)
        rate, scale = int(round(self.fps * 1000)), 1000
        strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, scale, rate, 0, n, biggest, 0xFFFFFFFF, 0) + struct.pack("<hhhh", 0, 0, self.w, self.h)
        strf = struct.pack("<IiiHHIIiiII", 40, self.w, self.h, 1, 24, 0x47504A4D, self.w * self.h * 3, 0, 0, 0, 0)
        strl = b"LIST" + struct.pack("<I", 4 + 8 + len(strh) + 8 + len(strf)) + b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        hdrl_body = b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + strl
        hdrl = b"LIST" + struct.pack("<I", len(hdrl_body)) + hdrl_body
        junk_len = 224 - 12 - len(hdrl) - 8
        assert junk_len >= 0
        head = b"RIFF" + struct.pack("<I", end - 8) + b"AVI " + hdrl + b"JUNK" + struct.pack("<I", junk_len) + b"\0" * junk_len
        self.f.seek(0)
        self.f.write(head)
        self.f.seek(self.movi_at + 4)
        self.f.write(struct.pack("<I", movi_end - self.movi_at - 8))
        self.f.close()


def open_writer(stem, width, height, fps):
    """The encoder for `<stem>.mp4` (ffmpeg) or, without ffmpeg, `<stem>.avi` (Motion-JPEG).  Returns (writer, path)."""
    if shutil.which("ffmpeg"):
        path = stem + ".mp4"
        return FfmpegWriter(path, width, height, fps), path
    path = stem + ".avi"
    logger.warning(f"ffmpeg is not on PATH: writing Motion-JPEG to {path} instead of the reference's .mp4")
    return MjpegAviWriter(path, width, height, fps), path


def read_mjpeg_avi(path):
    """[frame ndarray, ...] of a file MjpegAviWriter wrote (tests, and a quick look without a player)."""
    import io

    from PIL import Image

    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI "
    at = data.index(b"movi") + 4
    idx_at = data.rindex(b"idx1")
    frames = []
    while at < idx_at and data[at:at + 4] == b"00dc":
        size = struct.unpack_from("<I", data, at + 4)[0]
        frames.append(np.asarray(Image.open(io.BytesIO(data[at + 8:at + 8 + size])).convert("RGB")))
        at += 8 + size + (size & 1)
    return frames


# ---- the two videos ------------------------------------------------------------------------------------------------------------------
def _frame_files(core, img_id):
    out = []
    for cam in GRID_CAMERAS:
        path = core._image_path.format(cam_id=cam, img_id=img_id)
        if not os.path.exists(path):
            path = core._image_path.format(cam_id=cam, img_id=f"{img_id:06d}")
        out.append(path)
    return out


def _grid_frames(core, renderer, batch=16):
    """Yield (img_id, [2 H, 3 W, 3] uint8 cuda) for every image of the recording: files -> device JPEG decode -> one drawing launch."""
    from . import jpeg

    W, H = core.image_shape
    pts = np.stack([core.camNet.cam_list[c].points2d[: core.num_images] for c in GRID_CAMERAS], axis=1)   # (T, 6, J, 2) pixels (row, col)
    pts_dev = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float64)).to(renderer.dev)
    for t0 in range(0, core.num_images, batch):
        ids = range(t0, min(t0 + batch, core.num_images))
        blobs = [open(p, "rb").read() for i in ids for p in _frame_files(core, i)]
        luma = jpeg.decode_luma(blobs, W, H, device=renderer.dev).reshape(len(ids), 6, H, W)
        for k, i in enumerate(ids):
            yield i, renderer.grid2d(luma[k], pts_dev[i])


def _video_stem(core, kind):
    return os.path.join(core.output_folder, f"video_{kind}_" + core.input_folder.replace("/", "_"))


def make_pose2d_video(core, fps=None, progress=None):
    """video_pose2d_<folder>: per image the 2 x 3 camera grid with the 2-D pose drawn on it (reference video.py:21-49).  Returns the path."""
    fps = fps or DEFAULT_FPS
    W, H = core.image_shape
    renderer = FrameRenderer(H, W, config["num_joints"], core.device)
    writer, path = open_writer(_video_stem(core, "pose2d"), 3 * W, 2 * H, fps)
    host = torch.empty((2 * H, 3 * W, 3), dtype=torch.uint8).pin_memory()
    try:
        for _, frame in _grid_frames(core, renderer):
            host.copy_(frame)
            writer.write(host.numpy())
            if progress is not None:
                progress()   # (multi-rank: rank 0's heartbeat to the waiting peers, distributed.primary_section)
    finally:
        writer.close()
    logger.info(f"Video created at {path}\n")
    return path


def make_pose3d_video(core, fps=None, progress=None):
    """video_pose3d_<folder>: two rows of the six camera images with their 2-D pose (200 x 100 each) over a row of three 3-D views of
    the pose `Core.get_points3d` returns (reference video.py:52-82).  Returns the path."""
    fps = fps or DEFAULT_FPS
    W, H = core.image_shape
    renderer = FrameRenderer(H, W, config["num_joints"], core.device)
    pose = torch.from_numpy(merge_stripes(core.get_points3d())).to(renderer.dev)   # (T, J, 3)
    sh, sw = SMALL_2D
    fw, fh = 3 * sw, 2 * sh + PANEL_SIZE
    frame = torch.empty((fh, fw, 3), dtype=torch.uint8, device=renderer.dev)
    writer, path = open_writer(_video_stem(core, "pose3d"), fw, fh, fps)
    host = torch.empty((fh, fw, 3), dtype=torch.uint8).pin_memory()
    try:
        for i, grid in _grid_frames(core, renderer):
            renderer.resize(grid, frame[: 2 * sh])
            renderer.panels3d(pose[i].contiguous(), out=frame[2 * sh:])
            host.copy_(frame)
            writer.write(host.numpy())
            if progress is not None:
                progress()   # (multi-rank: rank 0's heartbeat to the waiting peers, distributed.primary_section)
    finally:
        writer.close()
    logger.info(f"Video created at {path}\n")
    return path
