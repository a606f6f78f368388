"""a1: the `df2d.inference`-shaped entry point `df3d.core.Core.pose2d_estimation` calls
(reference df3d/core.py:177-185):

    points2d, conf = inference_folder(folder=..., camera_ids_to_flip=[...], return_heatmap=False,
                                      return_confidence=True, max_img_id=..., batch_size=8,
                                      disable_pin_memory=False)
    points2d: (7, T, 19, 2) float32 normalised (row/64, col/128);  conf: (7, T, 19, 1) float32

Pipeline on the device: JPEG bytes -> df3d_jpeg_decode_luma -> df3d_preprocess_u8 (flip / resize / normalise) ->
df3d_hg_forward (stacked hourglass) -> df3d_heatmap_argmax.  The host lists and reads files.  No CPU fallback.

Weights: a bearpaw/df2d `state_dict` checkpoint (`sh8_deepfly.tar`, reference df3d/config.py:30-32) is looked
up in $DF3D_WEIGHTS or deepfly3d_amd/weights/.  It is not redistributable offline; for plumbing tests set
DF3D_SYNTHETIC_WEIGHTS=<seed> to use seeded synthetic parameters (the results are then meaningless poses).
"""
import ctypes
import os

import numpy as np
import torch

from . import _native, ops
from .config import config
from .hourglass import HourglassEngine
from .os_util import image_path_for

_HERE = os.path.dirname(os.path.abspath(__file__))
# df2d's preprocessing is not in the reference checkout ("parity unpinned"), so all of it is DATA:
#   mean / std   the reference names the source of the mean: `weights/mean.pth.tar` next to the checkpoint (reference
#                df3d/config.py:37-39; bearpaw's dataset cache {'mean': tensor[3], 'std': tensor[3]}).  load_state_dict()
#                reads that file when it finds it; bearpaw's `color_normalize` subtracts the mean and does NOT divide, so the
#                file's std is only applied with "divide_by_std": true.  Without the file: mean 0.22 as recalled.
#   resize       "bilinear" (half-pixel centres, no antialias: cv2.INTER_LINEAR) | "bilinear_align_corners" | "area"
#                (cv2.INTER_AREA); the rule df2d uses for 960x480 -> 512x256 is unknown here.
# DF3D_PREPROCESS='{"mean": [m, m, m], "std": [s, s, s], "resize": "area", "divide_by_std": false}' overrides any of
# them without a code change (every key optional); the dormant reference-pin test sweeps the resize rules.
PREPROCESS = {"mean": (0.22, 0.22, 0.22), "std": (1.0, 1.0, 1.0), "resize": "bilinear"}
_PREPROCESS_SOURCE = {"mean": "default (recalled)", "resize": "default"}
_warned_preprocess = []


def _three(v):
    v = [float(x) for x in np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float64).reshape(-1)]
    if len(v) == 1:
        v = v * 3
    if len(v) != 3:
        raise ValueError(f"expected 1 or 3 values, got {len(v)}")
    return tuple(v)


def _apply_env_preprocess():
    if not os.environ.get("DF3D_PREPROCESS"):
        return {}
    import json as _json

    p = _json.loads(os.environ["DF3D_PREPROCESS"])
    unknown = set(p) - {"mean", "std", "resize", "divide_by_std"}
    if unknown:
        raise ValueError(f"DF3D_PREPROCESS: unknown keys {sorted(unknown)}")
    if "mean" in p:
        PREPROCESS["mean"] = _three(p["mean"])
        _PREPROCESS_SOURCE["mean"] = "DF3D_PREPROCESS"
    if "std" in p:
        PREPROCESS["std"] = _three(p["std"])
    if "resize" in p:
        if p["resize"] not in _native.RESIZE_MODES:
            raise ValueError(f"DF3D_PREPROCESS: resize must be one of {sorted(_native.RESIZE_MODES)}")
        PREPROCESS["resize"] = p["resize"]
        _PREPROCESS_SOURCE["resize"] = "DF3D_PREPROCESS"
    return p


_env_preprocess = _apply_env_preprocess()

_engine_cache = {}


def load_mean_file(path):
    """bearpaw / df2d's `mean.pth.tar` ({'mean': tensor, 'std': tensor}) -> (mean3, std3) as float tuples."""
    meta = torch.load(path, map_location="cpu", weights_only=False)
    if not (isinstance(meta, dict) and "mean" in meta):
        raise ValueError(f"{path}: expected a dict with 'mean' (and 'std')")
    return _three(meta["mean"]), (_three(meta["std"]) if "std" in meta else (1.0, 1.0, 1.0))


def _adopt_mean_file(checkpoint):
    """The normalisation mean that belongs to `checkpoint`: $DF3D_MEAN, or mean.pth.tar beside it (reference
    df3d/config.py:37-39).  An explicit DF3D_PREPROCESS mean wins.  Returns the file used, or None."""
    cands = [os.environ.get("DF3D_MEAN"), os.path.join(os.path.dirname(os.path.abspath(checkpoint)), "mean.pth.tar")]
    for cand in cands:
        if cand and os.path.exists(cand):
            mean, std = load_mean_file(cand)
            if "mean" not in _env_preprocess:
                PREPROCESS["mean"] = mean
                _PREPROCESS_SOURCE["mean"] = cand
            if _env_preprocess.get("divide_by_std") and "std" not in _env_preprocess:
                PREPROCESS["std"] = std
            return cand
    return None


def load_state_dict(path=None):
    """Locate and load the hourglass parameters ({name: array}); with a trained checkpoint also its normalisation mean
    (mean.pth.tar beside it)."""
    if os.environ.get("DF3D_SYNTHETIC_WEIGHTS") is not None:
        from .synthetic import synthetic_state_dict

        return synthetic_state_dict(int(os.environ["DF3D_SYNTHETIC_WEIGHTS"]))
    cands = [path, os.environ.get("DF3D_WEIGHTS"), os.path.join(_HERE, "weights", "sh8_deepfly.tar")]
    for cand in cands:
        if cand and os.path.exists(cand):
            mean_file = _adopt_mean_file(cand)
            if not _warned_preprocess and (mean_file is None and "mean" not in _env_preprocess or _PREPROCESS_SOURCE["resize"] == "default"):
                _warned_preprocess.append(True)
                from . import logger

                logger.warning(
                    f"Trained weights {cand}: input preprocessing is only partly pinned -- mean {PREPROCESS['mean']} from "
                    f"{_PREPROCESS_SOURCE['mean']}, std {PREPROCESS['std']}, resize '{PREPROCESS['resize']}' ({_PREPROCESS_SOURCE['resize']}).  df2d's "
                    "constants are not in the reference checkout: put mean.pth.tar beside the checkpoint (reference df3d/config.py:37-39) "
                    'and/or set DF3D_PREPROCESS=\'{"mean": [..], "std": [..], "resize": "bilinear|bilinear_align_corners|area"}\'; '
                    "tests/test_gpu_reference_pin.py reports which resize rule meets the reference's bars.")
            ckpt = torch.load(cand, map_location="cpu", weights_only=False)
            sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
            return {k[len("module.") :] if k.startswith("module.") else k: v for k, v in sd.items()}
    raise FileNotFoundError(
        "hourglass weights not found: put sh8_deepfly.tar under deepfly3d_amd/weights/ or set DF3D_WEIGHTS "
        "(or DF3D_SYNTHETIC_WEIGHTS=<seed> for plumbing tests)"
    )


def get_engine(dtype="f32", device=None, state_dict=None):
    _native.require_gpu()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    key = (dtype, str(dev), id(state_dict) if state_dict is not None else None)
    if key not in _engine_cache:
        sd = state_dict if state_dict is not None else load_state_dict()
        _engine_cache[key] = HourglassEngine(sd, dtype=dtype, device=dev, num_stacks=config["num_stacks"])
    return _engine_cache[key]


def preprocess_u8(frames_u8, flip, out_hw=(256, 512)):
    """frames_u8 [n, H, W] or [n, H, W, C] uint8 cuda; flip [n] uint8/bool cuda or None -> float32 NHWC [n, OH, OW, 3]."""
    lib = _native.load()
    if frames_u8.dim() == 3:
        frames_u8 = frames_u8.unsqueeze(-1)
    if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()):
        raise ValueError("frames must be a contiguous uint8 CUDA tensor")
    n, H, W, C = frames_u8.shape
    out = torch.empty((n, out_hw[0], out_hw[1], 3), dtype=torch.float32, device=frames_u8.device)
    mean = (ctypes.c_float * 3)(*PREPROCESS["mean"])
    std = (ctypes.c_float * 3)(*PREPROCESS["std"])
    fl = None
    if flip is not None:
        fl = flip.to(device=frames_u8.device, dtype=torch.uint8).contiguous()
    with torch.cuda.device(frames_u8.device):  # kernels launch on the current HIP device
        _native.check(
            lib.df3d_preprocess_u8(frames_u8.data_ptr(), fl.data_ptr() if fl is not None else None, n, H, W, C, out.data_ptr(), out_hw[0], out_hw[1],
                                   mean, std, _native.RESIZE_MODES[PREPROCESS["resize"]], torch.cuda.current_stream(frames_u8.device).cuda_stream),
            "df3d_preprocess_u8",
        )
    return out


def inference_views(images, engine, return_heatmap=False):
    """images: float32 NHWC [n, 256, 512, 3] cuda -> (points [n, 19, 2], conf [n, 19]) (+ heat-maps)."""
    hm = engine.forward(images)
    pts, conf = ops.heatmap_argmax(hm, nonfinite=engine.nonfinite_planes)   # (the overflow guard: engine.check_finite() reads the counter)
    return (pts, conf, hm) if return_heatmap else (pts, conf)


def inference_frames(frames_u8, flip, engine, return_heatmap=False):
    """frames_u8: uint8 [n, H, W(, C)] cuda camera frames, flip [n] uint8 or None -> (points [n, 19, 2], conf [n, 19]) (+ heat-maps):
    `inference_views(preprocess_u8(frames, flip), engine)` with the resize / normalisation done inside the network's first kernel."""
    if tuple(config["input_shape"]) != (engine.height, engine.width):
        raise ValueError("engine input size differs from config['input_shape']")
    hm = engine.forward_u8(frames_u8, flip, PREPROCESS["mean"], PREPROCESS["std"], resize=PREPROCESS["resize"])
    pts, conf = ops.heatmap_argmax(hm, nonfinite=engine.nonfinite_planes)
    return (pts, conf, hm) if return_heatmap else (pts, conf)


def _image_size(path):
    """(width, height) from the JPEG header (host IO only; pixels are decoded on the device)."""
    from PIL import Image

    with Image.open(path) as im:
        return im.size


# views per device batch: the reference's `batch_size` (8) is a lower bound, results do not depend on the batch
DEVICE_BATCH_VIEWS = 896


def inference_folder(folder, camera_ids_to_flip=(), return_heatmap=False, return_confidence=True, max_img_id=None,
                     batch_size=8, disable_pin_memory=False, dtype="f32", device=None, state_dict=None, frame_range=None,
                     as_device_tensors=False):
    """Drop-in for df2d.inference.inference_folder (see module docstring).  Host work: listing and reading the
    files.  Device work: JPEG decode (csrc/jpeg.hip), flip / resize / normalise, hourglass, arg-max.
    `frame_range=(t0, t1)` (multi-GPU sharding) restricts the call to images t0 <= id < t1; the outputs then have
    t1 - t0 frames.  `as_device_tensors=True` returns the results as CUDA tensors instead of numpy arrays (the
    multi-GPU path gathers them without a host round trip)."""
    from .jpeg import JpegFolderReader

    _native.require_gpu()
    if max_img_id is None:
        from .os_util import get_max_img_id

        max_img_id = get_max_img_id(folder)
    t_first, t_stop = (0, max_img_id + 1) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
    T = max(0, t_stop - t_first)
    ncam = config["num_cameras"]
    if T == 0 and frame_range is not None:  # an empty shard
        out = [np.zeros((ncam, 0, config["num_predict"], 2), np.float32)]
        if return_heatmap:
            out.append(np.zeros((ncam, 0, config["num_predict"], config["input_shape"][0] // 4, config["input_shape"][1] // 4), np.float32))
        if return_confidence:
            out.append(np.zeros((ncam, 0, config["num_predict"], 1), np.float32))
        if as_device_tensors:
            dev0 = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
            out = [torch.from_numpy(a).to(dev0) for a in out]
        return tuple(out) if len(out) > 1 else out[0]
    engine = get_engine(dtype=dtype, device=device, state_dict=state_dict)
    dev = engine.device
    flip_set = set(int(c) for c in camera_ids_to_flip)
    items = [(c, t) for c in range(ncam) for t in range(T)]  # t is relative to t_first
    points = torch.empty((ncam, T, config["num_predict"], 2), dtype=torch.float32, device=dev)
    conf = torch.empty((ncam, T, config["num_predict"], 1), dtype=torch.float32, device=dev)
    points_flat, conf_flat = points.view(ncam * T, config["num_predict"], 2), conf.view(ncam * T, config["num_predict"], 1)
    heat = [] if return_heatmap else None
    bs = max(1, int(batch_size), DEVICE_BATCH_VIEWS if not return_heatmap else 1)
    # a short first batch gets the GPU going while the reader threads fetch the first full-size one
    first = min(bs, 224)
    starts = [0] + list(range(first, len(items), bs))
    chunks = [items[lo:hi] for lo, hi in zip(starts, starts[1:] + [len(items)]) if hi > lo]
    if not chunks:
        raise FileNotFoundError(f"no images to process in {folder}")
    class _Batches:
        """Paths of batch k, resolved when the reader asks for them (two batches ahead of the GPU): the per-file existence
        checks of a long recording then run under the previous batches' device work instead of in front of the first one."""

        def __len__(self):
            return len(chunks)

        def __getitem__(self, k):
            return [image_path_for(folder, c, t_first + t) for c, t in chunks[k]]

    paths = _Batches()
    width, height = _image_size(image_path_for(folder, chunks[0][0][0], t_first + chunks[0][0][1]))
    reader = JpegFolderReader(width, height, dev, pinned=not disable_pin_memory, batch_capacity=max(len(c) for c in chunks))
    done = False
    try:  # the reader's threads and pinned buffers are released on every path (df3d-cli -r/-f continues after a failed folder)
        with torch.cuda.device(dev):
            flips = [torch.from_numpy(np.fromiter((1 if c in flip_set else 0 for c, _ in chunk), dtype=np.uint8, count=len(chunk))).to(dev, non_blocking=True)
                     for chunk in chunks]
            # file reads two batches ahead, H2D + JPEG decode one batch ahead on a second stream, under this batch's hourglass
            for k, luma in enumerate(reader.stream(paths)):
                chunk = chunks[k]
                if k == 0 and engine.dtype != "f32":
                    # a reduced-precision engine proves itself on the recording's first views against the exact engine (same weights) before its
                    # results count: saturated / overflowed activations give finite, wrong heat-maps (HourglassEngine.canary)
                    exact = get_engine(dtype="f32", device=device, state_dict=state_dict)
                    ns = min(7, luma.shape[0])
                    engine.canary(exact, lambda e: e.forward_u8(luma[:ns], flips[0][:ns], PREPROCESS["mean"], PREPROCESS["std"], resize=PREPROCESS["resize"]),
                                  what=f"the first {ns} views of {folder}")
                res = inference_frames(luma, flips[k], engine, return_heatmap=return_heatmap)
                # items are (camera, frame) in camera-major order = the flat order of points[ncam, T]: contiguous copies
                lo = starts[k]
                points_flat[lo : lo + len(chunk)] = res[0]
                conf_flat[lo : lo + len(chunk), :, 0] = res[1]
                if return_heatmap:
                    heat.append(res[2].cpu())
        done = True
    finally:
        reader.finish(check=done)   # a bad frame raises JpegDecodeError from inside the loop, within two batches of it
    engine.check_finite(f"{folder} (frames {t_first}..{t_stop - 1})")   # a reduced-precision engine that overflowed: an error, not a pickle
    host = (lambda t: t) if as_device_tensors else (lambda t: t.cpu().numpy())
    out = [host(points)]
    if return_heatmap:
        hm = torch.cat(heat)
        hm = hm.reshape(ncam, T, *hm.shape[1:])
        out.append(hm.to(dev) if as_device_tensors else hm.numpy())
    if return_confidence:
        out.append(host(conf))
    return tuple(out) if len(out) > 1 else out[0]
