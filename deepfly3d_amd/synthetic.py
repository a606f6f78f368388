"""Seeded synthetic inputs for benchmarks and smoke tests (no network, no trained weights offline).

* hourglass parameters with the df2d/bearpaw state_dict key names: He-normal convolution
  weights scaled by `gain`, BN gamma ~ U[0.5, 1.5], beta / running_mean ~ N(0, 0.1), running_var ~ U[0.5, 1.5].
* geometry-consistent 2-D detections for the triangulation / bundle-adjustment stages: a 3-D
  pose sequence projected through given cameras, quantised to the 64 x 128 heat-map grid, with the
  reference's visibility pattern (reference df3d/core.py:187-203).
"""
import numpy as np


def _bottleneck_shapes(prefix, cin, planes):
    cout = 2 * planes
    out = []
    for bn, c in (("bn1", cin), ("bn2", planes), ("bn3", planes)):
        out += [(f"{prefix}.{bn}.weight", (c,)), (f"{prefix}.{bn}.bias", (c,)),
                (f"{prefix}.{bn}.running_mean", (c,)), (f"{prefix}.{bn}.running_var", (c,))]
    out += [(f"{prefix}.conv1.weight", (planes, cin, 1, 1)), (f"{prefix}.conv1.bias", (planes,)),
            (f"{prefix}.conv2.weight", (planes, planes, 3, 3)), (f"{prefix}.conv2.bias", (planes,)),
            (f"{prefix}.conv3.weight", (cout, planes, 1, 1)), (f"{prefix}.conv3.bias", (cout,))]
    if cin != cout:
        out += [(f"{prefix}.downsample.0.weight", (cout, cin, 1, 1)), (f"{prefix}.downsample.0.bias", (cout,))]
    return out


def hourglass_param_shapes(num_stacks=2, num_classes=19, feats=128, depth=4):
    ch = 2 * feats
    shapes = [("conv1.weight", (64, 3, 7, 7)), ("conv1.bias", (64,)),
              ("bn1.weight", (64,)), ("bn1.bias", (64,)), ("bn1.running_mean", (64,)), ("bn1.running_var", (64,))]
    shapes += _bottleneck_shapes("layer1.0", 64, 64)
    shapes += _bottleneck_shapes("layer2.0", 128, feats)
    shapes += _bottleneck_shapes("layer3.0", ch, feats)
    for s in range(num_stacks):
        for lvl in range(depth):
            for k in range(4 if lvl == 0 else 3):
                shapes += _bottleneck_shapes(f"hg.{s}.hg.{lvl}.{k}.0", ch, feats)
        shapes += _bottleneck_shapes(f"res.{s}.0", ch, feats)
        shapes += [(f"fc.{s}.0.weight", (ch, ch, 1, 1)), (f"fc.{s}.0.bias", (ch,)),
                   (f"fc.{s}.1.weight", (ch,)), (f"fc.{s}.1.bias", (ch,)),
                   (f"fc.{s}.1.running_mean", (ch,)), (f"fc.{s}.1.running_var", (ch,))]
        shapes += [(f"score.{s}.weight", (num_classes, ch, 1, 1)), (f"score.{s}.bias", (num_classes,))]
        if s < num_stacks - 1:
            shapes += [(f"fc_.{s}.weight", (ch, ch, 1, 1)), (f"fc_.{s}.bias", (ch,)),
                       (f"score_.{s}.weight", (ch, num_classes, 1, 1)), (f"score_.{s}.bias", (ch,))]
    return shapes


def synthetic_state_dict(seed=0, num_stacks=2, gain=0.6):
    """{name: float32 ndarray}; deterministic in `seed`."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in hourglass_param_shapes(num_stacks):
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            std = (2.0 / fan_in) ** 0.5 * (0.5 * gain if name.endswith("conv3.weight") else gain)
            sd[name] = (rng.standard_normal(shape) * std).astype(np.float32)
        elif name.endswith("running_var") or (name.endswith(".weight") and len(shape) == 1):
            sd[name] = (rng.random(shape) + 0.5).astype(np.float32)
        else:
            sd[name] = (rng.standard_normal(shape) * 0.1).astype(np.float32)
    return sd


def synthetic_points2d(points3d, R, tvec, intr, camera_ordering=(0, 1, 2, 3, 4, 5, 6), image_shape=(960, 480),
                       heatmap_shape=(64, 128)):
    """Project a (T, 38, 3) pose through the cameras, quantise to the heat-map grid and apply the reference's
    visibility layout.  Returns points2d (7, T, 38, 2) float64, normalised (row, col)."""
    X = np.asarray(points3d, np.float64)
    T = X.shape[0]
    W, H = image_shape
    out = np.zeros((7, T, 38, 2))
    o = list(camera_ordering)
    for pos, cam in enumerate(o):
        if pos == 3:
            continue
        Xc = np.einsum("ij,tkj->tki", R[cam], X) + tvec[cam]
        u = intr[cam][0, 0] * Xc[..., 0] / Xc[..., 2] + intr[cam][0, 2]
        v = intr[cam][1, 1] * Xc[..., 1] / Xc[..., 2] + intr[cam][1, 2]
        row = np.clip(np.round(v / H * heatmap_shape[0]), 1, heatmap_shape[0] - 1) / heatmap_shape[0]
        col = np.clip(np.round(u / W * heatmap_shape[1]), 1, heatmap_shape[1] - 1) / heatmap_shape[1]
        sl = slice(0, 19) if pos < 3 else slice(19, 38)
        out[cam, :, sl, 0] = row[:, sl]
        out[cam, :, sl, 1] = col[:, sl]
    out[o[2], :, 15:] = 0
    out[o[4], :, 34:] = 0
    # the reference leaves "unseen" joints of the left-side cameras at (0, 1) after its un-flip
    for pos in (4, 5, 6):
        unseen = out[o[pos], ..., 0] == 0
        out[o[pos], ..., 1] = np.where(unseen, 1.0, out[o[pos], ..., 1])
    return out


def synthetic_ba_window(pose, R, tvec, intr, window, seed, index=0, image_shape=(960, 480)):
    """Geometry-consistent detections of one bundle-adjustment window in PIXELS (row, col), shape (7, window, 38, 2):
    the (T0, 38, 3) `pose` tiled to `window` frames + N(0, 0.05 mm) jitter seeded by (seed, index), projected through
    the cameras and quantised to the heat-map grid (the BA stage of BASELINE configs[4])."""
    pose = np.asarray(pose, np.float64)
    rng = np.random.default_rng([int(seed), int(index)])
    X = np.tile(pose, (window // pose.shape[0] + 1, 1, 1))[:window] + rng.normal(0, 0.05, size=(window, 38, 3))
    return synthetic_points2d(X, R, tvec, intr, image_shape=image_shape) * np.array([float(image_shape[1]), float(image_shape[0])])
