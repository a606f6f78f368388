"""CPU restatement of the input front-end's arithmetic (flip -> resize -> grey to 3 channels -> (v / 255 - mean) / std).
TEST INFRASTRUCTURE ONLY -- the product package never imports this file.

PARITY UNPINNED w.r.t. the reference: the code lives in the un-vendored `nely-df2d` (dataset class behind the call site
reference df3d/core.py:177-185); the reference checkout only names the mean's file (df3d/config.py:37-39).  The resize rule
is therefore data on the product side (deepfly3d_amd.inference.PREPROCESS["resize"]) and this file restates each
candidate rule independently of the HIP code:
    bilinear                 torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=False)
                             (= cv2.INTER_LINEAR: half-pixel centres)
    bilinear_align_corners   torch.nn.functional.interpolate(mode="bilinear", align_corners=True)
    area                     cv2.INTER_AREA for a down-scale, restated from its definition (cv2 is not in this image): the
                             output pixel is the mean of the source rectangle [o*s, (o+1)*s) it covers, every source pixel
                             weighted by its overlap with that rectangle -- two banded weight matrices, float64
"""
import numpy as np
import torch
import torch.nn.functional as F

RESIZE_RULES = ("bilinear", "bilinear_align_corners", "area")


def _area_matrix(n_in, n_out):
    s = n_in / n_out
    w = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        a, b = o * s, min((o + 1) * s, n_in)
        for i in range(int(np.floor(a)), min(int(np.ceil(b)), n_in)):
            w[o, i] = min(b, i + 1) - max(a, i)
        w[o] /= b - a
    return w


def resize(x, out_hw, rule):
    """x: float tensor [n, H, W] -> [n, OH, OW] (float64 for "area", float32 otherwise)."""
    if rule == "bilinear":
        return F.interpolate(x[:, None].float(), size=out_hw, mode="bilinear", align_corners=False, antialias=False)[:, 0]
    if rule == "bilinear_align_corners":
        return F.interpolate(x[:, None].float(), size=out_hw, mode="bilinear", align_corners=True)[:, 0]
    if rule == "area":
        wy = torch.from_numpy(_area_matrix(x.shape[1], out_hw[0]))
        wx = torch.from_numpy(_area_matrix(x.shape[2], out_hw[1]))
        return torch.einsum("oy,nyx,px->nop", wy, x.double(), wx)
    raise ValueError(rule)


def preprocess_u8(frames_u8, flip, out_hw=(256, 512), mean=(0.22, 0.22, 0.22), std=(1.0, 1.0, 1.0), rule="bilinear"):
    """frames_u8 [n, H, W] or [n, H, W, C] uint8 (tensor / array), flip [n] bool or None -> float32 NHWC [n, OH, OW, 3]."""
    x = torch.as_tensor(np.asarray(frames_u8))
    if x.dim() == 3:
        x = x[..., None]
    n, H, W, C = x.shape
    x = x.expand(n, H, W, 3) if C == 1 else x
    outs = []
    for c in range(3):
        p = x[..., c].double()
        if flip is not None:
            f = torch.as_tensor(np.asarray(flip)).bool()
            p = torch.where(f[:, None, None], p.flip(-1), p)
        r = resize(p if rule == "area" else p.float(), out_hw, rule).double()
        outs.append((r / 255.0 - mean[c]) / std[c])
    return torch.stack(outs, dim=-1).float()
