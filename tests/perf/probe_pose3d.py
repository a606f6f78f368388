"""Timing of the sequence-global tail (a9 Procrustes, get_points3d chain) on the device vs the CPU oracle."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from deepfly3d_amd import ops  # noqa: E402
from deepfly3d_amd.procrustes import template_constants  # noqa: E402
from oracle import geometry as og  # noqa: E402
from oracle import postprocess as pp  # noqa: E402

g3 = np.load("tests/golden/golden_3d.npz")
tmpl = np.load("tests/golden/template.npz")["points3d"]
seg, fit = template_constants()
rng = np.random.default_rng(0)
for T in (1000, 100000):
    X = np.tile(g3["points3d_wo_procrustes"], (T // 15 + 1, 1, 1))[:T] + rng.normal(0, 0.05, size=(T, 38, 3))
    x = torch.as_tensor(X).cuda()
    for _ in range(2):
        p = ops.procrustes(x, seg, fit)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = ops.procrustes(x, seg, fit)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = ops.pose_normalize(p)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    f = ops.oneeuro_filter(n)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    c0 = time.perf_counter()
    po = og.procrustes_separate(X, tmpl)
    c1 = time.perf_counter()
    Tf = min(T, 2000)
    fo = pp.oneeuro_filter(pp.normalize_pose_3d(po)[:Tf])
    c2 = time.perf_counter()
    print(f"T={T}: device procrustes {1e3*(t1-t0):.2f} ms, normalize {1e3*(t2-t1):.2f} ms, one-euro {1e3*(t3-t2):.2f} ms | "
          f"oracle procrustes {1e3*(c1-c0):.1f} ms, one-euro (python, {Tf} frames) {1e3*(c2-c1):.0f} ms | "
          f"max |dev-oracle| procrustes {np.abs(p.cpu().numpy()-po).max():.1e}")
