#!/usr/bin/env python3
"""Golden vectors for the `Core.get_points3d` chain (reference df3d/core.py:332-343):

    procrustes_seperate -> normalize_pose_3d(rotate=True) -> filter_batch        (One-Euro filter)

produced by *importing and executing* the reference's own modules (df3d.procrustes, df3d.plot_util,
df3d.signal_util) in the build container:

    python tests/golden/make_golden_post.py

Outputs (pure data): pose_chain_golden.npz (the 15-frame golden sequence), pose_chain_jitter.npz (a 46-frame
seeded sequence: even length, so the medians average two elements), oneeuro_random.npz (filter_batch alone on a
seeded random walk of 300 frames).  Nothing at test/bench time reads /root/reference.
"""
import os
import pickle
import sys

import numpy as np

REF = os.environ.get("DF3D_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from df3d.plot_util import normalize_pose_3d
    from df3d.procrustes import procrustes_seperate
    from df3d.signal_util import filter_batch

    with open(os.path.join(REF, "tests/data/reference_df3d/df3d_result_3d.pkl"), "rb") as f:
        g3 = pickle.load(f)
    wo = np.asarray(g3["points3d_wo_procrustes"], dtype=np.float64)
    rng = np.random.default_rng(20260928)

    def chain(x):
        p = procrustes_seperate(np.copy(x))
        n = normalize_pose_3d(np.copy(p), rotate=True)
        return p, np.copy(n), filter_batch(np.copy(n))

    p, n, f = chain(wo)
    np.savez(os.path.join(OUT, "pose_chain_golden.npz"), inp=wo, procrustes=p, normalized=n, filtered=f)

    wo2 = np.concatenate([np.tile(wo, (3, 1, 1)), wo[:1]]) + rng.normal(0.0, 0.02, size=(46, 38, 3))
    wo2[7, 22] = 0.0
    p, n, f = chain(wo2)
    np.savez(os.path.join(OUT, "pose_chain_jitter.npz"), inp=wo2, procrustes=p, normalized=n, filtered=f)

    walk = np.cumsum(rng.normal(0.0, 0.05, size=(300, 38, 3)), axis=0)
    walk[100:103] += 3.0  # a jump, so the adaptive cut-off is exercised
    np.savez(os.path.join(OUT, "oneeuro_random.npz"), inp=walk, out=filter_batch(np.copy(walk)))
    print("written")


if __name__ == "__main__":
    main()
