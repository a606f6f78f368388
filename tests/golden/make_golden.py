#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

The outputs are pure data (npz): the reference's own committed golden pickles
(tests/data/reference_df3d/*.pkl, data/*.pkl) re-encoded without pickle, plus
input/output vectors produced by *importing and executing* the reference's own
Python modules on seeded inputs:

  * relayout_*.npz   - df3d.core.Core.pose2d_estimation (reference core.py:170-203)
                       run with a stubbed df2d.inference.inference_folder, for
                       three camera orderings.
  * procrustes_*.npz - df3d.procrustes.procrustes_seperate (reference
                       procrustes.py:51-89) on the golden and on a seeded
                       perturbed 3-D sequence.

No -m gpu test, bench.py or smoke() reads /root/reference at run time (tests/test_shims.py, a CPU test of the import
seam, does and is skipped where the checkout is absent); they
read only these npz files.  No reference source text is stored here.
"""
import os
import pickle
import sys
import types

import numpy as np

REF = os.environ.get("DF3D_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_pkl(rel):
    with open(os.path.join(REF, rel), "rb") as f:
        return pickle.load(f)


def cams_to_arrays(d):
    R = np.stack([np.asarray(d[c]["R"], dtype=np.float64) for c in range(7)])
    t = np.stack([np.asarray(d[c]["tvec"], dtype=np.float64) for c in range(7)])
    K = np.stack([np.asarray(d[c]["intr"], dtype=np.float64) for c in range(7)])
    dist = np.stack([np.asarray(d[c]["distort"], dtype=np.float64) for c in range(7)])
    return dict(R=R, tvec=t, intr=K, distort=dist)


def main():
    # ---- 1. the reference's committed golden artefacts, re-encoded -------------------------
    calib = load_pkl("data/calib.pkl")
    np.savez(os.path.join(OUT, "calib.npz"), **cams_to_arrays(calib))

    template = load_pkl("data/df3d_result.pkl")
    np.savez(
        os.path.join(OUT, "template.npz"),
        points3d=np.asarray(template["points3d"], dtype=np.float64),
        points2d=np.asarray(template["points2d"], dtype=np.float64),
        **cams_to_arrays(template),
    )

    g2 = load_pkl("tests/data/reference_df3d/df3d_result_2d.pkl")
    np.savez(
        os.path.join(OUT, "golden_2d.npz"),
        points2d=g2["points2d"],
        camera_ordering=g2["camera_ordering"],
        heatmap_confidence=g2["heatmap_confidence"],
        key_order=np.array([str(k) for k in g2.keys()]),
    )

    for name, rel in (
        ("golden_3d", "tests/data/reference_df3d/df3d_result_3d.pkl"),
        ("golden_3d_run2", "tests/data/reference_df3d/df3d_result.pkl"),
    ):
        g3 = load_pkl(rel)
        np.savez(
            os.path.join(OUT, name + ".npz"),
            points3d=g3["points3d"],
            points2d=g3["points2d"],
            points3d_wo_procrustes=g3["points3d_wo_procrustes"],
            camera_ordering=g3["camera_ordering"],
            heatmap_confidence=g3["heatmap_confidence"],
            key_order=np.array([str(k) for k in g3.keys()]),
            cam_key_order=np.array([str(k) for k in g3[0].keys()]),
            **cams_to_arrays(g3),
        )

    # ---- 2. vectors produced by executing the reference's own modules -----------------------
    sys.path.insert(0, REF)
    import matplotlib

    matplotlib.use("Agg")

    # stub the two absent third-party back-ends so df3d.core imports
    captured = {}

    def fake_inference_folder(**kw):
        captured["kwargs"] = {k: (list(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in kw.items()}
        return captured["pts"].copy(), captured["conf"].copy()

    df2d = types.ModuleType("df2d")
    df2d_inf = types.ModuleType("df2d.inference")
    df2d_inf.inference_folder = fake_inference_folder
    df2d.inference = df2d_inf
    pyba = types.ModuleType("pyba")
    pyba_cn = types.ModuleType("pyba.CameraNetwork")
    pyba_cn.CameraNetwork = object
    pyba.CameraNetwork = pyba_cn
    sys.modules.update({"df2d": df2d, "df2d.inference": df2d_inf, "pyba": pyba, "pyba.CameraNetwork": pyba_cn})

    import df3d.core as ref_core  # noqa: E402  (the reference's own module)
    from df3d.procrustes import procrustes_seperate as ref_procrustes  # noqa: E402

    rng = np.random.default_rng(0)
    T = 6
    for tag, order in (("id", [0, 1, 2, 3, 4, 5, 6]), ("rev", [6, 5, 4, 3, 2, 1, 0]), ("clc", [0, 6, 5, 4, 3, 2, 1])):
        rows = rng.integers(0, 64, size=(7, T, 19, 1)) / 64.0
        cols = rng.integers(0, 128, size=(7, T, 19, 1)) / 128.0
        pts = np.concatenate([rows, cols], axis=-1).astype(np.float32)
        conf = rng.random((7, T, 19, 1)).astype(np.float32)
        captured["pts"], captured["conf"] = pts, conf
        core = ref_core.Core.__new__(ref_core.Core)  # bypass __init__ (needs a folder)
        core._input_folder = "/nonexistent"
        core.camera_ordering = np.array(order)
        core.max_img_id = T - 1
        core.pose2d_estimation(batch_size=8, disable_pin_memory=False)
        kw = captured["kwargs"]
        np.savez(
            os.path.join(OUT, f"relayout_{tag}.npz"),
            camera_ordering=np.array(order),
            in_points2d=pts,
            in_conf=conf,
            out_points2d=core.points2d,
            out_conf=core.conf,
            camera_ids_to_flip=np.array(kw["camera_ids_to_flip"]),
            max_img_id=np.array(kw["max_img_id"]),
            batch_size=np.array(kw["batch_size"]),
        )

    g3 = load_pkl("tests/data/reference_df3d/df3d_result_3d.pkl")
    wo = np.asarray(g3["points3d_wo_procrustes"], dtype=np.float64)
    out = ref_procrustes(wo.copy())
    np.savez(os.path.join(OUT, "procrustes_golden.npz"), inp=wo, out=out)
    # a perturbed, longer sequence (tile + seeded jitter) so medians are not trivially the golden's
    wo2 = np.tile(wo, (3, 1, 1)) + rng.normal(0.0, 0.02, size=(45, 38, 3))
    wo2[5, 3] = 0.0  # an untriangulated joint, as the reference would leave it
    out2 = ref_procrustes(wo2.copy())
    np.savez(os.path.join(OUT, "procrustes_jitter.npz"), inp=wo2, out=out2)

    # ---- 3. a few reference jpgs for IO plumbing (data files of the reference's own tests) ---
    import shutil

    os.makedirs(os.path.join(OUT, "images"), exist_ok=True)
    for cam in range(7):
        for img in (0, 1):
            shutil.copy(
                os.path.join(REF, "tests/data/reference", f"camera_{cam}_img_{img}.jpg"),
                os.path.join(OUT, "images", f"camera_{cam}_img_{img}.jpg"),
            )
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
