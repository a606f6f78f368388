"""Constants of the fly set-up (values from reference df3d/config.py:15-69 and df3d/skeleton_fly.py)."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

config = {
    "name": "fly",
    "num_cameras": 7,                 # reference config.py:17
    "heatmap_shape": [64, 128],       # reference config.py:18
    "left_cameras": [0, 1, 2],
    "right_cameras": [6, 5, 4],
    "num_stacks": 2,                  # reference config.py:33
    "flip_cameras": [4, 5, 6],
    "num_joints": 38,                 # reference skeleton_fly.py (2 x 19)
    "num_predict": 19,                # reference config.py:36
    "input_shape": [256, 512],        # network input (rows, cols): 4 x the heat-map
    "calib_path": os.path.join(_HERE, "data", "calib.npz"),                       # reference data/calib.pkl
    "procrustes_template": os.path.join(_HERE, "data", "procrustes_template.npz"),  # reference data/df3d_result.pkl
    "procrustes_apply": True,
    # frames per bundle-adjustment problem on the Core/CLI path (0 = no limit): longer recordings are sub-sampled
    # with an even stride (camera_network.py:bundle_adjust)
    "ba_max_images": 1000,
}

# tracked-point class per joint of ONE side (19 joints): 3 legs x (body-coxa, coxa-femur, femur-tibia,
# tibia-tarsus, tarsus-tip), antenna, 3 stripes           (reference skeleton_fly.py:16-55)
BODY_COXA, COXA_FEMUR, FEMUR_TIBIA, TIBIA_TARSUS, TARSUS_TIP, ANTENNA, STRIPE = range(7)
TRACKED_SIDE = [BODY_COXA, COXA_FEMUR, FEMUR_TIBIA, TIBIA_TARSUS, TARSUS_TIP] * 3 + [ANTENNA, STRIPE, STRIPE, STRIPE]
TRACKED = TRACKED_SIDE * 2


def load_calibration():
    """{cam_id: {R, tvec, intr, distort}} as in the reference's data/calib.pkl."""
    d = np.load(config["calib_path"])
    return {c: {"R": d["R"][c].copy(), "tvec": d["tvec"][c].copy(), "intr": d["intr"][c].copy(), "distort": d["distort"][c].copy()} for c in range(7)}


def load_procrustes_template():
    return np.load(config["procrustes_template"])["points3d"]
