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


# ---- drawing tables of the 38-joint skeleton (what pyba.config.df3d_bones / df3d_colors hold for the reference's
# Core.plot_2d, reference df3d/core.py:311-319).  Joints of one side: three legs of five joints, antenna, three stripes.
def skeleton_bones():
    """[[joint_a, joint_b], ...]: consecutive joints of every leg and of the stripe chain, both sides."""
    bones = []
    for side in (0, 19):
        for leg in range(3):
            bones += [[side + 5 * leg + k, side + 5 * leg + k + 1] for k in range(4)]
        bones += [[side + 16, side + 17], [side + 17, side + 18]]
    return bones


def limb_of_joint(joint):
    """Limb id 0..9: right-side view legs 0-2, antenna 3, stripes 4; the other side 5-9."""
    side, j = divmod(joint, 19)
    return 5 * side + (j // 5 if j < 15 else 3 if j == 15 else 4)


_REDS = [(186, 30, 49), (201, 86, 79), (213, 133, 121)]
_BLUES = [(15, 115, 153), (26, 141, 175), (117, 190, 203)]
_GREY = (210, 210, 210)
LIMB_COLORS = _REDS + [_GREY, _GREY] + _BLUES + [_GREY, _GREY]
