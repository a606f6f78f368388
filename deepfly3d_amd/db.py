"""Manual-correction store (reference df3d/db.py:11-83): `pose_corr_<folder>.pkl` in the output folder, created
by `Core.__init__` (reference df3d/core.py:102).  The GUI that fills it is out of scope;
reading, writing and removing corrections and the pixel-scaled view `manual_corrections()` are kept because
`Core.corrected_points2d*` feed them to the triangulation."""
import copy
import glob
import os
import pickle

import numpy as np

from .config import config


class PoseDB:
    def __init__(self, folder, meta=None, num_cameras=None):
        self.folder = folder
        self.num_cameras = config["num_cameras"] if num_cameras is None else num_cameras
        self.last_write_image_id = 0
        found = glob.glob(os.path.join(folder, "pose_corr*.pkl"))
        if found:
            self.db_path = found[0]
            with open(self.db_path, "rb") as f:
                self.db = pickle.load(f)
        else:
            self.db_path = os.path.join(folder, "pose_corr_{}.pkl".format(folder.replace("/", "-")))
            self.db = {i: dict() for i in range(self.num_cameras)}
            self.db["folder"] = folder
            self.db["meta"] = meta
            self.db["train"] = {i: dict() for i in range(self.num_cameras)}
            self.db["modified"] = {i: dict() for i in range(self.num_cameras)}
            self.dump()

    def read(self, cam_id, img_id):
        return np.array(self.db[cam_id][img_id]) if img_id in self.db[cam_id] else None

    def read_modified_joints(self, cam_id, img_id):
        return self.db["modified"][cam_id].get(img_id, [])

    def write(self, pts, cam_id, img_id, train, modified_joints):
        assert pts.shape[0] == config["num_joints"] and pts.shape[1] == 2
        assert modified_joints is not None
        self.db[cam_id][img_id] = pts
        self.db["train"][cam_id][img_id] = train
        self.db["modified"][cam_id][img_id] = modified_joints
        self.last_write_image_id = img_id

    def remove_corrections(self, cam_id, img_id):
        for table in (self.db, self.db["train"], self.db["modified"]):
            if img_id in table.get(cam_id, {}):
                del table[cam_id][img_id]

    def dump(self):
        with open(self.db_path, "wb") as f:
            pickle.dump(self.db, f)

    def has_key(self, cam_id, img_id):
        return img_id in self.db[cam_id]

    def manual_corrections(self):
        """{cam_id: {img_id: corrected joints scaled by config['image_shape']}} -- a deep copy, like the reference."""
        mc = copy.deepcopy({cam_id: self.db[cam_id] for cam_id in range(self.num_cameras)})
        for cam_id in range(self.num_cameras):
            for img_id in mc[cam_id]:
                mc[cam_id][img_id] = np.array(mc[cam_id][img_id]) * config["image_shape"]
        return mc
