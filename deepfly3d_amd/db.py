"""Store of manually corrected 2-D poses, file-compatible with the reference's `pose_corr*.pkl`
(reference df3d/db.py; created as a side effect of `Core.__init__`, reference df3d/core.py:102).

The pickle is one dict:  {cam_id: {img_id: joints[38, 2] normalised}, ..., "folder", "meta",
"train": {cam_id: {img_id: bool}}, "modified": {cam_id: {img_id: [joint ids]}}}.  The GUI that edits it is out of
scope; `Core.corrected_points2d*` read it so that stored corrections reach triangulation.
"""
import copy
import pickle
from pathlib import Path

import numpy as np

from .config import config

_SIDE_TABLES = ("train", "modified")


class PoseDB:
    def __init__(self, folder, meta=None, num_cameras=None):
        self.folder = folder
        self.num_cameras = int(num_cameras or config["num_cameras"])
        self.last_write_image_id = 0
        existing = sorted(Path(folder).glob("pose_corr*.pkl"))
        if existing:
            self.db_path = str(existing[0])
            self.db = pickle.loads(existing[0].read_bytes())
            return
        self.db_path = str(Path(folder) / f"pose_corr_{str(folder).replace('/', '-')}.pkl")
        cams = range(self.num_cameras)
        self.db = {cam: {} for cam in cams}
        self.db.update(folder=folder, meta=meta, **{name: {cam: {} for cam in cams} for name in _SIDE_TABLES})
        self.dump()

    # -- persistence -------------------------------------------------------------------------------------
    def dump(self):
        Path(self.db_path).write_bytes(pickle.dumps(self.db))

    # -- single entries ------------------------------------------------------------------------------------
    def has_key(self, cam_id, img_id):
        return img_id in self.db[cam_id]

    def read(self, cam_id, img_id):
        entry = self.db[cam_id].get(img_id)
        return None if entry is None else np.array(entry)

    def read_modified_joints(self, cam_id, img_id):
        return self.db["modified"][cam_id].get(img_id, [])

    def write(self, pts, cam_id, img_id, train, modified_joints):
        if pts.shape != (config["num_joints"], 2):
            raise AssertionError(f"a corrected pose must be [{config['num_joints']}, 2], got {pts.shape}")
        if modified_joints is None:
            raise AssertionError("modified_joints must be a list")
        self.db[cam_id][img_id] = pts
        for name, value in zip(_SIDE_TABLES, (train, modified_joints)):
            self.db[name][cam_id][img_id] = value
        self.last_write_image_id = img_id

    def remove_corrections(self, cam_id, img_id):
        self.db.get(cam_id, {}).pop(img_id, None)
        for name in _SIDE_TABLES:
            self.db[name].get(cam_id, {}).pop(img_id, None)

    # -- bulk view -------------------------------------------------------------------------------------------
    def manual_corrections(self):
        """{cam_id: {img_id: joints * config['image_shape']}}: an independent copy in the reference's pixel scaling."""
        scale = config["image_shape"]
        return {cam: {img: np.array(pose) * scale for img, pose in copy.deepcopy(self.db[cam]).items()} for cam in range(self.num_cameras)}
