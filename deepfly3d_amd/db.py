"""Manual-correction store.  Only the side effect `Core.__init__` has in the reference is kept: creating
`pose_corr_<folder>.pkl` in the output folder (reference df3d/core.py:102, df3d/db.py:12-31).  The GUI's
correction workflow is out of scope."""
import glob
import os
import pickle


class PoseDB:
    def __init__(self, folder, meta=None, num_cameras=7):
        self.folder = folder
        found = glob.glob(os.path.join(folder, "pose_corr*.pkl"))
        if found:
            self.db_path = found[0]
            with open(self.db_path, "rb") as f:
                self.db = pickle.load(f)
        else:
            self.db_path = os.path.join(folder, "pose_corr_{}.pkl".format(folder.replace("/", "-")))
            self.db = {i: dict() for i in range(num_cameras)}
            self.db["folder"] = folder
            self.db["meta"] = meta
            self.db["train"] = {i: dict() for i in range(num_cameras)}
            self.db["modified"] = {i: dict() for i in range(num_cameras)}
            self.dump()

    def dump(self):
        with open(self.db_path, "wb") as f:
            pickle.dump(self.db, f)

    def manual_corrections(self):
        return {c: dict(self.db.get(c, {})) for c in range(7)}
