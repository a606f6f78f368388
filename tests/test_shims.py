"""The drop-in seam with the UNMODIFIED reference package: with shims/ on the path, `df3d/core.py`'s two imports
(`df2d.inference.inference_folder`, `pyba.CameraNetwork.CameraNetwork`) resolve to this back-end.  Runs only where the
reference checkout exists (the build container); host-only paths are exercised here, the device paths in -m gpu."""
import os
import pickle
import sys

import numpy as np
import pytest

REF = os.environ.get("DF3D_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "df3d")), reason="reference checkout not present")


@pytest.fixture()
def reference_core(monkeypatch):
    for p in (REF, ROOT, os.path.join(ROOT, "shims")):
        monkeypatch.syspath_prepend(p)
    for name in [m for m in sys.modules if m == "df3d" or m.startswith("df3d.") or m.split(".")[0] in ("df2d", "pyba")]:
        monkeypatch.delitem(sys.modules, name)
    import df3d.core as rc

    yield rc
    for name in [m for m in sys.modules if m == "df3d" or m.startswith("df3d.") or m.split(".")[0] in ("df2d", "pyba")]:
        sys.modules.pop(name, None)


def test_reference_core_binds_to_this_backend(reference_core):
    rc = reference_core
    assert rc.inference_folder.__module__ == "deepfly3d_amd.inference"
    assert rc.CameraNetwork.__module__ == "deepfly3d_amd.camera_network"
    from pyba.config import df3d_bones, df3d_colors

    assert len(df3d_colors) == 38 and max(max(b) for b in df3d_bones) == 37


def test_reference_core_resume_path_on_our_camera_network(reference_core, tmp_path, golden_dir):
    """reference df3d/core.py:109-126: an existing result pickle is re-opened through `CameraNetwork(points2d_px,
    calib=result_dict, image_path=..., colors=..., bones=...)` -- here ours; then the reference's own accessors work."""
    rc = reference_core
    folder = tmp_path / "working"
    folder.mkdir()
    for f in os.listdir(os.path.join(golden_dir, "images")):
        os.symlink(os.path.join(golden_dir, "images", f), folder / f)
    g3 = np.load(f"{golden_dir}/golden_3d.npz")
    result = {c: {"R": g3["R"][c], "tvec": g3["tvec"][c], "distort": g3["distort"][c], "intr": g3["intr"][c]} for c in range(7)}
    result.update(points3d=g3["points3d"], points2d=g3["points2d"], points3d_wo_procrustes=g3["points3d_wo_procrustes"],
                  camera_ordering=g3["camera_ordering"], heatmap_confidence=g3["heatmap_confidence"])
    out = tmp_path / "working_df3d"
    out.mkdir()
    name = "df3d_result_{}.pkl".format(str(folder).replace("/", "_"))
    with open(out / name, "wb") as f:
        pickle.dump(result, f)
    core = rc.Core(str(folder), str(out), 0, [0, 1, 2, 3, 4, 5, 6])
    assert type(core.camNet).__module__ == "deepfly3d_amd.camera_network" and core.has_calibration
    assert np.array_equal(core.points2d, g3["points2d"]) and core.num_images == 2
    px = core.corrected_points2d(3, 1)  # reference helper on top of our Camera.__getitem__
    assert np.allclose(px, g3["points2d"][3, 1] * np.array([480.0, 960.0]))
    assert core.get_image(2, 0).shape[:2] == (480, 960)
