"""-m gpu: f4, the pose videos (deepfly3d_amd/video.py, csrc/render.hip) -- reference df3d/video.py:21-108, df3d/cli.py:308-321.

The frames are drawn on the device; oracle/render.py restates the drawing rule in numpy and must agree bit for bit.  The encoder side
(ffmpeg when present, else Motion-JPEG in AVI) is exercised through the real `df3d-cli --video-2d --video-3d` run on the reference's
sample images."""
import os
import pickle
import shutil

import numpy as np
import pytest
import torch

from oracle import render as orr

pytestmark = pytest.mark.gpu


def _golden_frame(golden_dir, img_id=0):
    from deepfly3d_amd import jpeg, video

    blobs = [open(os.path.join(golden_dir, "images", f"camera_{c}_img_{img_id}.jpg"), "rb").read() for c in video.GRID_CAMERAS]
    g2 = np.load(os.path.join(golden_dir, "golden_2d.npz"))
    pts = np.stack([g2["points2d"][c, img_id] * np.array([480.0, 960.0]) for c in video.GRID_CAMERAS])   # (6, 38, 2) pixels (row, col)
    return blobs, pts


def test_grid_frame_equals_the_numpy_rasteriser(native_lib, cuda, golden_dir):
    from deepfly3d_amd import jpeg, video

    blobs, pts = _golden_frame(golden_dir)
    luma = jpeg.decode_luma(blobs, 960, 480, device=cuda)
    r = video.FrameRenderer(480, 960, 38, cuda)
    got = r.grid2d(luma, torch.from_numpy(pts).to(cuda)).cpu().numpy()
    ref = orr.pose2d_grid(luma.cpu().numpy(), pts, r.bones, r.rgb, video.JOINT_RADIUS, video.BONE_WIDTH)
    assert got.shape == (960, 2880, 3) and np.array_equal(got, ref)
    drawn = (got != np.repeat(np.concatenate([np.concatenate(list(luma.cpu().numpy()[:3]), 1), np.concatenate(list(luma.cpu().numpy()[3:]), 1)], 0)[:, :, None], 3, 2)).any(2)
    assert 5000 < drawn.sum() < 400000   # a skeleton was drawn, not a blank frame and not a filled one
    # unseen joints (a 0 coordinate) and their bones are skipped; a degenerate bone (both ends on one pixel) is a disc of half the line width
    pts2 = pts.copy()
    pts2[:, 5:, :] = 0.0
    pts2[0, 1] = pts2[0, 0]
    got2 = r.grid2d(luma, torch.from_numpy(pts2).to(cuda)).cpu().numpy()
    assert np.array_equal(got2, orr.pose2d_grid(luma.cpu().numpy(), pts2, r.bones, r.rgb, video.JOINT_RADIUS, video.BONE_WIDTH))


def test_pose3d_panels_and_resize_equal_the_restatement(native_lib, cuda, golden_dir):
    from deepfly3d_amd import video
    from deepfly3d_amd.procrustes import video_pose

    g3 = np.load(os.path.join(golden_dir, "golden_3d.npz"))
    pose = video.merge_stripes(video_pose(g3["points3d_wo_procrustes"], device=cuda))
    assert np.array_equal(pose[:, 16], pose[:, 35]) and pose.shape == (15, 38, 3)
    r = video.FrameRenderer(480, 960, 38, cuda)
    for t in (0, 7):
        got = r.panels3d(torch.from_numpy(pose[t]).to(cuda)).cpu().numpy()
        ref = orr.pose3d_panels(pose[t], r.bones, r.rgb, video.panel_azimuths(), video.PANEL_ELEV, video.PANEL_LIM, video.PANEL_SIZE, video.PANEL_LINE_WIDTH)
        assert got.shape == (200, 600, 3) and np.array_equal(got, ref)
        assert all(got[:, k * 200:(k + 1) * 200].any() for k in range(3))   # every view shows the fly
    assert video.panel_azimuths() == [120.0, 165.0, 210.0] and video.panel_azimuths((0, 1, 2)) == [-60.0, -30.0, 0.0]   # reference plot_util.py:48-51
    img = torch.randint(0, 256, (960, 2880, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    frame = torch.zeros((400, 600, 3), dtype=torch.uint8, device=cuda)
    r.resize(img.to(cuda), frame[:200])
    assert np.array_equal(frame[:200].cpu().numpy(), orr.resize_rgb(img.numpy(), 200, 600)) and not frame[200:].any()


def test_cli_writes_both_videos(native_lib, cuda, tmp_path, golden_dir, monkeypatch):
    """`df3d-cli INPUT --video-2d --video-3d` (reference cli.py:305-321): one file per video in the output folder, named like the
    reference's, one frame per image, the reference's layouts (2 x 3 grid of the camera images; 600 x 400 with the 3-D row)."""
    from deepfly3d_amd import cli, video
    from deepfly3d_amd.config import config

    config.pop("image_shape", None)
    monkeypatch.setenv("DF3D_SYNTHETIC_WEIGHTS", "0")
    folder = str(tmp_path / "images")
    os.makedirs(folder)
    for f in os.listdir(os.path.join(golden_dir, "images")):
        shutil.copy(os.path.join(golden_dir, "images", f), folder)
    assert cli.main([folder, "--batch-size", "7", "-n", "2", "--video-2d", "--video-3d", "--output-fps", "12"]) == 0
    out_dir = folder + "_df3d"
    flat = folder.replace("/", "_")
    ext = ".mp4" if shutil.which("ffmpeg") else ".avi"
    v2, v3 = (os.path.join(out_dir, f"video_{k}_{flat}{ext}") for k in ("pose2d", "pose3d"))
    assert os.path.getsize(v2) > 10000 and os.path.getsize(v3) > 2000
    if ext == ".avi":
        f2, f3 = video.read_mjpeg_avi(v2), video.read_mjpeg_avi(v3)
        assert len(f2) == 2 and f2[0].shape == (960, 2880, 3) and len(f3) == 2 and f3[0].shape == (400, 600, 3)
        # the first 2-D frame is the device frame, up to JPEG loss: rebuild it from the saved result
        with open(os.path.join(out_dir, [f for f in os.listdir(out_dir) if f.startswith("df3d_result")][0]), "rb") as fh:
            d = pickle.load(fh)
        from deepfly3d_amd import jpeg

        blobs = [open(os.path.join(folder, f"camera_{c}_img_0.jpg"), "rb").read() for c in video.GRID_CAMERAS]
        luma = jpeg.decode_luma(blobs, 960, 480, device=cuda)
        pts = np.stack([d["points2d"][c, 0] * np.array([480.0, 960.0]) for c in video.GRID_CAMERAS])
        ref = video.FrameRenderer(480, 960, 38, cuda).grid2d(luma, torch.from_numpy(pts).to(cuda)).cpu().numpy()
        assert np.abs(f2[0].astype(np.int32) - ref.astype(np.int32)).mean() < 3.0
        assert f3[0][200:].any() and f3[0][:200].any()
    config.pop("image_shape", None)
