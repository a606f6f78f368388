"""-m gpu end-to-end tests through the reference-shaped surface (Core / CameraNetwork / pipeline / inference)."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import geometry as og
from oracle import hourglass_torch as oh

pytestmark = pytest.mark.gpu


def _sample_folder(tmp_path, golden_dir):
    src = os.path.join(golden_dir, "images")
    folder = tmp_path / "working"
    folder.mkdir()
    for f in os.listdir(src):
        os.symlink(os.path.join(src, f), folder / f)
    return str(folder)


def test_calibration_like_the_reference_test(native_lib, cuda, tmp_path, golden_dir):
    """Mirror of reference tests/test_df3d.py:198-244 (test_calibration): golden 2-D injected -> calibrate_calc ->
    save -> pickle equals the golden 3-D result (points atol 1e-5, cameras atol 1e-4, ordering bit-exact, schema)."""
    from deepfly3d_amd.config import config
    from deepfly3d_amd.core import Core

    config.pop("image_shape", None)
    g2, g3 = np.load(f"{golden_dir}/golden_2d.npz"), np.load(f"{golden_dir}/golden_3d.npz")
    core = Core(_sample_folder(tmp_path, golden_dir), str(tmp_path / "working_df3d"), num_images_max=0, camera_ordering=[0, 1, 2, 3, 4, 5, 6])
    core.num_images = 15
    core.points2d = g2["points2d"]
    core.conf = g2["heatmap_confidence"]
    core.calibrate_calc(0, 100)
    core.save()
    with open(core.save_path, "rb") as f:
        saved = pickle.load(f)
    assert [str(k) for k in saved.keys()] == list(g3["key_order"])
    np.testing.assert_allclose(saved["points3d_wo_procrustes"], g3["points3d_wo_procrustes"], atol=1e-5)
    np.testing.assert_allclose(saved["points3d"], g3["points3d"], atol=1e-5)
    for cam in range(7):
        assert list(saved[cam].keys()) == list(g3["cam_key_order"])
        for key in ("R", "tvec", "intr", "distort"):
            np.testing.assert_allclose(saved[cam][key], g3[key][cam], atol=1e-4)
    assert saved["camera_ordering"].dtype == np.int64 and np.array_equal(saved["camera_ordering"], g3["camera_ordering"])
    assert np.array_equal(saved["points2d"], g3["points2d"]) and np.array_equal(saved["heatmap_confidence"], g3["heatmap_confidence"])
    # reprojection error (what calibrate_calc prints): device residual kernel + device reduction == the oracle's value on
    # the same cameras / points, and on the reference's golden result
    net = core.camNet
    R, t, K = net._stack()
    assert abs(net.reprojection_error() - og.reprojection_error(net.points2d, net.points3d, R, t, K)) < 1e-9
    from deepfly3d_amd.bundle_adjust import reprojection_error

    px = g3["points2d"] * np.array([480.0, 960.0])
    want = og.reprojection_error(px, g3["points3d_wo_procrustes"], g3["R"], g3["tvec"], g3["intr"])
    assert abs(reprojection_error(px, g3["points3d_wo_procrustes"], g3["R"], g3["tvec"], g3["intr"], device=cuda) - want) < 1e-9
    # a recording without a single joint seen by two cameras: nan (the reference only prints the number), not an exception
    assert np.isnan(reprojection_error(np.zeros_like(px), g3["points3d_wo_procrustes"], g3["R"], g3["tvec"], g3["intr"], device=cuda))
    # beyond config["ba_max_images"] the adjustment and its reported error use the same evenly strided subset, and say so once
    import logging

    from deepfly3d_amd.camera_network import CameraNetwork

    class _Grab(logging.Handler):
        def __init__(self):
            super().__init__()
            self.msgs = []

        def emit(self, record):
            self.msgs.append(record.getMessage())

    grab = _Grab()
    logging.getLogger("df3d.logger").addHandler(grab)
    old_cap = config.get("ba_max_images")
    config["ba_max_images"] = 8
    try:
        calib = {c: {k: g3[k][c] for k in ("R", "tvec", "intr", "distort")} for c in range(7)}
        big = CameraNetwork(points2d=px, calib=calib)
        big.triangulate()
        sub = CameraNetwork(points2d=np.ascontiguousarray(px[:, ::2]), calib=calib)
        sub.triangulate()
        assert abs(big.reprojection_error() - sub.reprojection_error()) < 1e-12
        big.reprojection_error()
        assert sum("ba_max_images" in m and "RANDOM" in m for m in grab.msgs) == 1
    finally:
        config["ba_max_images"] = old_cap
        logging.getLogger("df3d.logger").removeHandler(grab)
    # Core.get_points3d (reference df3d/core.py:332-343): Procrustes -> median-centre + axis swap -> One-Euro filter, against
    # the chain executed with the reference's own functions on the golden pose (the pose here comes from OUR bundle
    # adjustment, 1.5e-6 mm from the golden one)
    chain = np.load(f"{golden_dir}/pose_chain_golden.npz")
    video_pose = core.get_points3d()
    assert video_pose.shape == (15, 38, 3) and np.abs(video_pose - chain["filtered"]).max() < 1e-4
    # stored corrections reach corrected_points2d (pixel units, like the reference's PoseDB.manual_corrections)
    fix = np.full((38, 2), 0.25)
    core.db.write(fix, 2, 3, True, [0])
    assert np.allclose(core.corrected_points2d(2, 3), fix * np.array([960, 480]))
    assert np.allclose(core.corrected_points2d(2, 4), core.camNet.cam_list[2][4])
    core.save_corrections()
    config.pop("image_shape", None)


def test_resume_skip_pose_estimation_on_device(native_lib, cuda, tmp_path, golden_dir):
    """f3 on the device (reference df3d/core.py:109-126, cli.py:296-303): an existing result pickle -> `Core.__init__`
    resumes on it (poses + cameras) -> `df3d-cli --skip-pose-estimation --video-3d` = calibrate_calc -> save, all
    computing on the GPU; the file it writes equals the reference's golden 3-D result at the reference's bars."""
    from deepfly3d_amd import cli
    from deepfly3d_amd.config import config
    from deepfly3d_amd.core import Core

    config.pop("image_shape", None)
    g2, g3 = np.load(f"{golden_dir}/golden_2d.npz"), np.load(f"{golden_dir}/golden_3d.npz")
    folder = _sample_folder(tmp_path, golden_dir)
    out_dir = folder + "_df3d"
    os.makedirs(out_dir)
    # (a) resume from the reference's golden 3-D result: the cameras and poses of the pickle are taken over
    flat = os.path.abspath(folder).replace("/", "_")
    pkl = os.path.join(out_dir, f"df3d_result_{flat}.pkl")
    golden = {c: {"R": g3["R"][c], "tvec": g3["tvec"][c], "distort": g3["distort"][c], "intr": g3["intr"][c]} for c in range(7)}
    golden.update(points3d=g3["points3d"], points2d=g3["points2d"], points3d_wo_procrustes=g3["points3d_wo_procrustes"],
                  camera_ordering=g3["camera_ordering"], heatmap_confidence=g3["heatmap_confidence"])
    with open(pkl, "wb") as f:
        pickle.dump(golden, f)
    core = Core(folder, out_dir, num_images_max=0, camera_ordering=[0, 1, 2, 3, 4, 5, 6])
    assert core.save_path == pkl and core.camNet is not None and core.has_calibration
    assert np.array_equal(core.points2d, g3["points2d"]) and np.array_equal(core.points3d, g3["points3d"])
    assert np.allclose(core.camNet.points2d, g3["points2d"] * np.array([480.0, 960.0]))
    core.save()  # triangulates with the resumed cameras on the device
    with open(pkl, "rb") as f:
        again = pickle.load(f)
    np.testing.assert_allclose(again["points3d_wo_procrustes"], g3["points3d_wo_procrustes"], atol=1e-8)
    np.testing.assert_allclose(again["points3d"], g3["points3d"], atol=1e-8)
    # (b) resume from a 2-D-only result and run the CLI with --skip-pose-estimation: calibrate_calc -> save
    with open(pkl, "wb") as f:
        pickle.dump({"points2d": g2["points2d"], "camera_ordering": g2["camera_ordering"], "heatmap_confidence": g2["heatmap_confidence"]}, f)
    assert cli.main([folder, "--skip-pose-estimation", "--video-3d", "--order", "0", "1", "2", "3", "4", "5", "6"]) == 0
    with open(pkl, "rb") as f:
        saved = pickle.load(f)
    assert [str(k) for k in saved.keys()] == list(g3["key_order"])
    np.testing.assert_allclose(saved["points3d_wo_procrustes"], g3["points3d_wo_procrustes"], atol=1e-5)
    np.testing.assert_allclose(saved["points3d"], g3["points3d"], atol=1e-5)
    for cam in range(7):
        for key in ("R", "tvec", "intr", "distort"):
            np.testing.assert_allclose(saved[cam][key], g3[key][cam], atol=1e-4)
    assert np.array_equal(saved["points2d"], g3["points2d"]) and np.array_equal(saved["heatmap_confidence"], g3["heatmap_confidence"])
    assert saved["camera_ordering"].dtype == np.int64 and np.array_equal(saved["camera_ordering"], g3["camera_ordering"])
    config.pop("image_shape", None)


def test_pipeline_matches_oracle_stage_by_stage(native_lib, cuda, golden_dir):
    """Device pipeline (hourglass -> arg-max -> layout -> DLT) on 3 frames against the oracle fed the DEVICE
    heat-maps' arg-max input: index work bit-exact, DLT to 1e-9 relative; heat-maps within the fp32 tolerance."""
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.pipeline import FramePipeline
    from deepfly3d_amd.synthetic import synthetic_state_dict

    sd = synthetic_state_dict(0)
    eng = HourglassEngine(sd, dtype="f32", device=cuda)
    c = np.load(f"{golden_dir}/calib.npz")
    pipe = FramePipeline(eng, c["R"], c["tvec"], c["intr"], camera_ordering=[6, 5, 4, 3, 2, 1, 0])
    frames = torch.rand((3, 7, 256, 512, 3), generator=torch.Generator().manual_seed(1))
    p2, conf, p3 = pipe.run(frames.to(cuda), frames_per_batch=2)
    hm = eng.forward(frames.reshape(21, 256, 512, 3).to(cuda)).cpu().numpy()
    pts, cf = og.heatmap_argmax(hm)
    pts = pts.reshape(3, 7, 19, 2).transpose(1, 0, 2, 3)
    p38 = og.relayout_19_to_38(pts, [6, 5, 4, 3, 2, 1, 0])
    assert np.array_equal(p2.cpu().numpy(), p38)
    assert np.array_equal(conf.cpu().numpy(), cf.reshape(3, 7, 19).transpose(1, 0, 2))
    ref = og.triangulate_dlt(og.pixels_from_normalised(p38, [960, 480]), og.projection_matrices(c["R"], c["tvec"], c["intr"]))
    got = p3.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    net = oh.HourglassNet()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net.eval()
    hm_ref = oh.forward_nhwc(net, frames[0]).numpy()
    assert np.abs(hm[:7] - hm_ref).max() < 2e-4 * np.abs(hm_ref).max()


def test_inference_folder_plumbing(native_lib, cuda, tmp_path, golden_dir, monkeypatch):
    """df2d-shaped entry point on the reference's sample jpgs with synthetic weights: shapes, dtypes, value grid,
    flip handling and batch-size independence (numerical parity with df2d is unpinned: weights absent)."""
    from deepfly3d_amd import inference

    monkeypatch.setenv("DF3D_SYNTHETIC_WEIGHTS", "0")
    folder = _sample_folder(tmp_path, golden_dir)
    pts, conf = inference.inference_folder(folder=folder, camera_ids_to_flip=[4, 5, 6], return_heatmap=False, return_confidence=True,
                                           max_img_id=1, batch_size=8, disable_pin_memory=False)
    assert pts.shape == (7, 2, 19, 2) and conf.shape == (7, 2, 19, 1) and pts.dtype == np.float32 and conf.dtype == np.float32
    grid = pts.astype(np.float64) * np.array([64.0, 128.0])
    assert np.array_equal(grid, np.round(grid)) and grid[..., 0].max() < 64 and grid[..., 1].max() < 128
    # the same frames decoded by libjpeg (Pillow) on the host and pushed through the same device stages: identical
    # results, i.e. the device JPEG decode is bit-exact inside the real pipeline
    from PIL import Image

    from deepfly3d_amd import ops

    frames = np.stack([np.asarray(Image.open(os.path.join(folder, f"camera_{c}_img_{t}.jpg")).convert("L")) for c in range(7) for t in range(2)])
    flip = torch.tensor([1 if c in (4, 5, 6) else 0 for c in range(7) for _ in range(2)], dtype=torch.uint8)
    x = inference.preprocess_u8(torch.from_numpy(frames).to(cuda), flip.to(cuda), (256, 512))
    p_ref, c_ref = ops.heatmap_argmax(inference.get_engine().forward(x))
    assert np.array_equal(p_ref.cpu().numpy().reshape(7, 2, 19, 2), pts) and np.array_equal(c_ref.cpu().numpy().reshape(7, 2, 19, 1), conf)
    pts2, conf2, hm = None, None, None
    out = inference.inference_folder(folder=folder, camera_ids_to_flip=[4, 5, 6], return_heatmap=True, return_confidence=True,
                                     max_img_id=1, batch_size=3, disable_pin_memory=True)
    pts2, hm, conf2 = out
    assert np.array_equal(pts, pts2) and np.array_equal(conf, conf2) and hm.shape == (7, 2, 19, 64, 128)
    # flipping a camera changes its detections, not the others'
    pts3, _ = inference.inference_folder(folder=folder, camera_ids_to_flip=[5, 6], max_img_id=1, batch_size=8)
    assert np.array_equal(pts3[[0, 1, 2, 3, 5, 6]], pts[[0, 1, 2, 3, 5, 6]]) and not np.array_equal(pts3[4], pts[4])


def test_preprocess_kernel_against_torch(native_lib, cuda):
    """Front-end kernel vs a torch restatement: bilinear (half-pixel centres, no antialias) + flip + normalise."""
    from deepfly3d_amd import inference

    img = torch.randint(0, 256, (3, 480, 960), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    flip = torch.tensor([0, 1, 0], dtype=torch.uint8)
    out = inference.preprocess_u8(img.to(cuda), flip.to(cuda)).cpu()
    x = img.float() / 255.0
    x[1] = x[1].flip(-1)
    ref = torch.nn.functional.interpolate(x[:, None], size=(256, 512), mode="bilinear", align_corners=False, antialias=False)[:, 0]
    ref = (ref - 0.22)[..., None].expand(-1, -1, -1, 3)
    assert out.shape == (3, 256, 512, 3)
    assert (out - ref).abs().max() < 1e-5


@pytest.mark.parametrize("rule", ["bilinear", "bilinear_align_corners", "area"])
def test_every_resize_rule_against_its_restatement(native_lib, cuda, rule):
    """df2d's resize rule is data (inference.PREPROCESS["resize"]): each candidate against oracle/preprocess.py (torch
    interpolate for the two bilinear forms, overlap-weight matrices in float64 for cv2.INTER_AREA's definition), grey and
    colour frames, flipped and not, the camera size and an odd one; and the network fed with the frames themselves
    (df3d_hg_forward_u8) equals preprocess + forward bit for bit under every rule."""
    from deepfly3d_amd import inference
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.synthetic import synthetic_state_dict
    from oracle import preprocess as opre

    g = torch.Generator().manual_seed(17)
    mean, std = (0.22, 0.31, 0.18), (0.9, 1.1, 1.3)
    saved = dict(inference.PREPROCESS)
    inference.PREPROCESS.update(mean=mean, std=std, resize=rule)
    try:
        eng = HourglassEngine(synthetic_state_dict(3), dtype="f32", device=cuda)
        for shape in ((3, 480, 960), (2, 301, 517, 3), (1, 256, 512)):
            frames = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
            flip = torch.tensor([0, 1, 1][: shape[0]], dtype=torch.uint8)
            got = inference.preprocess_u8(frames.to(cuda), flip.to(cuda), (256, 512)).cpu()
            ref = opre.preprocess_u8(frames, flip, (256, 512), mean, std, rule)
            err = float((got - ref).abs().max())
            assert err < 2e-6, (rule, shape, err)   # values are O(1): float32 rounding of a handful of operations
            x = inference.preprocess_u8(frames.to(cuda), flip.to(cuda), (256, 512))
            assert torch.equal(eng.forward(x), eng.forward_u8(frames.to(cuda), flip.to(cuda), mean, std, resize=rule)), (rule, shape)
        if rule != "bilinear":   # the rules really differ on a down-scale
            frames = torch.randint(0, 256, (1, 480, 960), dtype=torch.uint8, generator=g).to(cuda)
            a = inference.preprocess_u8(frames, None)
            inference.PREPROCESS.update(resize="bilinear")
            assert not torch.equal(a, inference.preprocess_u8(frames, None))
    finally:
        inference.PREPROCESS.clear()
        inference.PREPROCESS.update(saved)
    with pytest.raises(KeyError):
        eng.forward_u8(torch.zeros((1, 8, 8), dtype=torch.uint8, device=cuda), resize="lanczos")


def test_full_cli_run_on_sample_images(native_lib, cuda, tmp_path, golden_dir, monkeypatch):
    """df3d-cli end to end on the sample jpgs (synthetic weights): one result file with the reference's schema."""
    from deepfly3d_amd import cli
    from deepfly3d_amd.config import config

    config.pop("image_shape", None)
    monkeypatch.setenv("DF3D_SYNTHETIC_WEIGHTS", "0")
    folder = _sample_folder(tmp_path, golden_dir)
    # random-weight detections are not geometrically consistent; the run must still complete and save
    assert cli.main([folder, "--batch-size", "7", "-n", "2"]) == 0
    out_dir = folder + "_df3d"
    files = [f for f in os.listdir(out_dir) if f.startswith("df3d_result")]
    assert len(files) == 1
    with open(os.path.join(out_dir, files[0]), "rb") as f:
        d = pickle.load(f)
    assert [str(k) for k in d.keys()] == ["0", "1", "2", "3", "4", "5", "6", "points3d", "points2d", "points3d_wo_procrustes", "camera_ordering", "heatmap_confidence"]
    assert d["points2d"].shape == (7, 2, 38, 2) and d["points3d"].shape == (2, 38, 3) and d["heatmap_confidence"].shape == (7, 2, 19, 1)
    config.pop("image_shape", None)


def test_frame_sharding_is_bit_consistent(native_lib, cuda, golden_dir):
    """Shard-consistency on one GPU: processing a sequence in one piece, in different batch sizes, or as two
    contiguous shards (what two ranks would do before the gather) gives bit-identical results."""
    from deepfly3d_amd import distributed as dd
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.pipeline import FramePipeline
    from deepfly3d_amd.synthetic import synthetic_state_dict

    eng = HourglassEngine(synthetic_state_dict(0), dtype="f32", device=cuda)
    c = np.load(f"{golden_dir}/calib.npz")
    pipe = FramePipeline(eng, c["R"], c["tvec"], c["intr"])
    frames = torch.rand((6, 7, 256, 512, 3), generator=torch.Generator().manual_seed(4)).to(cuda)
    whole = [t.clone() for t in pipe.run(frames, frames_per_batch=6)]
    other = [t.clone() for t in pipe.run(frames, frames_per_batch=4)]
    for a, b in zip(whole, other):
        assert torch.equal(a, b)
    (a0, a1), (b0, b1) = dd.all_ranges(6, 2)
    s0 = [t.clone() for t in pipe.run(frames[a0:a1], frames_per_batch=8)]
    s1 = [t.clone() for t in pipe.run(frames[b0:b1], frames_per_batch=8)]
    assert torch.equal(torch.cat([s0[0], s1[0]], dim=1), whole[0])
    assert torch.equal(torch.cat([s0[1], s1[1]], dim=1), whole[1])
    assert torch.equal(torch.cat([s0[2], s1[2]], dim=0), whole[2])


def test_cli_two_ranks_match_one_rank(native_lib, cuda, tmp_path, golden_dir):
    """`torch.distributed.run --nproc-per-node 2 -m deepfly3d_amd.cli` (two ranks sharing this GPU, gloo for the
    gather) writes the same result file as the single-process CLI: frames are sharded, rank 0 gathers and saves."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DF3D_SYNTHETIC_WEIGHTS="0", DF3D_DIST_BACKEND="gloo", PYTHONPATH=root)
    results = []
    for tag, launcher in (("one", [sys.executable, "-m", "deepfly3d_amd.cli"]),
                          ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                   "--master-port", "29631", "-m", "deepfly3d_amd.cli"])):
        base = tmp_path / tag
        base.mkdir()
        folder = _sample_folder(base, golden_dir)
        r = subprocess.run(launcher + [folder, "-n", "2", "-vv"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        files = [f for f in os.listdir(folder + "_df3d") if f.startswith("df3d_result")]
        assert len(files) == 1
        with open(os.path.join(folder + "_df3d", files[0]), "rb") as f:
            results.append(pickle.load(f))
        if tag == "two":
            # the 3-D stage is sharded too: with the cameras fixed (after the bundle adjustment on rank 0) BOTH ranks triangulate
            # their own frame on their device, one more gather carries points3d to rank 0
            log = r.stdout + r.stderr
            assert "rank 0 of 2: triangulated frames [0, 1)" in log and "rank 1 of 2: triangulated frames [1, 2)" in log, log[-3000:]
            # resume on two ranks (--skip-pose-estimation): every rank re-opens the result, the DLT is sharded again, same file
            r2 = subprocess.run(launcher + [folder, "-n", "2", "-vv", "--skip-pose-estimation", "--video-3d"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
            assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
            assert "rank 1 of 2: triangulated frames [1, 2)" in r2.stdout + r2.stderr
            with open(os.path.join(folder + "_df3d", files[0]), "rb") as f:
                again = pickle.load(f)
            for k in ("points2d", "heatmap_confidence", "camera_ordering"):
                assert np.array_equal(again[k], results[-1][k]), k
            assert np.allclose(again["points3d_wo_procrustes"], results[-1]["points3d_wo_procrustes"], atol=1e-6)
    one, two = results
    assert list(one.keys()) == list(two.keys())
    for k in ("points2d", "heatmap_confidence", "camera_ordering"):
        assert np.array_equal(one[k], two[k]), k
    assert np.allclose(one["points3d_wo_procrustes"], two["points3d_wo_procrustes"], atol=1e-9)


@pytest.mark.parametrize("dtype", ["f32", "f32s", "bf16", "f16"])
def test_full_size_workload_properties(native_lib, cuda, golden_dir, dtype):
    """BASELINE configs[1] / configs[2] at FULL size (1 000 frames x 7 views of 256x512x3, one GPU), checked through
    size-independent properties: the run is deterministic (bit-identical twice), every frame's result is independent
    of its position and batch (16 scattered frames re-run one by one are bit-identical), detections lie on the
    heat-map grid, confidences are finite, and for two frames the whole chain (device heat-maps -> oracle arg-max ->
    oracle layout -> oracle DLT) reproduces the device output."""
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.pipeline import FramePipeline
    from deepfly3d_amd.synthetic import synthetic_state_dict

    T = 1000
    eng = HourglassEngine(synthetic_state_dict(0), dtype=dtype, device=cuda)
    c = np.load(f"{golden_dir}/calib.npz")
    pipe = FramePipeline(eng, c["R"], c["tvec"], c["intr"])
    gen = torch.Generator(device=cuda).manual_seed(7)
    frames = torch.rand((T, 7, 256, 512, 3), generator=gen, device=cuda, dtype=torch.float32)
    first = [t.clone() for t in pipe.run(frames, frames_per_batch=32)]
    again = pipe.run(frames, frames_per_batch=32)
    for a, b in zip(first, again):
        assert torch.equal(a, b)
    p2, conf, p3 = first
    assert p2.shape == (7, T, 38, 2) and conf.shape == (7, T, 19) and p3.shape == (T, 38, 3)
    assert bool(torch.isfinite(conf).all()) and bool(torch.isfinite(p3).all())
    rng = np.random.default_rng(3)
    for t in sorted(rng.choice(T, size=16, replace=False).tolist()):
        q2, qc, q3 = pipe.run(frames[t : t + 1], frames_per_batch=1)
        assert torch.equal(q2[:, 0], p2[:, t]) and torch.equal(qc[:, 0], conf[:, t]) and torch.equal(q3[0], p3[t])
    # grid: un-flipped columns are k/128 or 1 - k/128, rows k/64, all in [0, 1]
    g = p2.cpu().numpy()
    rows, cols = g[..., 0] * 64.0, g[..., 1] * 128.0
    assert np.array_equal(rows, np.round(rows)) and np.array_equal(cols, np.round(cols)) and rows.min() >= 0 and rows.max() < 64 and cols.max() <= 128
    # two frames against the oracle's geometry on the DEVICE heat-maps
    for t in (0, T - 1):
        hm = eng.forward(frames[t].contiguous()).cpu().numpy()
        pts, cf = og.heatmap_argmax(hm)
        p38 = og.relayout_19_to_38(pts.reshape(7, 1, 19, 2), list(range(7)))
        assert np.array_equal(p2[:, t].cpu().numpy(), p38[:, 0]) and np.array_equal(conf[:, t].cpu().numpy(), cf)
        X = og.triangulate_dlt(og.pixels_from_normalised(p38, [960, 480]), og.projection_matrices(c["R"], c["tvec"], c["intr"]))
        assert np.abs(p3[t].cpu().numpy() - X[0]).max() < 1e-6 * max(1.0, np.abs(X).max())


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "f32s"])
def test_forward_from_camera_frames_equals_preprocess_then_forward(native_lib, cuda, dtype):
    """df3d_hg_forward_u8 (the stem samples the uint8 frames itself) is bit for bit df3d_hg_forward(df3d_preprocess_u8(frames)):
    grey and 3-channel frames, flipped and not, a frame size that is not a multiple of the network input, non-trivial mean / std."""
    from deepfly3d_amd import inference
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.synthetic import synthetic_state_dict

    eng = HourglassEngine(synthetic_state_dict(3), dtype=dtype, device=cuda)
    g = torch.Generator().manual_seed(11)
    mean, std = (0.22, 0.31, 0.18), (0.9, 1.1, 1.3)
    saved = dict(inference.PREPROCESS)
    inference.PREPROCESS.update(mean=mean, std=std)
    try:
        for shape in ((3, 480, 960), (2, 301, 517, 3)):
            frames = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).to(cuda)
            flip = torch.tensor([0, 1, 1][: shape[0]], dtype=torch.uint8, device=cuda)
            ref = eng.forward(inference.preprocess_u8(frames, flip, (256, 512)))
            got = eng.forward_u8(frames, flip, mean, std)
            assert torch.equal(ref, got), shape
            assert torch.equal(eng.forward(inference.preprocess_u8(frames, None, (256, 512))), eng.forward_u8(frames, None, mean, std))
            p, c = inference.inference_frames(frames, flip, eng)
            assert p.shape == (shape[0], 19, 2) and c.shape[0] == shape[0]
    finally:
        inference.PREPROCESS.clear()
        inference.PREPROCESS.update(saved)
    with pytest.raises(ValueError):
        eng.forward_u8(torch.zeros((1, 8, 8, 2), dtype=torch.uint8, device=cuda))
