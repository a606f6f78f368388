"""CPU tests of the host-side logic: parameter packing / BN folding, frame sharding, folder discovery, CLI flags,
the correction store, result schema."""
import ctypes
import os
import pickle

import numpy as np
import pytest
import torch


def test_procrustes_template_constants(golden_dir):
    """The only host arithmetic of a9: the constant template reduced to 60 numbers (the transform itself is a
    device computation: tests/test_gpu_pose3d.py)."""
    from deepfly3d_amd.procrustes import template_constants

    tmpl = np.load(f"{golden_dir}/template.npz")["points3d"]
    seg, fit = template_constants()
    assert seg.shape == (2, 12) and fit.shape == (2, 6, 3)
    legs = tmpl[:, 19:34].reshape(15, 3, 5, 3)
    assert np.array_equal(seg[1], np.median(np.linalg.norm(np.diff(legs, axis=2), axis=-1).reshape(15, 12), axis=0))
    assert np.array_equal(fit[0], np.median(tmpl[:, [0, 1, 5, 6, 10, 11]], axis=0))
    seg2, _ = template_constants(tmpl * 2.0)
    assert np.allclose(seg2, 2.0 * seg)


def test_pose_db_round_trip(tmp_path):
    """Correction store (reference df3d/db.py): file name, write / read / remove, pixel-scaled deep copy."""
    from deepfly3d_amd.config import config
    from deepfly3d_amd.db import PoseDB

    folder = str(tmp_path)
    db = PoseDB(folder)
    assert os.path.basename(db.db_path) == "pose_corr_{}.pkl".format(folder.replace("/", "-")) and os.path.exists(db.db_path)
    assert db.read(0, 3) is None and db.read_modified_joints(0, 3) == [] and not db.has_key(0, 3)
    pts = np.full((38, 2), 0.5)
    db.write(pts, 2, 7, True, [4, 5])
    db.dump()
    again = PoseDB(folder)
    assert again.has_key(2, 7) and again.read_modified_joints(2, 7) == [4, 5] and np.array_equal(again.read(2, 7), pts)
    old_shape = config.get("image_shape")
    config["image_shape"] = [960, 480]
    try:
        mc = again.manual_corrections()
        assert np.array_equal(mc[2][7], pts * [960, 480]) and np.array_equal(again.read(2, 7), pts)
    finally:
        if old_shape is None:
            config.pop("image_shape")
        else:
            config["image_shape"] = old_shape
    again.remove_corrections(2, 7)
    assert not again.has_key(2, 7) and 7 not in again.db["train"][2] and 7 not in again.db["modified"][2]


def test_package_data_matches_reference_fixtures(golden_dir):
    from deepfly3d_amd.config import load_calibration, load_procrustes_template

    c = np.load(f"{golden_dir}/calib.npz")
    cal = load_calibration()
    for cam in range(7):
        for k in ("R", "tvec", "intr", "distort"):
            assert np.array_equal(cal[cam][k], c[k][cam])
    assert np.array_equal(load_procrustes_template(), np.load(f"{golden_dir}/template.npz")["points3d"])


def test_bn_folding_and_packing_reproduce_the_oracle_layer(native_lib):
    """pack_state_dict folds bn2 into conv1 etc.; check one bottleneck numerically against torch (CPU, float64)."""
    from deepfly3d_amd import _native
    from deepfly3d_amd.hourglass import pack_state_dict
    from oracle import hourglass_torch as oh

    net = oh.build(seed=1)
    h = ctypes.c_void_p()
    assert native_lib.df3d_hg_create(_native.DF3D_DTYPE_F32, 2, ctypes.byref(h)) == 0
    blob = pack_state_dict(h, net.state_dict())
    d = _native.HGParam()
    params = {}
    for i in range(native_lib.df3d_hg_num_params(h)):
        native_lib.df3d_hg_param_desc(h, i, ctypes.byref(d))
        params[(d.name.decode(), d.kind)] = (blob[d.offset : d.offset + d.count].copy(), d.taps, d.cin, d.cout, d.cin_pad, d.cout_pad)
    native_lib.df3d_hg_destroy(h)
    blk = net.layer3[0].double()
    x = torch.randn(1, 256, 5, 6, dtype=torch.float64)
    with torch.no_grad():
        ref = torch.relu(blk.bn2(blk.conv1(torch.relu(blk.bn1(x)))))
    w, taps, cin, cout, cin_pad, cout_pad = params[("layer3.0.conv1", 0)]
    b = params[("layer3.0.conv1", 1)][0][:cout]
    s = params[("layer3.0.conv1", 2)][0][:cin]
    t = params[("layer3.0.conv1", 3)][0][:cin]
    W = w.reshape(taps, cout_pad, cin_pad)[0, :cout, :cin].astype(np.float64)
    a = np.maximum(x.numpy() * s[None, :, None, None] + t[None, :, None, None], 0)
    got = np.maximum(np.einsum("oc,nchw->nohw", W, a) + b[None, :, None, None], 0)
    assert np.abs(got - ref.numpy()).max() < 1e-5 * np.abs(ref.numpy()).max()
    # stem packing: [148][64], k = ky*21 + kx*3 + c, BN folded
    ws = params[("conv1", 0)][0].reshape(184, 64)[:148]
    sd = net.state_dict()
    scale = (sd["bn1.weight"] / torch.sqrt(sd["bn1.running_var"] + 1e-5)).numpy()
    assert np.allclose(ws[2 * 21 + 3 * 3 + 1, 5], sd["conv1.weight"][5, 1, 2, 3].item() * scale[5], rtol=1e-6)
    assert np.all(ws[147] == 0)
    # padded score_ input channels are zero
    wsc = params[("score_.0", 0)]
    assert wsc[4] == 32 and np.all(wsc[0].reshape(1, 256, 32)[0, :, 19:] == 0)


def test_checkpoint_packing_is_strict(native_lib):
    """A checkpoint the engine does not understand must not load "successfully" (round-3 review): bearpaw's num_blocks > 1
    (`layer1.1.*`, `hg.0.hg.3.0.1.*`), more stacks than the engine, a missing parameter or another width raise
    CheckpointMismatch naming the keys; `num_batches_tracked` counters (torch's BatchNorm book-keeping) are ignored.
    Reference: df3d/config.py:30-39 (the one checkpoint the reference loads, `num_stacks`)."""
    from deepfly3d_amd import _native
    from deepfly3d_amd.hourglass import CheckpointMismatch, describe_state_dict, pack_state_dict
    from deepfly3d_amd.synthetic import synthetic_state_dict

    def handle(stacks):
        h = ctypes.c_void_p()
        assert native_lib.df3d_hg_create(_native.DF3D_DTYPE_F32, stacks, ctypes.byref(h)) == 0
        return h

    h2 = handle(2)
    sd = synthetic_state_dict(0)
    assert describe_state_dict(sd) == {"num_stacks": 2, "num_blocks": 1, "depth": 4, "feats": 128, "num_classes": 19}
    ok = dict(sd)
    ok["bn1.num_batches_tracked"] = np.array(7)
    ok["hg.1.hg.0.2.0.bn3.num_batches_tracked"] = np.array(7)
    assert np.array_equal(pack_state_dict(h2, ok), pack_state_dict(h2, sd, strict=False))
    # num_blocks = 2: a second bottleneck per residual unit
    two = dict(sd)
    for k, v in sd.items():
        if k.startswith("layer1.0.") and "downsample" not in k:
            two["layer1.1." + k[len("layer1.0."):]] = v
        if k.startswith("hg.0.hg.3.0.0."):
            two["hg.0.hg.3.0.1." + k[len("hg.0.hg.3.0.0."):]] = v
    with pytest.raises(CheckpointMismatch) as e:
        pack_state_dict(h2, two)
    assert "num_blocks = 2" in str(e.value) and "layer1.1.bn1.{bias," in str(e.value) and "hg.0.hg.3.0.1." in str(e.value) and "Unconsumed" in str(e.value)
    assert pack_state_dict(h2, two, strict=False).shape == pack_state_dict(h2, sd).shape   # (the old behaviour, on request)
    # a 4-stack checkpoint in a 2-stack engine, and the other way round
    four = synthetic_state_dict(0, num_stacks=4)
    with pytest.raises(CheckpointMismatch) as e:
        pack_state_dict(h2, four)
    assert "4 stacks" in str(e.value) and "built for 2" in str(e.value) and "fc.2.0.weight" in str(e.value) or "hg.2." in str(e.value)
    first_two = pack_state_dict(h2, four, strict=False)   # on request: the first two stacks of the 4-stack checkpoint (left-over keys tolerated)
    assert first_two.shape == pack_state_dict(h2, sd).shape
    with pytest.raises(CheckpointMismatch):
        pack_state_dict(h2, {k: v for k, v in four.items() if k != "bn1.weight"}, strict=False)   # ... a MISSING key never is
    h4 = handle(4)
    assert pack_state_dict(h4, four).size > pack_state_dict(h2, sd).size
    with pytest.raises(CheckpointMismatch) as e:
        pack_state_dict(h4, sd)
    assert "Missing keys" in str(e.value) and "hg.2." in str(e.value)
    # one parameter missing; one with another shape
    less = {k: v for k, v in sd.items() if k != "res.1.0.bn2.running_var"}
    with pytest.raises(CheckpointMismatch) as e:
        pack_state_dict(h2, less)
    assert "Missing keys: res.1.0.bn2.running_var." in str(e.value)
    wide = dict(sd)
    wide["score.1.weight"] = np.zeros((21, 256, 1, 1), np.float32)
    wide["score.1.bias"] = np.zeros(21, np.float32)
    with pytest.raises(CheckpointMismatch) as e:
        pack_state_dict(h2, wide)
    assert "score.1.weight" in str(e.value) and "(21, 256, 1, 1)" in str(e.value)
    native_lib.df3d_hg_destroy(h2)
    native_lib.df3d_hg_destroy(h4)


def test_bench_strong_plan_covers_the_stream_for_every_world_size():
    """bench.py --strong: the shards of the ONE stream (BASELINE configs[3]/[4]: 100 000 frames, window 1 000) tile it exactly for
    1, 2, 4 and 8 ranks, whole windows per rank, and `steps` is the largest shard's batch count."""
    import bench

    for world in (1, 2, 4, 8):
        for stream, align in ((100000, 1000), (100000, 1), (2560, 1000), (7, 1)):
            per_rank, steps = bench.strong_plan(stream, world, align, 128)
            assert len(per_rank) == world and per_rank[0][0] == 0 and per_rank[-1][1] == stream
            assert all(a[1] == b[0] for a, b in zip(per_rank, per_rank[1:]))
            assert all(t0 % align == 0 for t0, t1, _ in per_rank if t1 > t0)   # (an empty shard sits at the stream's end)
            assert steps == max(-(-(t1 - t0) // 128) for t0, t1, _ in per_rank)
    per_rank, steps = bench.strong_plan(100000, 8, 1000, 128)
    assert [t1 - t0 for t0, t1, _ in per_rank] == [13000] * 4 + [12000] * 4 and steps == 102


def test_mjpeg_avi_writer_round_trip(tmp_path):
    """f4's fallback encoder (no ffmpeg in this image): Motion-JPEG in a RIFF/AVI container, header fields consistent with the
    frames written (count, size, rate), every frame recoverable."""
    import struct

    from deepfly3d_amd import video

    rng = np.random.default_rng(0)
    frames = [np.kron(rng.integers(0, 256, (12, 20, 3), dtype=np.uint8), np.ones((10, 10, 1), np.uint8)) for _ in range(5)]   # blocky: JPEG keeps it
    path = str(tmp_path / "clip.avi")
    w = video.MjpegAviWriter(path, 200, 120, 12.5)
    for f in frames:
        w.write(f)
    w.close()
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and struct.unpack_from("<I", data, 4)[0] == len(data) - 8 and data[8:12] == b"AVI "
    avih = data.index(b"avih") + 8
    usec, _, _, _, total, _, streams, _, width, height = struct.unpack_from("<10I", data, avih)
    assert (usec, total, streams, width, height) == (80000, 5, 1, 200, 120)
    strh = data.index(b"strh") + 8
    assert data[strh:strh + 8] == b"vidsMJPG" and struct.unpack_from("<II", data, strh + 20) == (1000, 12500)
    back = video.read_mjpeg_avi(path)
    assert len(back) == 5 and all(b.shape == (120, 200, 3) for b in back)
    assert max(np.abs(b.astype(int) - f.astype(int)).mean() for b, f in zip(back, frames)) < 12.0   # (random colours in 10 x 10 blocks through 4:2:0 JPEG)
    assert video.merge_stripes(np.arange(38 * 3, dtype=float).reshape(1, 38, 3))[0, 17].tolist() == [(51 + 108) / 2, (52 + 109) / 2, (53 + 110) / 2]


def test_synthetic_state_dict_matches_oracle_module_shapes():
    from deepfly3d_amd.synthetic import synthetic_state_dict
    from oracle import hourglass_torch as oh

    sd = synthetic_state_dict(0)
    ref = {k: tuple(v.shape) for k, v in oh.HourglassNet().state_dict().items() if not k.endswith("num_batches_tracked")}
    assert {k: v.shape for k, v in sd.items()} == ref
    again = synthetic_state_dict(0)
    assert all(np.array_equal(sd[k], again[k]) for k in sd)


@pytest.mark.parametrize("T,world,align", [(100000, 8, 1000), (1000, 8, 1), (15, 4, 1), (7, 8, 1), (100000, 3, 1000), (0, 2, 1)])
def test_shard_ranges_cover_exactly(T, world, align):
    from deepfly3d_amd.distributed import all_ranges

    r = all_ranges(T, world, align)
    assert r[0][0] == 0 and r[-1][1] == T
    for (a0, a1), (b0, b1) in zip(r, r[1:]):
        assert a1 == b0 and a0 <= a1
    assert all(a % align == 0 for a, _ in r if a < T)
    sizes = [b - a for a, b in r]
    assert max(sizes) - min(sizes) <= align


def test_os_util_and_core_folder_discovery(tmp_path, golden_dir):
    from deepfly3d_amd import os_util

    src = os.path.join(golden_dir, "images")
    folder = tmp_path / "images"
    folder.mkdir()
    for f in os.listdir(src):
        os.symlink(os.path.join(src, f), folder / f)
    assert os_util.get_max_img_id(str(folder)) == 1
    assert os_util.parse_img_name("camera_3_img_000012.jpg") == (3, 12)
    assert os_util.parse_vid_name("camera_5.mp4") == 5
    with pytest.raises(FileNotFoundError):
        os_util.get_max_img_id(str(tmp_path))
    # Core without a GPU: construction (discovery, image shape, PoseDB side effect, ordering) works on CPU
    from deepfly3d_amd.config import config
    from deepfly3d_amd.core import Core, find_default_camera_ordering

    config.pop("image_shape", None)
    core = Core(str(folder), str(tmp_path / "out"), num_images_max=0, camera_ordering=[0, 1, 2, 3, 4, 5, 6])
    assert core.num_images == 2 and core.image_shape == [960, 480]
    assert np.array_equal(core.camera_ordering, np.arange(7))
    assert core.save_path.endswith("df3d_result_" + str(folder).replace("/", "_") + ".pkl")
    assert any(f.startswith("pose_corr") for f in os.listdir(tmp_path / "out"))
    assert Core(str(folder), str(tmp_path / "out"), num_images_max=1).num_images == 1
    assert list(find_default_camera_ordering("/data/CLC/x")) == [0, 6, 5, 4, 3, 2, 1]
    with pytest.raises(NotImplementedError):
        find_default_camera_ordering("/nowhere")
    config["image_shape"] = [1, 1]
    with pytest.raises(ValueError):
        Core(str(folder), str(tmp_path / "out"))
    config.pop("image_shape", None)


def test_core_save_schema_2d_only_and_resume(tmp_path, golden_dir):
    """2-D-only save (reference core.py:351-365) and the resume path (core.py:109-126) without a GPU."""
    from deepfly3d_amd.config import config
    from deepfly3d_amd.core import Core

    src = os.path.join(golden_dir, "images")
    folder = tmp_path / "images"
    folder.mkdir()
    for f in os.listdir(src):
        os.symlink(os.path.join(src, f), folder / f)
    config.pop("image_shape", None)
    g2 = np.load(f"{golden_dir}/golden_2d.npz")
    core = Core(str(folder), str(tmp_path / "out"))
    core.points2d, core.conf = g2["points2d"][:, :2], g2["heatmap_confidence"][:, :2]
    core.save()
    with open(core.save_path, "rb") as f:
        d = pickle.load(f)
    assert list(d.keys()) == ["points2d", "camera_ordering", "heatmap_confidence"]
    assert d["camera_ordering"].dtype == np.int64
    core2 = Core(str(folder), str(tmp_path / "out"))  # resumes from the pickle
    assert np.array_equal(core2.points2d, d["points2d"]) and core2.camNet is not None
    assert not core2.camNet.has_calibration()
    assert np.array_equal(core2.camNet.points2d, d["points2d"] * np.array([480.0, 960.0]))
    config.pop("image_shape", None)


def test_cli_flags_and_exit_codes(tmp_path, capsys):
    from deepfly3d_amd import cli

    a = cli.parse_cli_args([str(tmp_path / "imgs")])
    assert a.output_folder == str(tmp_path / "imgs_df3d") and a.batch_size == 8 and a.order == list(range(7))
    assert not a.skip_estimation and a.num_images_max == 0 and a.output_fps is None
    a = cli.parse_cli_args([str(tmp_path), "--camera-ids", "6", "5", "4", "3", "2", "1", "0", "-n", "5", "--batch-size", "4", "--pin-memory-disabled", "-x"])
    assert a.order == [6, 5, 4, 3, 2, 1, 0] and a.num_images_max == 5 and a.batch_size == 4 and a.pin_memory_disabled and a.delete_images
    assert cli.main([str(tmp_path), "-d"]) == 0
    assert cli.main([str(tmp_path), "-r", "-f"]) == 1
    assert cli.main([str(tmp_path / "missing.txt"), "-f"]) == 1
    assert cli.main([str(tmp_path), "-f"]) == 1  # a directory is not a list file
    assert cli.main([str(tmp_path), "--skip-pose-estimation"]) == 0  # nothing to do
    (tmp_path / "a" / "images").mkdir(parents=True)
    (tmp_path / "b" / "c" / "images").mkdir(parents=True)
    assert sorted(cli.find_subfolders(str(tmp_path), "images")) == sorted([str(tmp_path / "a" / "images"), str(tmp_path / "b" / "c" / "images")])


def test_camera_network_bookkeeping(golden_dir):
    """CameraNetwork tolerates a whole result dict as `calib`, keeps pixel (row, col) points, summarises in the
    golden key order (the reprojection error is a device computation: tests/test_gpu_core.py)."""
    from deepfly3d_amd.camera_network import CameraNetwork

    g3 = np.load(f"{golden_dir}/golden_3d.npz")
    calib = {c: {"R": g3["R"][c], "tvec": g3["tvec"][c], "intr": g3["intr"][c], "distort": g3["distort"][c]} for c in range(7)}
    calib.update({"points2d": g3["points2d"], "meta": None, np.int64(3): calib[3]})
    net = CameraNetwork(g3["points2d"] * np.array([480.0, 960.0]), calib=calib)
    assert net.has_calibration() and net[2].cam_id == 2 and net.cam_list[5][3].shape == (38, 2)
    net.points3d = g3["points3d_wo_procrustes"]
    s = net.summarize()
    assert list(s.keys()) == [0, 1, 2, 3, 4, 5, 6, "points3d", "points2d"]
    assert list(s[0].keys()) == ["R", "tvec", "distort", "intr"]
    with pytest.raises(NotImplementedError):
        net.bundle_adjust(update_intrinsic=True)
    assert not CameraNetwork(g3["points2d"], calib=None).has_calibration()


def test_assemble_result_schema(golden_dir):
    from deepfly3d_amd.distributed import assemble_result

    g3 = np.load(f"{golden_dir}/golden_3d.npz")
    cams = {k: g3[k] for k in ("R", "tvec", "intr", "distort")}
    from oracle import geometry as og

    tmpl = np.load(f"{golden_dir}/template.npz")["points3d"]
    out = assemble_result(g3["points2d"], g3["heatmap_confidence"][..., 0], g3["points3d_wo_procrustes"], cams, g3["camera_ordering"],
                          procrustes=lambda p: og.procrustes_separate(p, tmpl))  # device Procrustes: tests/test_gpu_pose3d.py
    assert [str(k) for k in out.keys()] == list(g3["key_order"])
    assert np.abs(out["points3d"] - g3["points3d"]).max() < 1e-12
    assert out["heatmap_confidence"].shape == (7, 15, 19, 1) and out["points2d"].dtype == np.float64


def _fake_tool(bin_dir, name, body):
    path = os.path.join(bin_dir, name)
    with open(path, "w") as f:
        f.write("#!/bin/bash\n" + body)
    os.chmod(path, 0o755)


def test_core_with_videos_fps_and_delete_images(tmp_path, golden_dir, monkeypatch):
    """Mirror of reference tests/test_df3d.py: test_load_core_with_videos / test_delete_images, with stand-in ffmpeg /
    ffprobe executables (the image has neither): a folder that only holds camera_x.mp4 is expanded to frames once, the
    frame rate comes from ffprobe's avg_frame_rate, --delete-images removes the frames and keeps the videos."""
    from deepfly3d_amd.config import config
    from deepfly3d_amd.core import Core

    bin_dir = tmp_path / "bin"
    bin_dir.mkdir()
    src = os.path.join(golden_dir, "images")
    log = tmp_path / "ffmpeg_calls.log"
    _fake_tool(str(bin_dir), "ffprobe", 'echo "30000/1001"\n')
    _fake_tool(str(bin_dir), "ffmpeg", f'''
while [ $# -gt 0 ]; do if [ "$1" = "-i" ]; then vid="$2"; fi; last="$1"; shift; done
cam=$(basename "$vid" .mp4); cam=${{cam#camera_}}
echo "$vid" >> {log}
for n in 0 1; do cp {src}/camera_${{cam}}_img_$n.jpg "$(printf "$last" $n)"; done
''')
    monkeypatch.setenv("PATH", f"{bin_dir}:{os.environ['PATH']}")
    config.pop("image_shape", None)
    folder = tmp_path / "working"
    folder.mkdir()
    for c in range(7):
        (folder / f"camera_{c}.mp4").write_bytes(b"not really a video")
    core = Core(str(folder), None, 0, [0, 1, 2, 3, 4, 5, 6])
    assert core.num_images == 2 and core.max_img_id == 1 and core.image_shape == [960, 480]
    assert core.output_folder == str(folder) + "_df3d" and os.path.isdir(core.output_folder)
    assert abs(core.fps - 30000 / 1001) < 1e-12
    assert len(open(log).read().split()) == 7
    Core(str(folder), None, 0, [0, 1, 2, 3, 4, 5, 6])  # frames are there now: no second expansion
    assert len(open(log).read().split()) == 7
    assert len(list(folder.glob("camera_*_img_*.jpg"))) == 14
    core.delete_images()
    assert not list(folder.glob("camera_*.jpg")) and len(list(folder.glob("camera_*.mp4"))) == 7
    # frame-rate strings ffprobe may print
    from deepfly3d_amd.os_util import parse_frame_rate

    assert parse_frame_rate("25\n") == 25.0 and parse_frame_rate("0/0") is None and parse_frame_rate("n/a") is None
    # no ffprobe at all -> fps None, like the reference when the command fails
    os.remove(bin_dir / "ffprobe")
    monkeypatch.setenv("PATH", str(bin_dir))
    (folder / "camera_0_img_0.jpg").write_bytes(open(os.path.join(src, "camera_0_img_0.jpg"), "rb").read())
    assert Core(str(folder), None, 0, [0, 1, 2, 3, 4, 5, 6]).fps is None
    config.pop("image_shape", None)


def test_plot_2d_draws_the_pose(golden_dir):
    """Camera.plot_2d / the drawing tables (reference df3d/core.py:298-319): an RGB image of the frame's size, coloured only
    around the seen joints and along their bones; unseen joints (0, .) are skipped."""
    from deepfly3d_amd.camera_network import CameraNetwork
    from deepfly3d_amd.config import LIMB_COLORS, limb_of_joint, skeleton_bones

    bones = skeleton_bones()
    assert len(bones) == 28 and [15, 34] not in bones and all(limb_of_joint(a) == limb_of_joint(b) for a, b in bones)
    assert [limb_of_joint(j) for j in (0, 4, 5, 14, 15, 16, 18, 19, 33, 34, 37)] == [0, 0, 1, 2, 3, 4, 4, 5, 7, 8, 9] and len(LIMB_COLORS) == 10
    g2 = np.load(f"{golden_dir}/golden_2d.npz")
    px = g2["points2d"][:, :2] * np.array([480.0, 960.0])
    net = CameraNetwork(px, image_path=os.path.join(golden_dir, "images", "camera_{cam_id}_img_{img_id}.jpg"))
    img = net[0].plot_2d(1)
    raw = net[0].get_image(1)
    assert img.shape == (480, 960, 3) and img.dtype == np.uint8
    changed = np.any(img != np.stack([raw] * 3, axis=-1) if raw.ndim == 2 else img != raw[..., :3], axis=-1)
    seen = (px[0, 1, :, 0] != 0) & (px[0, 1, :, 1] != 0)
    assert seen.sum() >= 15 and changed.sum() > 30 * seen.sum()
    for j in np.flatnonzero(seen):
        r, c = np.round(px[0, 1, j]).astype(int)
        assert changed[min(r, 479), min(c, 959)] and tuple(img[min(r, 479), min(c, 959)]) == LIMB_COLORS[limb_of_joint(j)]
    # nothing is drawn far from every seen joint / bone: the left-side joints of camera 0 are unseen
    assert not changed[:, :5].any()
    only = net[0].plot_2d(1, points2d=np.where((np.arange(38) == 2)[:, None], px[0, 1], 0.0))
    assert 50 < np.any(only != img * 0 + np.asarray(net[0].plot_2d(1, points2d=np.zeros((38, 2)))), axis=-1).sum() < 400


def test_mean_file_beside_the_checkpoint_is_adopted(tmp_path, monkeypatch):
    """The reference names the normalisation mean's source: weights/mean.pth.tar next to the checkpoint (reference
    df3d/config.py:37-39; bearpaw format {'mean': tensor[3], 'std': tensor[3]}).  load_state_dict() reads it; an explicit
    DF3D_PREPROCESS mean wins; the file's std is applied only on request (bearpaw's color_normalize subtracts only)."""
    import importlib

    import torch

    from deepfly3d_amd import inference

    ckpt = tmp_path / "sh8_deepfly.tar"
    torch.save({"state_dict": {"module.conv1.weight": torch.zeros(1)}, "epoch": 1}, ckpt)
    torch.save({"mean": torch.tensor([0.2154, 0.2154, 0.2154]), "std": torch.tensor([0.25, 0.25, 0.25])}, tmp_path / "mean.pth.tar")
    for k in ("DF3D_SYNTHETIC_WEIGHTS", "DF3D_PREPROCESS", "DF3D_MEAN", "DF3D_WEIGHTS"):
        monkeypatch.delenv(k, raising=False)
    try:
        importlib.reload(inference)
        assert inference.PREPROCESS == {"mean": (0.22, 0.22, 0.22), "std": (1.0, 1.0, 1.0), "resize": "bilinear"}
        sd = inference.load_state_dict(str(ckpt))
        assert list(sd) == ["conv1.weight"]
        assert np.allclose(inference.PREPROCESS["mean"], 0.2154) and inference.PREPROCESS["std"] == (1.0, 1.0, 1.0)
        assert inference._PREPROCESS_SOURCE["mean"].endswith("mean.pth.tar")
        # explicit settings win over the file; resize is data too; the file's std only on request
        monkeypatch.setenv("DF3D_PREPROCESS", '{"mean": [0.3], "resize": "area"}')
        importlib.reload(inference)
        inference.load_state_dict(str(ckpt))
        assert inference.PREPROCESS["mean"] == (0.3, 0.3, 0.3) and inference.PREPROCESS["resize"] == "area"
        monkeypatch.setenv("DF3D_PREPROCESS", '{"divide_by_std": true}')
        importlib.reload(inference)
        inference.load_state_dict(str(ckpt))
        assert np.allclose(inference.PREPROCESS["mean"], 0.2154) and np.allclose(inference.PREPROCESS["std"], 0.25)
        # a one-element mean (grey data set) is replicated; $DF3D_MEAN names a file elsewhere
        other = tmp_path / "elsewhere" / "m.pth.tar"
        other.parent.mkdir()
        torch.save({"mean": torch.tensor([0.5])}, other)
        monkeypatch.delenv("DF3D_PREPROCESS")
        monkeypatch.setenv("DF3D_MEAN", str(other))
        importlib.reload(inference)
        inference.load_state_dict(str(ckpt))
        assert inference.PREPROCESS["mean"] == (0.5, 0.5, 0.5)
        monkeypatch.setenv("DF3D_PREPROCESS", '{"resize": "lanczos"}')
        with pytest.raises(ValueError, match="resize must be one of"):
            importlib.reload(inference)
    finally:
        for k in ("DF3D_PREPROCESS", "DF3D_MEAN"):
            monkeypatch.delenv(k, raising=False)
        importlib.reload(inference)
