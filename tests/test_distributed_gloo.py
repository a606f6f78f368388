"""CPU test of the N > 1 path: world_size-2 gloo processes shard a frame sequence, run the single gather, and
rank 0 must hold exactly the single-process result (shard-consistency: N-rank == 1-rank bit for bit)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _full_sequence(T):
    g = torch.Generator().manual_seed(7)
    p2 = torch.rand((7, T, 38, 2), generator=g, dtype=torch.float64)
    cf = torch.rand((7, T, 19), generator=g, dtype=torch.float32)
    p3 = torch.randn((T, 38, 3), generator=g, dtype=torch.float64)
    return p2, cf, p3


def _window_cameras(T, align):
    nwin = (T + align - 1) // align
    return torch.randn((nwin, 7, 12), generator=torch.Generator().manual_seed(11), dtype=torch.float64)


def _worker(rank, world, port, T, align, outdir, with_cameras=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from deepfly3d_amd import distributed as dd

    r, w, _ = dd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    p2, cf, p3 = _full_sequence(T)
    a, b = dd.shard_range(T, world, rank, align)
    cams = _window_cameras(T, align)[a // align : (b + align - 1) // align].contiguous() if with_cameras else None
    calls = []
    real_gather = dist.gather
    dist.gather = lambda *args, **kw: (calls.append(1), real_gather(*args, **kw))[1]
    out = dd.gather_results(p2[:, a:b].contiguous(), cf[:, a:b].contiguous(), p3[a:b].contiguous(), T, rank, world, align, cameras=cams)
    dist.gather = real_gather
    assert len(calls) == 1, "the data path has exactly ONE collective"
    if rank == 0:
        torch.save([t for t in out], os.path.join(outdir, "gathered.pt"))
    else:
        assert all(t is None for t in out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,align", [(37, 1), (40, 8), (3, 1)])
def test_two_rank_gather_equals_single_process(tmp_path, T, align):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, T, align, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    ref = _full_sequence(T)
    for g, r in zip(got, ref):
        assert g.dtype == r.dtype and torch.equal(g, r)


@pytest.mark.parametrize("T,align", [(40, 8), (35, 10)])
def test_two_rank_gather_with_window_cameras(tmp_path, T, align):
    """configs[4]: the per-window camera parameters ride in the same single collective as the frame records."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, T, align, str(tmp_path), True), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    ref = (*_full_sequence(T), _window_cameras(T, align))
    assert len(got) == 4
    for g, r in zip(got, ref):
        assert g.dtype == r.dtype and torch.equal(g, r)


@pytest.mark.parametrize("T,align,with_cameras", [(100, 10, False), (100, 10, True), (7, 1, False), (7, 1, True), (100000, 1000, True)])
def test_eight_rank_gather_with_uneven_and_empty_shards(tmp_path, T, align, with_cameras):
    """The node-level run is 8 ranks: uneven window counts ((100, 10): 2,2,1,1,1,1,1,1 windows; configs[3]/[4]'s
    (100 000, 1 000): 13 000 x 4 + 12 000 x 4 frames), a rank with NO frames ((7, 1): rank 7's shard is empty) -- padded
    shards, empty records and the window cameras all through the one collective, bit-identical to one process."""
    from deepfly3d_amd import distributed as dd

    ranges = dd.all_ranges(T, 8, align)
    assert ranges[0][0] == 0 and ranges[-1][1] == T and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    if (T, align) == (100, 10):
        assert [b - a for a, b in ranges] == [20, 20, 10, 10, 10, 10, 10, 10]
    if (T, align) == (7, 1):
        assert ranges[7] == (7, 7)
    if T == 100000:
        assert [b - a for a, b in ranges] == [13000] * 4 + [12000] * 4
    port = _free_port()
    mp.spawn(_worker, args=(8, port, T, align, str(tmp_path), with_cameras), nprocs=8, join=True)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    ref = (*_full_sequence(T), _window_cameras(T, align)) if with_cameras else _full_sequence(T)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g.dtype == r.dtype and torch.equal(g, r)


def _bad_worker(rank, world, port, T, align, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from deepfly3d_amd import distributed as dd

    dd.init_from_env(backend="gloo")
    p2, cf, p3 = _full_sequence(T)
    a, b = dd.shard_range(T, world, rank, align)
    cams = _window_cameras(T, align)[a // align : (b + align - 1) // align].contiguous()
    if rank == 1:
        cams = cams[:-1]   # one window short
    try:
        dd.gather_results(p2[:, a:b].contiguous(), cf[:, a:b].contiguous(), p3[a:b].contiguous(), T, rank, world, align, cameras=cams)
        outcome = "returned"
    except ValueError as e:
        outcome = "ValueError: " + str(e)
    with open(os.path.join(outdir, f"outcome{rank}.txt"), "w") as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_wrong_window_count_fails_behind_the_collective_not_in_front_of_it(tmp_path):
    """A rank with the wrong number of camera windows still enters the gather (its peers would otherwise hang in the
    collective until the back-end's timeout); the error is raised afterwards on that rank and on rank 0."""
    port = _free_port()
    mp.spawn(_bad_worker, args=(3, port, 60, 10, str(tmp_path)), nprocs=3, join=True)
    out = [open(os.path.join(tmp_path, f"outcome{r}.txt")).read() for r in range(3)]
    assert out[0].startswith("ValueError") and "rank 1 sent 1, expected 2" in out[0]
    assert out[1].startswith("ValueError") and "rank 1: 1 camera windows" in out[1]
    assert out[2] == "returned"


def _failing_rank0_worker(rank, world, port, outdir, golden):
    """cli.run's sequence on a Core whose 2-D stage is given: save -> calibrate_calc -> save, with rank 0 failing first in the
    bundle adjustment, then in the result write."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import datetime
    import pickle

    from deepfly3d_amd import camera_network, core as core_mod, distributed as dd

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    g2 = np.load(os.path.join(golden, "golden_2d.npz"))
    c = core_mod.Core.__new__(core_mod.Core)
    c.dtype, c.device, c.is_primary = "f32", None, rank == 0
    c._input_folder, c._output_folder = outdir, outdir
    c.num_images, c.max_img_id, c.image_shape = 15, 14, [960, 480]
    c._image_path = os.path.join(outdir, "camera_{cam_id}_img_{img_id}.jpg")
    c.camera_ordering = np.arange(7)
    c.points2d, c.conf = (g2["points2d"], g2["heatmap_confidence"]) if rank == 0 else (None, None)
    c.camNet = c.points3d = c._points2d_shard = None
    outcomes = []

    def attempt(fn, *a):
        try:
            fn(*a)
            outcomes.append("ok")
        except Exception as e:  # noqa: BLE001
            outcomes.append(f"{type(e).__name__}: {e}")

    def boom(self, **kw):
        raise RuntimeError("the solver blew up")

    camera_network.CameraNetwork.bundle_adjust = boom
    attempt(c.save)                      # 2-D only: fine everywhere
    attempt(c.calibrate_calc, 0, 14)     # rank 0 fails -> EVERY rank raises, nobody is left in a collective
    real_dump = pickle.dump
    core_mod.pickle.dump = lambda *a, **kw: (_ for _ in ()).throw(OSError(28, "No space left on device"))
    c.camNet = None
    attempt(c.save)                      # rank 0 fails writing -> every rank raises
    core_mod.pickle.dump = real_dump
    attempt(c.save)                      # and the ranks are still in step: the next collective step works
    dist.barrier()
    with open(os.path.join(outdir, f"outcome{rank}.txt"), "w") as f:
        f.write("\n".join(outcomes))
    dist.destroy_process_group()


def test_a_failure_on_rank0_between_the_saves_is_raised_on_every_rank(tmp_path, golden_dir):
    """Round-3 advisor finding: calibrate_calc / the result write run on rank 0 only; when they raised there, cli.run_in_folders
    moved rank 0 on to the next folder while its peers sat in the abandoned folder's next collective (mismatched broadcast / barrier
    until the back-end's timeout).  `distributed.agree` makes the failure collective: rank 0 re-raises its exception, every other
    rank raises RemoteRankError at the same point, and all of them are in step for whatever comes next."""
    port = _free_port()
    mp.spawn(_failing_rank0_worker, args=(2, port, str(tmp_path), str(golden_dir)), nprocs=2, join=True)
    out = [open(os.path.join(tmp_path, f"outcome{r}.txt")).read().split("\n") for r in range(2)]
    assert out[0] == ["ok", "RuntimeError: the solver blew up", "OSError: [Errno 28] No space left on device", "ok"]
    assert out[1][0] == "ok" and out[1][3] == "ok"
    assert out[1][1].startswith("RemoteRankError: rank 0 failed in calibrate_calc")
    assert out[1][2].startswith("RemoteRankError: rank 0 failed in save")
    assert os.path.exists(os.path.join(tmp_path, "df3d_result_" + str(tmp_path).replace("/", "_") + ".pkl"))


def _primary_section_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import datetime

    from deepfly3d_amd import distributed as dd

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    outcomes, ran = [], []

    def encode(beat):   # "the video stage": only rank 0 runs it, beating from inside its loop
        for k in range(5):
            ran.append(k)
            beat()
        return "video.mp4"

    def broken(beat):
        beat()
        raise RuntimeError("ffmpeg died")

    for fn in (encode, broken, encode):
        try:
            outcomes.append(repr(dd.primary_section(fn, "video", heartbeat_s=0.0)))
        except Exception as e:  # noqa: BLE001
            outcomes.append(f"{type(e).__name__}: {e}")
    dist.barrier()   # still in step: the next collective matches
    with open(os.path.join(outdir, f"outcome{rank}.txt"), "w") as f:
        f.write("\n".join(outcomes + [str(len(ran))]))
    dist.destroy_process_group()


def test_primary_section_releases_the_peers_with_rank0s_outcome(tmp_path):
    """Round-4 advisor finding: the video stage ran on rank 0 only and was not collective -- an encoder failure moved rank 0 on to the
    next folder while the peers sat in delete_images' barrier.  `distributed.primary_section`: rank 0 works (heart-beating, so no peer
    sits in one collective for the whole encode), every rank leaves with rank 0's outcome."""
    port = _free_port()
    mp.spawn(_primary_section_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    out = [open(os.path.join(tmp_path, f"outcome{r}.txt")).read().split("\n") for r in range(3)]
    assert out[0] == ["'video.mp4'", "RuntimeError: ffmpeg died", "'video.mp4'", "10"]
    for r in (1, 2):
        assert out[r][0] == "None" and out[r][2] == "None" and out[r][3] == "0"
        assert out[r][1].startswith("RemoteRankError: rank 0 failed in video")


def test_one_rank_group_executes_the_collective():
    """A 1-rank process group with force_collective runs the real `dist.gather` (what the GPU box does on RCCL with
    its single GPU) and returns the same tensors."""
    from deepfly3d_amd import distributed as dd

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        p2, cf, p3 = _full_sequence(9)
        cams = _window_cameras(9, 4)
        calls = []
        real_gather = dist.gather
        dist.gather = lambda *args, **kw: (calls.append(1), real_gather(*args, **kw))[1]
        try:
            out = dd.gather_results(p2, cf, p3, 9, 0, 1, align=4, cameras=cams, force_collective=True)
            one = dd.gather_frames(p2, 1, 9, force_collective=True)
        finally:
            dist.gather = real_gather
        assert len(calls) == 2
        for g, r in zip(out, (p2, cf, p3, cams)):
            assert g is not r and g.dtype == r.dtype and torch.equal(g, r)
        assert torch.equal(one, p2)
    finally:
        dist.destroy_process_group()


def test_single_rank_is_a_no_op():
    from deepfly3d_amd import distributed as dd

    p2, cf, p3 = _full_sequence(5)
    out = dd.gather_results(p2, cf, p3, 5, 0, 1)
    assert out[0] is p2 and out[1] is cf and out[2] is p3


def test_bench_eight_rank_dry_run_of_the_strong_scaled_stream():
    """`bench.py --dry-run`: the launch line the driver will use on an 8-GPU box (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8
    --strong --stream-frames 100000 --ba-window 1000`), executed here on CPU tensors over gloo with a stub in the pipeline's place: BASELINE
    configs[4]'s own split (13 000 x 4 + 12 000 x 4 frames, 102 batches on the largest shard), per-rank window records, ONE packed gather
    of 570 MB; rank 0 finds every frame and every window where it belongs."""
    import json
    import subprocess

    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--strong", "--stream-frames", "100000", "--ba-window", "1000", "--dry-run", "--verify"],
                       cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 1, r.stdout[-2000:]
    d = json.loads(rows[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 102
    assert d["frames_per_gpu"] == [13000] * 4 + [12000] * 4 and d["collective_backend"] == "gloo"
    assert d["gather_check"] is True and d["windows_gathered"] == 100
    # --verify: rank 0 recomputed a strided sample of EVERY rank's frames and found the gathered records bit-identical; per-rank rates, gather time
    v = d["verify"]
    assert v["bit_identical"] is True and v["ranks_sampled"] == 8 and v["frames_recomputed_on_rank0"] == 48 and v["ranks_differing"] == []
    assert len(d["per_rank_frames_per_s"]) == 8 and all(x > 0 for x in d["per_rank_frames_per_s"]) and d["gather_ms"] > 0 and d["rccl_world"] == 8


def test_verify_sample_plan_and_pool_replay():
    """bench.py --verify's pure pieces: the sample covers first / last / interior frames of every non-empty range, and a peer's frame pool
    regenerated chunk by chunk equals the pool that peer filled (CPU generator here; the GPU run uses the same code on `cuda`)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    plan = b.verify_sample_plan([(0, 13000), (13000, 13000), (13000, 13003), (13003, 13004)], 6)
    assert [r for r, _ in plan] == [0, 2, 3]
    assert plan[0][1][0] == 0 and plan[0][1][-1] == 12999 and len(plan[0][1]) == 6 and plan[1][1] == [0, 1, 2] and plan[2][1] == [0]
    old = b.POOL_CHUNK
    try:
        b.POOL_CHUNK = 3
        import torch

        dev = torch.device("cpu")
        pool = torch.empty((8, 7, 256, 512, 3))
        b.fill_pool(pool, 5, dev)
        got = b.pool_frames_of(5, 8, [0, 4, 7], dev)
        assert torch.equal(got, pool[[0, 4, 7]])
        assert not torch.equal(b.pool_frames_of(4, 8, [0], dev)[0], pool[0])   # another rank's seed: other frames
    finally:
        b.POOL_CHUNK = old
