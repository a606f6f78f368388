"""Multi-GPU layout of the hot path: one process per GPU, frames sharded by contiguous range, ONE gather of the
per-frame results to rank 0 (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" on CPU for
tests).  There is no collective on the data path before that: views/frames are independent for 2-D inference,
arg-max, re-layout and triangulation; bundle-adjustment windows are assigned whole to a rank; Procrustes is
sequence-global and runs on rank 0 after the gather.

Per frame the gather moves ONE packed record of 5 704 B: 4 256 B points2d + 532 B confidence (+4 B pad) + 912 B
points3d; 100 k frames = 570 MB in total, ~71 MB per rank -- one `dist.gather`, i.e. one direct point-to-point
transfer per peer, no ring.  Per-window camera parameters (configs[4]) ride behind the frame records.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (as torch.distributed.run sets them).
    Returns (rank, world_size, local_rank).  Single-process runs need no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DF3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def current():
    """(rank, world_size) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def local_device(local_rank=None):
    """The GPU of this process: cuda:LOCAL_RANK (wrapped when fewer devices are visible, e.g. 2 test ranks on 1 GPU)."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = torch.cuda.device_count()
    return torch.device(f"cuda:{local_rank % n}") if n else torch.device("cpu")


def shard_range(num_frames, world_size, rank, align=1):
    """Contiguous frame range [start, stop) of `rank`; boundaries are multiples of `align` (e.g. the 1 000-frame
    bundle-adjustment window) except the last.  Ranges cover [0, num_frames) exactly, in rank order."""
    if num_frames < 0 or world_size < 1 or not (0 <= rank < world_size) or align < 1:
        raise ValueError("bad shard arguments")
    units = (num_frames + align - 1) // align
    base, extra = divmod(units, world_size)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, num_frames), min(u1 * align, num_frames)


def all_ranges(num_frames, world_size, align=1):
    return [shard_range(num_frames, world_size, r, align) for r in range(world_size)]


def _wire_tensor(t, group):
    """The tensor as the collective's back-end needs it: gloo has no CUDA gather (CPU tests, or several test ranks
    sharing one GPU), RCCL wants device memory."""
    backend = dist.get_backend(group)
    if t.is_cuda and backend == "gloo":
        return t.cpu()
    if not t.is_cuda and backend == "nccl":
        return t.to(local_device())
    return t


class RemoteRankError(RuntimeError):
    """Another rank failed in a section every rank has to leave together (see `agree`)."""


def agree(error=None, what="", group=None):
    """Make a rank-local failure collective.  Every rank calls this at the same point with the exception it caught in the
    section before it (or None); ONE small all-reduce tells everybody whether anybody failed.  Without a failure it returns;
    with one, the ranks that failed re-raise their own exception and the others raise RemoteRankError -- so that a caller that
    catches per-folder errors and moves on (cli.run_in_folders) moves on with ALL ranks, instead of leaving the peers blocked in
    the next collective of a folder rank 0 has already abandoned (round-3 advisor finding: broadcast / barrier mismatch until the
    back-end's timeout).  A no-op without a process group."""
    rank, world = current()
    if world <= 1:
        if error is not None:
            raise error
        return
    flag = torch.tensor([0 if error is None else rank + 1], dtype=torch.int32)
    wire = _wire_tensor(flag, group)
    dist.all_reduce(wire, op=dist.ReduceOp.MAX, group=group)
    worst = int(wire.cpu()[0])
    if error is not None:
        raise error
    if worst:
        raise RemoteRankError(f"rank {worst - 1} failed{' in ' + what if what else ''}; this rank (rank {rank}) abandons the step with it")


def primary_section(fn, what="", heartbeat_s=60.0, group=None):
    """Run `fn(beat)` on rank 0 only while the peers WAIT for its outcome -- for work that only the primary does but every rank has
    to leave together (the video stage of cli.run: files rank 0 writes; the next collective belongs to the next step / folder).
    One small MAX all-reduce per signal: 0 = still working, 1 = done, r + 2 = rank r failed.  Rank 0 calls `beat()` from inside its
    loop; at most every `heartbeat_s` seconds that sends a 0, so no peer ever sits in one collective longer than a heartbeat and
    the back-end's watchdog (10 min by default) cannot fire on a long encode.  A failure on rank 0 is re-raised there and raised
    as RemoteRankError on the peers (round-4 advisor finding: rank 0 moving on alone left the peers in delete_images' barrier).
    Returns fn's value on rank 0, None elsewhere.  Without a process group it simply runs fn."""
    import time

    rank, world = current()
    if world <= 1:
        return fn(lambda: None)

    def signal(code):
        wire = _wire_tensor(torch.tensor([code], dtype=torch.int32), group)
        dist.all_reduce(wire, op=dist.ReduceOp.MAX, group=group)
        return int(wire.cpu()[0])

    if rank == 0:
        last = [time.monotonic()]

        def beat():
            if time.monotonic() - last[0] >= heartbeat_s:
                signal(0)
                last[0] = time.monotonic()

        try:
            out = fn(beat)
        except BaseException:
            signal(2)
            raise
        signal(1)
        return out
    while True:
        code = signal(0)
        if code == 1:
            return None
        if code >= 2:
            raise RemoteRankError(f"rank {code - 2} failed{' in ' + what if what else ''}; this rank (rank {rank}) abandons the step with it")


def _gather_rows(rows, counts, rank, world_size, group=None, force_collective=False):
    """ONE `dist.gather` of a [n_r, width] uint8 record table per rank (n_r = counts[rank]) to rank 0, which
    returns the concatenation in rank order; other ranks return None.  Shards are padded to the longest one.
    With one rank the collective is skipped unless `force_collective` (a 1-rank RCCL group: the call path the
    multi-GPU run takes, executable on a single GPU) and a process group exist."""
    collective = world_size > 1 or (force_collective and dist.is_available() and dist.is_initialized())
    if not collective:
        return rows
    longest = max(counts)
    wire = _wire_tensor(rows, group)
    if wire.shape[0] < longest:
        wire = torch.cat([wire, torch.zeros((longest - wire.shape[0], wire.shape[1]), dtype=wire.dtype, device=wire.device)], dim=0)
    wire = wire.contiguous()
    bufs = [torch.empty_like(wire) for _ in range(world_size)] if rank == 0 else None
    dist.gather(wire, gather_list=bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([bufs[r][: counts[r]] for r in range(world_size)], dim=0)


def _pack_rows(parts, n):
    """Byte-pack tensors that share their leading axis (n rows) into one [n, width] uint8 table; every field starts
    at a multiple of 8 bytes.  Returns (table, [(offset, nbytes, dtype, row_shape)])."""
    layout, width = [], 0
    for t in parts:
        nbytes = int(t[0].numel()) * t.element_size() if n else int(np.prod(t.shape[1:])) * t.element_size()
        layout.append((width, nbytes, t.dtype, tuple(t.shape[1:])))
        width += (nbytes + 7) // 8 * 8
    table = torch.zeros((n, width), dtype=torch.uint8, device=parts[0].device)
    for t, (off, nbytes, _, _) in zip(parts, layout):
        if n:
            table[:, off : off + nbytes] = t.contiguous().view(torch.uint8).reshape(n, nbytes)
    return table, layout


def _unpack_rows(table, layout):
    n = table.shape[0]
    return [table[:, off : off + nbytes].contiguous().view(dtype).reshape(n, *shape) for off, nbytes, dtype, shape in layout]


def collective_needed(world_size, force_collective=None):
    """True when the gather really runs: several ranks, or a 1-rank process group with the collective forced
    (DF3D_FORCE_COLLECTIVE=1: exercises the RCCL call path on a single GPU)."""
    if force_collective is None:
        force_collective = os.environ.get("DF3D_FORCE_COLLECTIVE", "0") not in ("", "0")
    return world_size > 1 or (bool(force_collective) and dist.is_available() and dist.is_initialized())


def gather_packed(parts, num_frames, align=1, group=None, force_collective=None):
    """ONE gather of several tensors that are sharded by `shard_range`: parts = [(tensor, frame_axis), ...].
    Rank 0 gets the list of full-sequence tensors (on the device of the inputs), other ranks None."""
    rank, world = current()
    if not collective_needed(world, force_collective):
        return [t for t, _ in parts]
    counts = [b - a for a, b in all_ranges(num_frames, world, align)]
    moved = [t.movedim(ax, 0) for t, ax in parts]
    table, layout = _pack_rows(moved, moved[0].shape[0])
    rows = _gather_rows(table, counts, rank, world, group, True)
    if rows is None:
        return None
    dev = parts[0][0].device
    return [u.to(dev).movedim(0, ax).contiguous() for u, (_, ax) in zip(_unpack_rows(rows, layout), parts)]


def gather_frames(local, frame_axis, num_frames, align=1, group=None, force_collective=False):
    """Gather a tensor that is sharded along `frame_axis` by `shard_range` to rank 0 (None on the other ranks)."""
    out = gather_packed([(local, frame_axis)], num_frames, align, group, force_collective)
    return None if out is None else out[0]


def gather_results(points2d, conf, points3d, num_frames, rank, world_size, align=1, group=None, cameras=None, force_collective=False):
    """THE collective of the path: one `dist.gather` (RCCL over xGMI on the GPU box) of one byte-packed record per
    frame.  Inputs are this rank's shard:
        points2d [7, Tr, 38, 2] f64, conf [7, Tr, 19] f32, points3d [Tr, 38, 3] f64          (5 704 B per frame)
        cameras  [Wr, 7, 12] f64 (optional): per bundle-adjustment window R (9) + tvec (3) of the 7 cameras; the
                 window records travel in the same buffer, behind the frame records (BASELINE configs[4]).
    Rank 0 receives the full-sequence tensors (frame / window order = rank order), other ranks receive None.
    Returns (points2d, conf, points3d) or, with `cameras`, (points2d, conf, points3d, cameras)."""
    ranges = all_ranges(num_frames, world_size, align)
    counts = [b - a for a, b in ranges]
    collective = world_size > 1 or (force_collective and dist.is_available() and dist.is_initialized())
    if not collective:
        return (points2d, conf, points3d) if cameras is None else (points2d, conf, points3d, cameras)
    Tr = points3d.shape[0]
    frame_tab, frame_layout = _pack_rows([points2d.movedim(1, 0), conf.movedim(1, 0), points3d], Tr)
    fw = frame_tab.shape[1]
    if cameras is not None:
        # windows per rank follow from the frame ranges (a window never straddles two ranks: `align`)
        wcounts = [(c + align - 1) // align for c in counts]
        # A rank whose window count is wrong must NOT raise in front of the collective (its peers would wait in the gather
        # until the back-end's timeout): it enters the gather like everybody else, its record says how many windows it
        # holds, and the mismatch is raised behind the collective -- on that rank and on rank 0.
        have = int(cameras.shape[0])
        keep = min(have, wcounts[rank])
        cam_tab, cam_layout = _pack_rows([cameras[:keep]], keep)
        cw = cam_tab.shape[1]
        # one flat byte record per rank: [window count (8 B) | longest frames x fw | longest windows x cw]
        lf, lw = max(counts), max(wcounts)
        flat = torch.zeros((1, 8 + lf * fw + lw * cw), dtype=torch.uint8, device=frame_tab.device)
        flat[0, :8] = torch.tensor([have], dtype=torch.int64).view(torch.uint8).to(flat.device)
        flat[0, 8 : 8 + Tr * fw] = frame_tab.reshape(-1)
        flat[0, 8 + lf * fw : 8 + lf * fw + cam_tab.numel()] = cam_tab.reshape(-1)
        rows = _gather_rows(flat, [1] * world_size, rank, world_size, group, force_collective)
        if have != wcounts[rank]:
            raise ValueError(f"rank {rank}: {have} camera windows for {counts[rank]} frames (align {align})")
        if rows is None:
            return None, None, None, None
        sent = rows[:, :8].contiguous().cpu().view(torch.int64).reshape(-1).tolist()
        bad = [(r, sent[r], wcounts[r]) for r in range(world_size) if sent[r] != wcounts[r]]
        if bad:
            raise ValueError("camera window counts do not match the frame ranges: " + ", ".join(f"rank {r} sent {n}, expected {e}" for r, n, e in bad))
        rows = rows[:, 8:]
        ftabs = torch.cat([rows[r, : counts[r] * fw].reshape(counts[r], fw) for r in range(world_size)], dim=0)
        ctabs = torch.cat([rows[r, lf * fw : lf * fw + wcounts[r] * cw].reshape(wcounts[r], cw) for r in range(world_size)], dim=0)
        cams = _unpack_rows(ctabs, cam_layout)[0].to(cameras.device)
    else:
        ftabs = _gather_rows(frame_tab, counts, rank, world_size, group, force_collective)
        if ftabs is None:
            return None, None, None
    p2, cf, p3 = (t.to(points3d.device) for t in _unpack_rows(ftabs, frame_layout))
    out = (p2.movedim(0, 1).contiguous(), cf.movedim(0, 1).contiguous(), p3)
    return out if cameras is None else (*out, cams)


def assemble_result(points2d, conf, points3d_wo, cameras, camera_ordering, procrustes=None):
    """Rank-0 epilogue: Procrustes on the full sequence + the reference's result dictionary
    (schema and key order of reference df3d/core.py:349-369).  `procrustes` defaults to the
    device implementation (deepfly3d_amd.procrustes.procrustes_separate: needs the GPU, no CPU fallback)."""
    if procrustes is None:
        from .procrustes import procrustes_separate as procrustes

    p3 = np.asarray(points3d_wo, np.float64)
    out = {c: {"R": cameras["R"][c], "tvec": cameras["tvec"][c], "distort": cameras["distort"][c], "intr": cameras["intr"][c]} for c in range(7)}
    out["points3d"] = procrustes(p3)
    out["points2d"] = np.asarray(points2d, np.float64)
    out["points3d_wo_procrustes"] = p3
    out["camera_ordering"] = np.asarray(camera_ordering)
    out["heatmap_confidence"] = np.asarray(conf, np.float64)[..., None]
    return out
