"""Multi-GPU layout of the hot path: one process per GPU, frames sharded by contiguous range, ONE gather of the
per-frame results to rank 0 (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" on CPU for
tests).  There is no collective on the data path before that: views/frames are independent for 2-D inference,
arg-max, re-layout and triangulation; bundle-adjustment windows are assigned whole to a rank; Procrustes is
sequence-global and runs on rank 0 after the gather.

Per frame the gather moves 912 B (points3d) + 4 256 B (points2d) + 532 B (confidence): 100 k frames = 570 MB in
total, ~71 MB per rank -- one direct point-to-point transfer per peer, no ring.
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


def _gather_frame_axis(local, frame_axis, ranges, rank, world_size, group=None):
    """Gather tensors that are sharded along `frame_axis` to rank 0 (padded to the largest shard)."""
    if world_size == 1:
        return local
    longest = max(b - a for a, b in ranges)
    moved = local.movedim(frame_axis, 0).contiguous()
    if moved.is_cuda and dist.get_backend(group) == "gloo":
        moved = moved.cpu()  # gloo has no CUDA gather (CPU tests, or several ranks sharing one GPU)
    if moved.shape[0] < longest:
        pad = torch.zeros((longest - moved.shape[0], *moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        moved = torch.cat([moved, pad], dim=0)
    bufs = [torch.empty_like(moved) for _ in range(world_size)] if rank == 0 else None
    dist.gather(moved, gather_list=bufs, dst=0, group=group)
    if rank != 0:
        return None
    parts = [bufs[r][: ranges[r][1] - ranges[r][0]] for r in range(world_size)]
    return torch.cat(parts, dim=0).movedim(0, frame_axis).contiguous()


def gather_frames(local, frame_axis, num_frames, align=1, group=None):
    """Gather a tensor that is sharded along `frame_axis` by `shard_range` to rank 0 (None on the other ranks)."""
    rank, world = current()
    return _gather_frame_axis(local, frame_axis, all_ranges(num_frames, world, align), rank, world, group)


def gather_results(points2d, conf, points3d, num_frames, rank, world_size, align=1, group=None):
    """The single collective of the path.  Inputs are this rank's shard:
        points2d [7, Tr, 38, 2] f64, conf [7, Tr, 19] f32, points3d [Tr, 38, 3] f64
    Rank 0 receives the full-sequence tensors (frame order = rank order), other ranks receive None."""
    ranges = all_ranges(num_frames, world_size, align)
    p2 = _gather_frame_axis(points2d, 1, ranges, rank, world_size, group)
    cf = _gather_frame_axis(conf, 1, ranges, rank, world_size, group)
    p3 = _gather_frame_axis(points3d, 0, ranges, rank, world_size, group)
    return p2, cf, p3


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
