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


def _worker(rank, world, port, T, align, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from deepfly3d_amd import distributed as dd

    r, w, _ = dd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    p2, cf, p3 = _full_sequence(T)
    a, b = dd.shard_range(T, world, rank, align)
    out = dd.gather_results(p2[:, a:b].contiguous(), cf[:, a:b].contiguous(), p3[a:b].contiguous(), T, rank, world, align)
    if rank == 0:
        torch.save([t for t in out], os.path.join(outdir, "gathered.pt"))
    else:
        assert out == (None, None, None)
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


def test_single_rank_is_a_no_op():
    from deepfly3d_amd import distributed as dd

    p2, cf, p3 = _full_sequence(5)
    out = dd.gather_results(p2, cf, p3, 5, 0, 1)
    assert out[0] is p2 and out[1] is cf and out[2] is p3
