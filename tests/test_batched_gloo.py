"""N > 1 path on CPU: instance sharding and the one all-gather, world_size 2, gloo."""

import os
import socket

import numpy as np
import pytest

from pygradflow_amd import batched


def test_shard_ranges_cover_batch():
    for B in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = batched.shard_range(B, world, r)
                assert 0 <= lo <= hi <= B
                seen.extend(range(lo, hi))
            assert seen == list(range(B))
    assert batched.shard_range(256, 8, 3) == (96, 128)  # instance i on rank i // 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = batched.shard_range(B, world, rank)
        local = torch.tensor([100.0 + i for i in range(lo, hi)], dtype=torch.float64)
        allv = batched.gather_residual_norms(local, B)
        q.put((rank, allv.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7, 1])
def test_all_gather_of_residual_norms_gloo(B):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [100.0 + i for i in range(B)]
    for r in range(world):
        assert got[r] == expect


def test_single_process_passthrough():
    import torch

    v = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    assert batched.gather_residual_norms(v, 3).tolist() == [1.0, 2.0, 3.0]
