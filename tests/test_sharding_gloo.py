"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: frame sharding, the scatter/gather
fan-out and the bench's max-over-ranks clock. The per-frame compute is stood in by a rank-local byte op
because no GPU exists here; what is under test is that every frame is owned exactly once and comes back in
order, and that the aggregate rate uses the slowest rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zignal_amd.sharding import (chunk_ranges, count_ranks, gather_frames, max_over_ranks, per_rank, scatter_compute_gather, scatter_frames,
                                 shard_range, shard_sizes, whole_job_rate)


def test_shard_ranges_cover_once():
    for n in (0, 1, 7, 8, 9, 1024, 1023):
        for world in (1, 2, 3, 4, 8):
            got = [shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert shard_sizes(1024, 8) == [128] * 8  # BASELINE configs[4]: 128 frames per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        shape = (6, 5, 4)
        batch = None
        if rank == 0:
            batch = (torch.arange(n_frames * 120, dtype=torch.int64) % 251).to(torch.uint8).reshape((n_frames,) + shape)
        mine = scatter_frames(batch, n_frames, shape, torch.uint8, dev)
        b, e = shard_range(n_frames, rank, world)
        assert mine.shape[0] == e - b
        processed = 255 - mine  # stand-in for the per-frame hot path (independent per frame)
        out = gather_frames(processed, n_frames)
        slow = max_over_ranks(0.5 + rank, dev)  # rank 1 is the slow one
        rate = whole_job_rate(int(mine.shape[0]), 0.5 + rank, dev)
        if rank == 0:
            results["ok"] = bool(torch.equal(out, 255 - batch))
            results["slow"] = slow
            results["rate"] = rate
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", (7, 8, 1))
def test_scatter_process_gather_world2(n_frames):
    world = 2
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), n_frames, results), nprocs=world, join=True)
        assert results["ok"]
        assert results["slow"] == 1.5
        assert abs(results["rate"] - n_frames / 1.5) < 1e-9


def test_chunk_ranges():
    assert chunk_ranges(0, 4) == []
    assert chunk_ranges(3, 4) == [(0, 1), (1, 2), (2, 3)]  # never an empty piece
    assert chunk_ranges(128, 4) == [(0, 32), (32, 64), (64, 96), (96, 128)]
    for n in (1, 5, 17, 128):
        for c in (1, 2, 4, 7):
            cr = chunk_ranges(n, c)
            assert cr[0][0] == 0 and cr[-1][1] == n and all(a[1] == b[0] for a, b in zip(cr, cr[1:])) and all(b > a for a, b in cr)


def _pipeline_worker(rank, world, port, n_frames, chunks, loopback, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        second = dist.new_group(backend="gloo")  # the results travel on a communicator of their own
        shape, out_shape = (6, 5, 4), (3, 5, 4)
        batch = None
        if rank == 0:
            batch = (torch.arange(n_frames * 120, dtype=torch.int64) % 251).to(torch.uint8).reshape((n_frames,) + shape)
        calls = []

        def compute(src, dst):  # stand-in for [blur, resize]: per frame, changes the frame's shape
            calls.append(int(src.shape[0]))
            dst.copy_(255 - src[:, ::2])
        out = scatter_compute_gather(batch, n_frames, shape, out_shape, torch.uint8, dev, compute, chunks=chunks, loopback=loopback,
                                     gather_group=second)
        assert sum(calls) == shard_sizes(n_frames, world)[rank]
        seen = count_ranks(dev)
        everyone = per_rank(10.0 + rank, dev)
        if rank == 0:
            results["ok"] = bool(torch.equal(out, 255 - batch[:, ::2]))
            results["pieces"] = len(calls)
            results["seen"] = seen
            results["everyone"] = everyone
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,chunks,loopback", ((2, 9, 4, False), (2, 8, 1, False), (3, 7, 4, False), (2, 1, 4, False), (3, 2, 4, False)))
def test_pipelined_scatter_compute_gather(world, n_frames, chunks, loopback):
    """The chunked exchange (what bench.py --scatter-gather times): every frame is computed exactly once, by its owner, and lands at its
    index on the root — including shards smaller than the chunk count and empty shards. (The one-rank loop-back needs a backend that can
    send to itself: tests/test_rccl_world1.py runs it over RCCL.)"""
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_pipeline_worker, args=(world, _free_port(), n_frames, chunks, loopback, results), nprocs=world, join=True)
        assert results["ok"]
        assert results["seen"] == world
        assert list(results["everyone"]) == [10.0 + r for r in range(world)]
        own = shard_sizes(n_frames, world)[0]
        assert results["pieces"] == (len(chunk_ranges(own, chunks)) if loopback else (1 if own else 0))
