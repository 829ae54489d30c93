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

from zignal_amd.sharding import gather_frames, max_over_ranks, scatter_frames, shard_range, shard_sizes, whole_job_rate


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
