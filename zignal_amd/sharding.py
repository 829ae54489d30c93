"""Frame sharding across the GPUs of one node.

The image hot path has no cross-image state: every op is per frame, so a batch shards embarrassingly
(SURVEY §8e). One process per GPU; rank r owns a contiguous block of frames; there is NO collective on the
data path and no halo (a single frame is never split across GPUs).

The only exchange a deployment may want is the distribution step itself — a root GPU holding the whole batch
fans the shards out and collects the results. That is one scatter and one gather over RCCL (backend "nccl" on
ROCm), point-to-point over xGMI; `scatter_frames` / `gather_frames` implement exactly that and nothing else.
With the "gloo" backend the same code runs on CPU tensors (used by the tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of frames owned by `rank`; blocks differ by at most one frame."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    base, extra = divmod(n_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(n_frames: int, world: int) -> List[int]:
    return [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]


def scatter_frames(batch: Optional[torch.Tensor], n_frames: int, frame_shape: Tuple[int, ...], dtype: torch.dtype,
                   device: torch.device, root: int = 0, loopback: bool = False) -> torch.Tensor:
    """Root holds `batch` (n_frames, *frame_shape); every rank returns its own shard (k, *frame_shape).

    Implemented as grouped point-to-point sends (ncclSend/ncclRecv == the scatter pattern): each peer link
    carries one shard, the per-link bound of xGMI, no ring. `loopback` sends the root's own shard through the
    communicator too (a send to and a receive from itself in one group) instead of a device copy: that is how a
    one-GPU box executes the RCCL path for real."""
    rank, world = dist.get_rank(), dist.get_world_size()
    begin, end = shard_range(n_frames, rank, world)
    mine = torch.empty((end - begin,) + tuple(frame_shape), dtype=dtype, device=device)
    if rank == root:
        assert batch is not None and batch.shape[0] == n_frames
        ops = []
        for r in range(world):
            b, e = shard_range(n_frames, r, world)
            if r == root and loopback and e > b:
                ops.append(dist.P2POp(dist.isend, batch[b:e].contiguous(), root))
                ops.append(dist.P2POp(dist.irecv, mine, root))
            elif r == root:
                mine.copy_(batch[b:e])
            elif e > b:
                ops.append(dist.P2POp(dist.isend, batch[b:e].contiguous(), r))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    elif end > begin:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine, root)]):
            req.wait()
    return mine


def gather_frames(shard: torch.Tensor, n_frames: int, root: int = 0, loopback: bool = False) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames: root returns (n_frames, *frame_shape), the others None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == root:
        out = torch.empty((n_frames,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        ops = []
        for r in range(world):
            b, e = shard_range(n_frames, r, world)
            if r == root and loopback and e > b:
                ops.append(dist.P2POp(dist.isend, shard.contiguous(), root))
                ops.append(dist.P2POp(dist.irecv, out[b:e], root))
            elif r == root:
                out[b:e].copy_(shard)
            elif e > b:
                ops.append(dist.P2POp(dist.irecv, out[b:e], r))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if shard.shape[0] > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, shard.contiguous(), root)]):
            req.wait()
    return None


def max_over_ranks(seconds: float, device: torch.device) -> float:
    """The bench's clock: the slowest rank defines the step time."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank: int, seconds: float, device: torch.device) -> float:
    """Units all ranks processed / max-over-ranks time."""
    total = units_per_rank
    if dist.is_initialized() and dist.get_world_size() > 1:
        u = torch.tensor([units_per_rank], dtype=torch.float64, device=device)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total = float(u.item())
    return total / max_over_ranks(seconds, device)
