"""Frame sharding across the GPUs of one node.

The image hot path has no cross-image state: every op is per frame, so a batch shards embarrassingly
(SURVEY §8e). One process per GPU; rank r owns a contiguous block of frames; there is NO collective on the
data path and no halo (a single frame is never split across GPUs).

The only exchange a deployment may want is the distribution step itself — a root GPU holding the whole batch
fans the shards out and collects the results. That is one scatter and one gather over RCCL (backend "nccl" on
ROCm), point-to-point over xGMI: `scatter_frames` / `gather_frames` do exactly that, and
`scatter_compute_gather` is the same exchange cut into chunks so that a rank computes chunk c while chunk
c + 1 is still arriving and chunk c - 1 is already on its way back (the results travel on a second
communicator, so the two directions of a link run at the same time). With the "gloo" backend the same code
runs on CPU tensors (used by the tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

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


def chunk_ranges(n: int, chunks: int) -> List[Tuple[int, int]]:
    """A shard of n frames cut into at most `chunks` contiguous pieces (none empty); every rank computes the same cut."""
    chunks = max(1, min(chunks, n))
    return [shard_range(n, c, chunks) for c in range(chunks)] if n else []


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


def scatter_compute_gather(batch: Optional[torch.Tensor], n_frames: int, frame_shape: Tuple[int, ...], out_frame_shape: Tuple[int, ...],
                           dtype: torch.dtype, device: torch.device, compute: Callable[[torch.Tensor, torch.Tensor], None],
                           chunks: int = 4, root: int = 0, loopback: bool = False, gather_group=None,
                           out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """scatter -> compute(shard_in, shard_out) -> gather as a pipeline over `chunks` pieces of every shard.

    The root posts every piece of every peer's shard (pieces of one peer in order, one grouped launch per piece index, so each
    xGMI link carries its own peer's pieces back to back) and works on its own frames in place meanwhile; a peer posts all its
    receives up front, then for each piece: wait for it (a stream wait, the host does not block), compute, send the result back
    through `gather_group` — a second process group, i.e. a second communicator with its own stream, so results flow back while
    later pieces are still arriving. `compute(src, dst)` must enqueue on the current stream. Returns (n_frames, *out_frame_shape)
    on the root (in `out` when given), None elsewhere. `loopback` treats the root as its own peer (one-GPU boxes)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(n_frames, world)
    begins = [shard_range(n_frames, r, world)[0] for r in range(world)]
    peers = [r for r in range(world) if (r != root or loopback) and sizes[r] > 0]
    pending = []  # keeps requests (and through them the tensors) alive until the end
    result = None
    if rank == root:
        assert batch is not None and batch.shape[0] == n_frames
        result = out if out is not None else torch.empty((n_frames,) + tuple(out_frame_shape), dtype=dtype, device=device)
    mine_in = mine_out = None
    if rank in peers:
        mine_in = torch.empty((sizes[rank],) + tuple(frame_shape), dtype=dtype, device=device)
        mine_out = torch.empty((sizes[rank],) + tuple(out_frame_shape), dtype=dtype, device=device)
    max_chunks = max([len(chunk_ranges(sizes[r], chunks)) for r in peers], default=0)

    # ---- scatter: piece c of every peer in one grouped launch; a peer's receives are posted in the same order -------------
    recv_reqs = {}
    for c in range(max_chunks):
        ops = []
        if rank == root:
            for r in peers:
                cr = chunk_ranges(sizes[r], chunks)
                if c < len(cr):
                    ops.append(dist.P2POp(dist.isend, batch[begins[r] + cr[c][0]:begins[r] + cr[c][1]], r))
        if rank in peers:
            cr = chunk_ranges(sizes[rank], chunks)
            if c < len(cr):
                ops.append(dist.P2POp(dist.irecv, mine_in[cr[c][0]:cr[c][1]], root))
        if ops:
            reqs = dist.batch_isend_irecv(ops)
            if rank in peers and c < len(chunk_ranges(sizes[rank], chunks)):
                # Everything this launch returned is waited for before piece c is computed, each request once (gloo hangs on a second
                # wait): a backend may hand back one request per op in any order, or one for the whole group (NCCL coalesces), so no
                # single element can be taken for "the receive". A peer's launch holds nothing but its receive; the loop-back root's
                # also holds its sends of piece c, and waiting for those is a stream wait behind work that is already queued.
                recv_reqs[c] = list(reqs)
            else:
                pending.extend(reqs)

    # ---- gather, root side: piece c of every (other) peer, posted before any local work so that nothing orders them behind it ---
    if rank == root:
        for c in range(max_chunks):
            ops = []
            for r in peers:
                if r == root:
                    continue
                cr = chunk_ranges(sizes[r], chunks)
                if c < len(cr):
                    ops.append(dist.P2POp(dist.irecv, result[begins[r] + cr[c][0]:begins[r] + cr[c][1]], r, group=gather_group))
            if ops:
                pending.extend(dist.batch_isend_irecv(ops))

    # ---- the root's own frames: in place, no copy (unless it is its own peer) -------------------------------------------------
    if rank == root and not loopback and sizes[root] > 0:
        b = begins[root]
        compute(batch[b:b + sizes[root]], result[b:b + sizes[root]])

    # ---- peers: piece by piece, the result straight back on the second communicator --------------------------------------------
    if rank in peers:
        for c, (c0, c1) in enumerate(chunk_ranges(sizes[rank], chunks)):
            for req in recv_reqs[c]:
                req.wait()  # orders the current stream behind the arrival; the host carries on
            compute(mine_in[c0:c1], mine_out[c0:c1])
            ops = [dist.P2POp(dist.isend, mine_out[c0:c1], root, group=gather_group)]
            if rank == root:  # loop-back: the send and its matching receive go into one group
                ops.append(dist.P2POp(dist.irecv, result[begins[root] + c0:begins[root] + c1], root, group=gather_group))
            pending.extend(dist.batch_isend_irecv(ops))
    for req in pending:
        req.wait()
    return result


def max_over_ranks(seconds: float, device: torch.device) -> float:
    """The bench's clock: the slowest rank defines the step time."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def per_rank(value: float, device: torch.device) -> List[float]:
    """`value` of every rank, in rank order, on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    everyone = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(everyone, mine)
    return [float(t.item()) for t in everyone]


def count_ranks(device: torch.device) -> int:
    """How many ranks the process group really spans: an all-reduce(SUM) of one 1 per rank."""
    if not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def whole_job_rate(units_per_rank: int, seconds: float, device: torch.device) -> float:
    """Units all ranks processed / max-over-ranks time."""
    total = units_per_rank
    if dist.is_initialized() and dist.get_world_size() > 1:
        u = torch.tensor([units_per_rank], dtype=torch.float64, device=device)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total = float(u.item())
    return total / max_over_ranks(seconds, device)
