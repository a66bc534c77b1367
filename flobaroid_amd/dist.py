"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The path shards by trajectory samples (they are independent, identification/model.py:370); the only
exchange steps are
  * one all-reduce (sum) of the (P+k)^2 fp64 Gram (<= 1.9 MB for WALK-MAN: latency bound, one fused buffer), and
  * a binary tree over the ranks for the TSQR factor: log2(N) rounds, each a point-to-point send of an R factor
    over a direct xGMI link followed by ``fbr_tsqr_merge`` on the receiver, then a broadcast of the result.
The same functions run on CPU tensors with the gloo backend (that is how the N > 1 logic is tested without GPUs).
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_range(num_samples: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block of samples of ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(num_samples, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_gram(G: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of the per-rank Gram matrices."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    return G


def _tree_edges(world: int) -> list[tuple[int, int]]:
    """(receiver, sender) pairs of the binary tree, level by level (group-local ranks)."""
    edges, step = [], 1
    while step < world:
        edges += [(r, r + step) for r in range(0, world, 2 * step) if r + step < world]
        step *= 2
    return edges


def warm_p2p(device=None, group=None) -> None:
    """Set up the communicators the exchange steps use BEFORE anything is timed: RCCL creates a point-to-point communicator per
    peer pair lazily on the first send / recv (tens of milliseconds each), and the collective one on the first all-reduce.  One
    8-byte message over every edge of the rank tree, one all-reduce and one broadcast."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    x = torch.zeros(1, dtype=torch.float64, device=device)
    for recv, send in _tree_edges(world):
        if rank == recv:
            dist.recv(x, src=g(send), group=group)
        elif rank == send:
            dist.send(x, dst=g(recv), group=group)
    dist.all_reduce(x, group=group)
    dist.broadcast(x, src=g(0), group=group)


def tsqr_tree(R: torch.Tensor, merge: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], group=None) -> torch.Tensor:
    """Binary-tree reduction of per-rank triangular factors; every rank returns the global factor.

    ``merge(Ra, Rb)`` must return the R factor of [Ra; Rb] (``Engine.tsqr_merge`` on the GPU)."""
    if not dist.is_initialized():
        return R
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return R
    R = R.contiguous()
    # rank arithmetic is group-local; send / recv / broadcast take GLOBAL ranks
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    step = 1
    while step < world:
        if rank % (2 * step) == 0:
            src = rank + step
            if src < world:
                other = torch.empty_like(R)
                dist.recv(other, src=g(src), group=group)
                R = merge(R, other).contiguous()
        elif rank % (2 * step) == step:
            dist.send(R, dst=g(rank - step), group=group)
            break
        step *= 2
    dist.broadcast(R, src=g(0), group=group)
    return R
