"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The path shards by trajectory samples (they are independent, identification/model.py:370); the only
exchange steps are
  * one all-reduce (sum) of the (P+k)^2 fp64 Gram (<= 1.9 MB for WALK-MAN: latency bound, one fused buffer), and
  * a binary tree over the ranks for the TSQR factor: log2(N) rounds, each a point-to-point send of an R factor's packed upper
    triangle over a direct xGMI link followed by ``fbr_tsqr_merge`` on the receiver, then (optionally) a broadcast of the result.
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


_TRIU_CACHE: dict = {}


def _triu_index(n: int, device) -> torch.Tensor:
    """Flat positions of the upper triangle (row-major) of an n x n matrix, cached per (n, device)."""
    key = (int(n), str(device))
    idx = _TRIU_CACHE.get(key)
    if idx is None:
        iu = torch.triu_indices(n, n, device=device)
        idx = _TRIU_CACHE[key] = (iu[0] * n + iu[1]).contiguous()
    return idx


def pack_triu(R: torch.Tensor) -> torch.Tensor:
    """The n (n + 1) / 2 entries of the upper triangle of a square factor, row by row: what travels over a link (0.93 MB
    instead of 1.86 MB for WALK-MAN's 482 columns)."""
    n = R.shape[0]
    return R.reshape(-1).index_select(0, _triu_index(n, R.device))


def unpack_triu(packed: torch.Tensor, n: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """Inverse of ``pack_triu`` (zeros below the diagonal)."""
    if out is None:
        out = torch.zeros((n, n), dtype=packed.dtype, device=packed.device)
    else:
        out.zero_()
    out.view(-1).index_copy_(0, _triu_index(n, packed.device), packed)
    return out


def tsqr_tree(R: torch.Tensor, merge: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], group=None, packed: bool = True,
              broadcast: bool = True) -> torch.Tensor | None:
    """Binary-tree reduction of per-rank triangular factors.

    ``merge(Ra, Rb)`` must return the R factor of [Ra; Rb] (``Engine.tsqr_merge`` on the GPU).  ``packed``: square upper-triangular factors
    travel as their packed upper triangle (half the bytes of every hop and of the broadcast).  ``broadcast=True``: every rank returns the
    global factor; ``False``: only rank 0 of the group does (the others return None) -- for callers whose consumer (the OLS / SDP inputs)
    lives on rank 0, the step behind the last merge disappears."""
    if not dist.is_initialized():
        return R
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return R
    R = R.contiguous()
    n = R.shape[0]
    packed = bool(packed) and R.dim() == 2 and R.shape[0] == R.shape[1]
    # rank arithmetic is group-local; send / recv / broadcast take GLOBAL ranks
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    step = 1
    root = True
    while step < world:
        if rank % (2 * step) == 0:
            src = rank + step
            if src < world:
                if packed:
                    buf = torch.empty(n * (n + 1) // 2, dtype=R.dtype, device=R.device)
                    dist.recv(buf, src=g(src), group=group)
                    other = unpack_triu(buf, n)
                else:
                    other = torch.empty_like(R)
                    dist.recv(other, src=g(src), group=group)
                R = merge(R, other).contiguous()
        elif rank % (2 * step) == step:
            dist.send(pack_triu(R) if packed else R, dst=g(rank - step), group=group)
            root = False
            break
        step *= 2
    if not broadcast:
        return R if root else None
    if packed:
        buf = pack_triu(R) if root else torch.empty(n * (n + 1) // 2, dtype=R.dtype, device=R.device)
        dist.broadcast(buf, src=g(0), group=group)
        return R if root else unpack_triu(buf, n)
    if not root:
        R = torch.empty_like(R)  # (never into the caller's own factor)
    dist.broadcast(R, src=g(0), group=group)
    return R


# ---------------------------------------------------------------------------------------------------------------------------------
# First contact with hardware: a self-check of every exchange step, with a watchdog instead of a hang
# ---------------------------------------------------------------------------------------------------------------------------------
class Watchdog:
    """Names the step a rank is in and, if the step does not finish in ``seconds``, prints WHERE this rank is stuck and what to look at,
    then ends the process (a blocked RCCL call cannot be interrupted from Python: the launcher sees the exit code and tears the job
    down instead of waiting for the collective's own time-out).  ``with Watchdog(60, "warm_p2p") as wd: ...; wd.step("next")``."""

    HINT = ("hints: NCCL_DEBUG=INFO (RCCL's own log: transport per peer, the call it is in); HSA_ENABLE_IPC_MODE_LEGACY=0 must be set for "
            "multi-process GPU work on this image (dmabuf IPC; without it hipIpcGetMemHandle fails); rendezvous on 127.0.0.1 "
            "(MASTER_ADDR); every rank needs its own device (LOCAL_RANK); `rocm-smi --showtopo` shows the xGMI links")

    def __init__(self, seconds: float, what: str, rank: int | None = None, hard_exit: bool = True):
        self.seconds, self.cur, self.hard_exit = float(seconds), what, hard_exit
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self._timer = None
        self.fired = False

    def _arm(self):
        import threading

        self._disarm()
        self._timer = threading.Timer(self.seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def _disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def _fire(self):
        import os
        import sys

        self.fired = True
        print(f"[rank {self.rank}] distributed step '{self.cur}' did not finish within {self.seconds:.0f} s -- a peer is missing, stuck in another "
              f"step or its link is down.  {self.HINT}", file=sys.stderr, flush=True)
        if self.hard_exit:
            os._exit(3)

    def step(self, what: str) -> None:
        self.cur = what
        self._arm()

    def __enter__(self):
        self._arm()
        return self

    def __exit__(self, *exc):
        self._disarm()
        return False


def _edge_payload(recv: int, send: int, n: int = 4096) -> torch.Tensor:
    """what the sender of a tree edge transmits in the self-check: a vector every rank can rebuild"""
    g = torch.Generator().manual_seed(1000003 * recv + send + 17)
    return torch.randn(n, dtype=torch.float64, generator=g)


def selfcheck(device=None, merge: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] | None = None, timeout: float = 120.0, n: int = 48,
              group=None) -> dict:
    """Exercise every exchange step of the N > 1 path once, on data whose result every rank can compute alone, and FAIL WITH A MESSAGE
    (which step, which peer, what was expected) instead of hanging or producing a wrong reduction:
      1. all-reduce of a small tensor (rank r contributes r + 1: the sum is known);
      2. one send / recv over every edge of the TSQR rank tree with a payload the receiver rebuilds and compares bit for bit;
      3. ``tsqr_tree`` on seeded random triangles against the same merges done locally, and R^T R against the sum of the R_r^T R_r;
      4. broadcast from rank 0.
    ``merge``: the R-factor merge the run uses (``Engine.tsqr_merge``); None: NumPy QR.  Returns the seconds every step took."""
    import time

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {"world": 1}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if merge is None:
        import numpy as np

        def merge(Ra, Rb):  # noqa: E306
            return torch.from_numpy(np.linalg.qr(np.vstack([Ra.cpu().numpy(), Rb.cpu().numpy()]), mode="r")).to(Ra.device)

    out: dict = {"world": world}

    bad: list = []

    def fail(msg):  # recorded, raised by EVERY rank together at the end of the step (agree): a rank that raised at once would leave its
        bad.append(msg)  # peers blocked in the next exchange until their watchdogs fire, with N - 1 misleading "did not finish" messages

    def agree(step: str):
        flag = torch.tensor([1.0 if bad else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
        if bad:
            raise RuntimeError(f"[rank {rank}] distributed self-check failed: {bad[0]}.  {Watchdog.HINT}")
        if float(flag.item()) > 0:
            raise RuntimeError(f"[rank {rank}] distributed self-check: {int(flag.item())} other rank(s) reported a failure in step '{step}' (see their "
                               "message); this rank's own checks passed")

    with Watchdog(timeout, "all-reduce", rank) as wd:
        t0 = time.perf_counter()
        x = torch.full((8,), float(rank + 1), dtype=torch.float64, device=device)
        dist.all_reduce(x, group=group)
        want = world * (world + 1) / 2.0
        if not bool((x.cpu() == want).all()):
            fail(f"all-reduce over {world} ranks gave {x.cpu().tolist()} instead of {want}")
        agree("all-reduce")
        out["allreduce_s"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        for recv, send in _tree_edges(world):
            wd.step(f"send/recv over tree edge {send} -> {recv}")
            if rank == send:
                dist.send(_edge_payload(recv, send).to(device) if device is not None else _edge_payload(recv, send), dst=g(recv), group=group)
            elif rank == recv:
                buf = torch.empty(4096, dtype=torch.float64, device=device)
                dist.recv(buf, src=g(send), group=group)
                if not torch.equal(buf.cpu(), _edge_payload(recv, send)):
                    ndiff = int((buf.cpu() != _edge_payload(recv, send)).sum())
                    fail(f"payload received over tree edge {send} -> {recv} differs from what rank {send} sends in {ndiff} of 4096 entries")
        wd.step("status exchange after the tree edges")
        agree("send/recv over the tree edges")
        out["edges_s"] = time.perf_counter() - t0

        wd.step("tsqr_tree on random triangles")
        t0 = time.perf_counter()

        def tri(r):
            gg = torch.Generator().manual_seed(7919 * r + 5)
            return torch.triu(torch.randn((n, n), dtype=torch.float64, generator=gg))

        mine = tri(rank).to(device) if device is not None else tri(rank)
        R = tsqr_tree(mine, merge, group=group)
        # the same tree, merged locally
        loc = {r: (tri(r).to(device) if device is not None else tri(r)) for r in range(world)}
        step = 1
        while step < world:
            for r in range(0, world, 2 * step):
                if r + step < world:
                    loc[r] = merge(loc[r], loc[r + step]).contiguous()
            step *= 2
        ref = loc[0].cpu()
        gram = sum(tri(r).T @ tri(r) for r in range(world))
        Rc = R.cpu()
        e1 = float(torch.linalg.norm(Rc.T @ Rc - gram) / torch.linalg.norm(gram))
        e2 = float(torch.linalg.norm(Rc - ref) / torch.linalg.norm(ref))
        if not (e1 <= 1e-12 and e2 <= 1e-12):
            fail(f"tsqr_tree over {world} ranks: ||R^T R - sum R_r^T R_r|| / ||.|| = {e1:.2e}, vs the locally merged tree {e2:.2e} (expected <= 1e-12)")
        agree("tsqr_tree on random triangles")
        out["tsqr_tree_s"] = time.perf_counter() - t0
        out["tsqr_tree_relerr"] = max(e1, e2)

        wd.step("broadcast from rank 0")
        t0 = time.perf_counter()
        y = torch.full((8,), 42.0 if rank == 0 else -1.0, dtype=torch.float64, device=device)
        dist.broadcast(y, src=g(0), group=group)
        if not bool((y.cpu() == 42.0).all()):
            fail("broadcast from rank 0 did not arrive")
        agree("broadcast")
        out["broadcast_s"] = time.perf_counter() - t0
    return out
