"""Data-parallel harness: one process per GPU, graphs sharded by edge count, ONE flat-gradient
all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference has no distributed code at all; the DGN batches are unions of
independent small graphs (``dgl.batch``, data/molecules.py:229), so the path shards with no exchange
inside the layer: every rank runs the full layer stack on its own graphs, and the only collective is
the gradient average.  The parameter set of a DGN net is ~0.1-0.3 M fp32 values (~1 MB): the
all-reduce is latency-bound on a ~153 GB/s xGMI link, so one call on one pre-flattened buffer, no
bucketing, no overlap machinery.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from RANK/WORLD_SIZE/LOCAL_RANK; initialises the process group
    when WORLD_SIZE > 1 (MASTER_ADDR/MASTER_PORT from the launcher, 127.0.0.1 by default)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    force = os.environ.get("DGN_FORCE_DIST") == "1"      # exercise the RCCL path on a single GPU (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_by_edges(edge_counts: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-first partition of graph indices into ``world`` shards with balanced EDGE
    totals (edges, not graph count, are what the sweep pays for).  Deterministic."""
    order = sorted(range(len(edge_counts)), key=lambda i: (-int(edge_counts[i]), i))
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(edge_counts[i])
    for s in shards:
        s.sort()
    return shards


class FlatGradAllReduce:
    """Average the gradients of ``params`` across ranks with one all-reduce on one flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()

    def __call__(self) -> None:
        if not self.active:
            return
        views, off = [], 0
        for p in self.params:
            views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / self.world)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def barrier_max_ms(ms: float, device) -> float:
    """MAX over ranks of a per-rank duration."""
    if not dist.is_initialized():
        return ms
    t = torch.tensor([ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
