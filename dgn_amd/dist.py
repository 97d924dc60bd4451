"""Data-parallel harness: one process per GPU, graphs sharded by edge count, ONE flat-gradient
all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference has no distributed code at all; the DGN batches are unions of
independent small graphs (``dgl.batch``, data/molecules.py:229), so the path shards with no exchange
inside the layer: every rank runs the full layer stack on its own graphs, and the only collective is
the gradient average.  The parameter set of a DGN net is ~0.1-0.3 M fp32 values (~1 MB): the
all-reduce is latency-bound on a ~153 GB/s xGMI link, so one call on one pre-flattened buffer, no
bucketing, no overlap machinery.

ONE giant graph (BASELINE config 5, the ogbn-scale power-law graph) is split the other way
(``row_ranges_by_edges`` / ``shard_rows`` / ``all_gather_rows``): every rank owns a contiguous range of destination
rows with a balanced edge count and keeps the node features whole; the sweep over a shard needs no exchange, the
layer OUTPUT rows (F wide, after the post-transformation) are all-gathered for the next layer, and the gradient
of the replicated features is summed with an all-reduce.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist

from .graph import DGNGraph


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from RANK/WORLD_SIZE/LOCAL_RANK; initialises the process group
    when WORLD_SIZE > 1 (MASTER_ADDR/MASTER_PORT from the launcher, 127.0.0.1 by default)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    force = os.environ.get("DGN_FORCE_DIST") == "1"      # exercise the RCCL path on a single GPU (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: start the ranks with a launcher (torch.distributed.run) "
                                   "or export MASTER_ADDR / MASTER_PORT")
            import socket                      # a forced single-rank group (tests): any free port, never a fixed one that may be taken
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
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


def row_ranges_by_edges(indptr: torch.Tensor, world: int) -> List[Tuple[int, int]]:
    """Cut the destination rows of ONE graph into ``world`` contiguous ranges with balanced edge totals
    (dst-range CSR shards).  ``indptr`` [N+1] on any device; deterministic."""
    n = indptr.numel() - 1
    ip = indptr.long()
    total = int(ip[-1].item())
    targets = torch.tensor([(total * r) // world for r in range(1, world)], dtype=torch.long, device=ip.device)
    cuts = torch.searchsorted(ip, targets, right=False).clamp(0, n).tolist() if world > 1 else []
    bounds = [0] + cuts + [n]
    for i in range(1, len(bounds)):              # monotone even when a hub row swallows several targets
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def shard_rows(indptr: torch.Tensor, src: torch.Tensor, r0: int, r1: int, **graph_kw) -> DGNGraph:
    """Rows [r0, r1) of a destination-major CSR as a bipartite shard: sources keep their global ids (the node
    features stay whole / replicated, ``x_src`` has N rows), the per-row arrays (``x_in``, ``x_dst``, output) are
    the shard's, and ``row_base = r0`` tells ``dgn_edge_weights`` where the shard's rows sit in ``eig``."""
    n = indptr.numel() - 1
    e0, e1 = int(indptr[r0].item()), int(indptr[r1].item())
    return DGNGraph.from_csr((indptr[r0:r1 + 1] - indptr[r0]), src[e0:e1], num_src=n, row_base=r0, **graph_kw)


class _AllGatherRows(torch.autograd.Function):
    """Row all-gather with its adjoint: the gradient of a rank's own rows is the SUM over ranks of the gradients of
    those rows of the gathered tensor (every rank consumed them), i.e. all-reduce + slice (= reduce-scatter; the
    all-reduce form also runs on gloo and with uneven shards)."""

    @staticmethod
    def forward(ctx, local, ranges, rank):
        longest = max(b - a for a, b in ranges)
        pad = local.new_zeros((longest,) + tuple(local.shape[1:]))
        pad[:local.shape[0]] = local
        parts = [torch.empty_like(pad) for _ in ranges]
        dist.all_gather(parts, pad)
        ctx.ranges, ctx.rank = ranges, rank
        return torch.cat([p[:b - a] for p, (a, b) in zip(parts, ranges)], dim=0)

    @staticmethod
    def backward(ctx, g_full):
        g_full = g_full.contiguous().clone()
        dist.all_reduce(g_full, op=dist.ReduceOp.SUM)
        a0 = ctx.ranges[0][0]
        a, b = ctx.ranges[ctx.rank]
        return g_full[a - a0:b - a0], None, None


def all_gather_rows(local: torch.Tensor, ranges: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """[rows_r, F] per rank -> [N, F] on every rank (the exchange a multi-layer net needs between two sharded layers:
    after the post-transformation the rows are F wide, 5.1 GB for 10 M x 128 fp32).  Uneven shards are padded to the
    largest one for the collective.  Differentiable: the backward sums the gathered tensor's gradient over the ranks
    and hands every rank the rows it owns.  Without an initialised process group it is the identity."""
    if not dist.is_initialized():
        return local
    if len(ranges) != dist.get_world_size():
        raise ValueError(f"{len(ranges)} row ranges for {dist.get_world_size()} ranks")
    return _AllGatherRows.apply(local, [tuple(r) for r in ranges], dist.get_rank())


class FlatGradAllReduce:
    """Average the gradients of ``params`` across ranks with one all-reduce on one flat buffer.

    EQUAL rank weights (sum / world), as DistributedDataParallel does: the result is the gradient of ``mean_r loss_r``.  With
    ``shard_by_edges`` the ranks hold different NUMBERS of graphs (the shards balance edges, not graphs), so for a loss that is a mean
    over the rank's graphs this is the global-batch gradient only when the shards hold equally many graphs; otherwise it is the
    "mean of per-shard means" -- the definition tests/test_dist_gloo.py pins (sequential shards, averaged).  A loop that wants the exact
    global-batch mean scales its local loss by ``n_local_graphs * world / n_global_graphs`` before ``backward()`` (one scalar)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()
        self.avg_in_collective = self.active and dist.get_backend() == "nccl" and os.environ.get("DGN_ALLREDUCE_AVG", "1") != "0"

    def __call__(self) -> None:
        """Three launches around the collective whatever the number of parameters: gather (foreach copy), scale, scatter."""
        if not self.active:
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        if self.avg_in_collective:
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)        # (RCCL averages inside the collective: one launch less)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)
        for p, g in zip(self.params, grads):
            if p.grad is None:
                p.grad = g
        torch._foreach_copy_(grads, self.views)


# ---- SyncBN (optional): BatchNorm over the union of every rank's rows ---------------------------------------------------
# Plain data parallelism normalises each shard with its own statistics (what FlatGradAllReduce's parity test pins).  With the modules
# below, a step over W shards is the step of ONE process over the whole batch: per BatchNorm one all-reduce of [2 F + 1] sums forward
# (sum, sum of squares, row count: shards are uneven) and one of [2 F] backward (sum g, sum g xhat), torch.nn.SyncBatchNorm's scheme on
# any backend (gloo on CPU included, which torch's own module refuses).  The layer's fused tail / whole-layer / graph-block calls compute
# their statistics in-kernel over the local rows, so layers holding these modules take the per-kernel route (ops.bn_tail_supported).

class _SyncBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, group):
        F_ = x.shape[1]
        # (fp64 sums: E[x^2] - mean^2 over the rows of all ranks cancels in fp32 once |mean| >> std; the library's own bn_stats
        #  accumulates in double as well)
        stats = torch.empty(2 * F_ + 1, dtype=torch.float64, device=x.device)
        stats[:F_], stats[F_:2 * F_], stats[2 * F_] = x.sum(0, dtype=torch.float64), (x * x).sum(0, dtype=torch.float64), float(x.shape[0])
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        n64 = stats[2 * F_]
        mean64 = stats[:F_] / n64
        var64 = (stats[F_:2 * F_] / n64 - mean64 * mean64).clamp_min(0)      # biased, as F.batch_norm normalises
        n, mean, var = n64.to(x.dtype), mean64.to(x.dtype), var64.to(x.dtype)
        invstd = torch.rsqrt(var64 + eps).to(x.dtype)
        xhat = (x - mean) * invstd
        ctx.save_for_backward(xhat, gamma, invstd, n)
        ctx.group = group
        ctx.mark_non_differentiable(mean, var, n)
        return xhat * gamma + beta, mean, var, n

    @staticmethod
    def backward(ctx, g, _gm, _gv, _gn):
        xhat, gamma, invstd, n = ctx.saved_tensors
        F_ = g.shape[1]
        sums = torch.cat([g.sum(0, dtype=torch.float64), (g * xhat).sum(0, dtype=torch.float64)])
        g_gamma, g_beta = sums[F_:].to(g.dtype), sums[:F_].to(g.dtype)      # local sums: the gradient all-reduce averages them
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=ctx.group)
        sums = (sums / n.double()).to(g.dtype)
        g_x = (g - sums[:F_] - xhat * sums[F_:]) * (gamma * invstd)
        return g_x, g_gamma, g_beta, None, None


def sync_batch_norm(x, gamma, beta, running_mean, running_var, momentum, eps, group=None):
    """Training-mode BatchNorm of ``x [n_local, F]`` over the rows of ALL ranks; running statistics updated in place with the global
    mean and the unbiased global variance (torch's rule)."""
    y, mean, var, n = _SyncBN.apply(x, gamma, beta, float(eps), group)
    with torch.no_grad():
        running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
        running_var.mul_(1 - momentum).add_((var * (n / (n - 1).clamp_min(1))).to(running_var.dtype), alpha=momentum)
    return y


class SyncBatchNorm1d(torch.nn.BatchNorm1d):
    """``nn.BatchNorm1d`` with the same parameters, buffers and ``state_dict`` keys whose training-mode statistics span every rank."""
    dgn_sync = True
    process_group = None

    def forward(self, x):
        if not (self.training and dist.is_initialized() and dist.get_world_size(self.process_group) > 1 and self.affine
                and self.track_running_stats and self.momentum is not None):
            return super().forward(x)
        self.num_batches_tracked.add_(1)
        return sync_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps, self.process_group)


def convert_sync_batchnorm(module: torch.nn.Module, process_group=None) -> torch.nn.Module:
    """Replace every ``nn.BatchNorm1d`` under ``module`` (the layers' ``batchnorm_h``, the MLPs' ``b_norm``) by a ``SyncBatchNorm1d``
    that SHARES its parameters and buffers (optimizer state and ``state_dict`` stay valid).  Returns ``module``."""
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.BatchNorm1d) and not isinstance(child, SyncBatchNorm1d):
            new = SyncBatchNorm1d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            new._parameters, new._buffers = child._parameters, child._buffers
            new.training, new.process_group = child.training, process_group
            setattr(module, name, new)
        else:
            convert_sync_batchnorm(child, process_group)
    return module


def gather_rank_stats(values: Sequence[float], device) -> List[List[float]]:
    """Every rank's list of numbers on every rank (one all-gather of a small fp64 tensor): per-rank shard sizes and collective
    timings for bench.py's line.  Without a process group: this rank's values alone."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [t.tolist()]
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [q.tolist() for q in parts]


def barrier_max_ms(ms: float, device) -> float:
    """MAX over ranks of a per-rank duration."""
    if not dist.is_initialized():
        return ms
    t = torch.tensor([ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
