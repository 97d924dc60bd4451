"""HIP-graph capture of a launch-bound training step.

At the reference's batch size (128 molecules, ~3 000 nodes) a DGN layer step is ~80 kernel launches around 0.3-0.4 ms
of GPU work: the host, not the GPU, sets the pace.  Every launch of this package goes to the caller's stream with no
host synchronisation (the C ABI's contract), so a whole step -- edge weights, forward, backward -- can be captured
once into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed with one launch.  Valid only while
the batch SHAPE is fixed: the graph freezes every kernel argument (pointers, sizes, the CSR of the batch).

Training batches differ in node and edge count from step to step.  ``PaddedBatch`` holds a batch at a fixed CAPACITY in static
device buffers (``DGNGraph.padded`` + ``rebuild``: the CSR, its transposed view, eig, a device scalar with the number of real
rows), so ONE captured graph per capacity bucket serves every batch that fits: rows beyond the batch are isolated zero rows that
the sweep, the Linears and the elementwise kernels process like any other, and BatchNorm -- the only place where they would
matter -- reads the valid-row count from the device (``ops.padded_rows``).  Per step the host then does: ``load`` (two C calls
that rebuild the graph in place + small copies) and one graph launch.  Contract for the captured step function: it reads its
inputs from the batch's buffers, starts with ``batch.graph.invalidate_caches()`` (so that edge weights and scaler tables are
recomputed inside the captured region) and the cotangent rows of the padding must be zero.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch


def capture(step: Callable[[], None], warmup: int = 3) -> torch.cuda.CUDAGraph:
    """Run ``step`` ``warmup`` times on a side stream (allocator pools, per-graph caches such as the csc view and the
    scaler tables), then capture one more call.  ``step`` must not synchronise and must leave ``.grad`` fields to the
    backward (set them to None before calling this: the captured backward then writes fresh gradient tensors from
    the graph's private pool on every replay instead of accumulating).  Do not keep autograd-attached outputs of
    EARLIER steps alive across the capture (keep ``y.detach()``): releasing such a graph inside the capture region
    crashes ``capture_end`` on this ROCm / PyTorch."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    return graph


class PaddedBatch:
    """Static buffers for one capacity bucket: the padded graph plus named per-node / per-edge tensors (features, graph norm,
    cotangent ...) whose rows beyond the loaded batch are kept at zero."""

    def __init__(self, n_cap: int, e_cap: int, device, eig_dim: int):
        from .graph import DGNGraph
        self.n_cap, self.e_cap, self.device = int(n_cap), int(e_cap), torch.device(device)
        self.graph = DGNGraph.padded(n_cap, e_cap, device, eig_dim=eig_dim)
        self.node: Dict[str, torch.Tensor] = {}
        self.edge: Dict[str, torch.Tensor] = {}

    def add_node_tensor(self, name: str, width: int, requires_grad: bool = False) -> torch.Tensor:
        t = torch.zeros(self.n_cap, width, dtype=torch.float32, device=self.device)
        self.node[name] = t.requires_grad_(requires_grad)
        return self.node[name]

    def add_edge_tensor(self, name: str, width: int, requires_grad: bool = False) -> torch.Tensor:
        t = torch.zeros(self.e_cap, width, dtype=torch.float32, device=self.device)
        self.edge[name] = t.requires_grad_(requires_grad)
        return self.edge[name]

    def fits(self, num_nodes: int, num_edges: int) -> bool:
        return num_nodes <= self.n_cap and num_edges <= self.e_cap

    @torch.no_grad()
    def load(self, src: torch.Tensor, dst: torch.Tensor, num_nodes: int, eig: Optional[torch.Tensor] = None, node: Optional[dict] = None,
             edge: Optional[dict] = None) -> None:
        """Rebuild the graph in place and copy the batch's tensors into the static buffers (rows beyond the batch zeroed)."""
        self.graph.rebuild(src, dst, num_nodes, eig)
        for name, val in (node or {}).items():
            buf = self.node[name]
            buf[:num_nodes].copy_(val, non_blocking=True)
            buf[num_nodes:].zero_()
        E = src.numel()
        for name, val in (edge or {}).items():
            buf = self.edge[name]
            buf[:E].copy_(val, non_blocking=True)
            buf[E:].zero_()


def bucket_capacity(num_nodes: int, num_edges: int, granularity: int = 256, headroom: float = 1.1) -> Tuple[int, int]:
    """Capacity bucket of a batch: sizes with ``headroom`` rounded up to multiples of ``granularity`` (one captured graph per
    distinct pair; batches of a data loader with a fixed number of graphs fall into very few buckets)."""
    up = lambda v: int(-(-int(v * headroom) // granularity) * granularity)
    return up(num_nodes), up(num_edges)
