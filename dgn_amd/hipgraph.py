"""HIP-graph capture of a launch-bound training step.

At the reference's batch size (128 molecules, ~3 000 nodes) a DGN layer step is ~80 kernel launches around 0.3-0.4 ms
of GPU work: the host, not the GPU, sets the pace.  Every launch of this package goes to the caller's stream with no
host synchronisation (the C ABI's contract), so a whole step -- edge weights, forward, backward -- can be captured
once into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed with one launch.  Valid only while
the batch SHAPE is fixed: the graph freezes every kernel argument (pointers, sizes, the CSR of the batch).
"""
from __future__ import annotations

from typing import Callable

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
