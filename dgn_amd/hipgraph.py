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
    import gc
    gc.collect()                      # autograd graphs of earlier eager steps that only a reference cycle keeps alive (e.g. a graph
    torch.cuda.synchronize()          # object whose ndata holds the features computed ON it) must not be released inside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(side)
    gc.collect()
    graph = torch.cuda.CUDAGraph()
    was_enabled = gc.isenabled()
    gc.disable()                      # (no collection in the middle of the capture region either)
    try:
        with torch.cuda.graph(graph):
            step()
    finally:
        if was_enabled:
            gc.enable()
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
             edge: Optional[dict] = None, graph_sizes=None) -> None:
        """Rebuild the graph in place and copy the batch's tensors into the static buffers (rows beyond the batch zeroed)."""
        self.graph.rebuild(src, dst, num_nodes, eig, graph_sizes=graph_sizes)      # (graph_sizes: required once set_block_capacity was called)
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


def rewrap_parameters(module: torch.nn.Module) -> None:
    """Replace every Parameter of ``module`` by a NEW Parameter object on the same storage.  A parameter that took part in an eager
    autograd pass on the DEFAULT stream keeps a gradient-accumulation node bound to that stream for the life of the Parameter object,
    and a later capture of a backward pass through it crashes in ``capture_end`` (ROCm 7 / PyTorch 2.10: measured, tools-free repro
    in tests/test_net_gpu.py).  New Parameter objects start clean; values, ``state_dict`` and storage are untouched, but optimizers and
    other holders of the OLD Parameter objects must be re-created.  The layers' parameter-list caches are dropped too."""
    for m in module.modules():
        for k, q in list(m._parameters.items()):
            if q is not None:
                m._parameters[k] = torch.nn.Parameter(q.detach(), requires_grad=q.requires_grad)
        m.__dict__.pop("_plist", None)
        m.__dict__.pop("_opmap", None)


class CapturedNetStep:
    """One captured HIP graph for the whole training step of a net around the layers (``dgn_amd.nets.DGNNet``: embedding, L layers,
    readout, MLP, L1 loss, backward, Adam) at a fixed capacity: ``load(batch)`` writes the batch into static buffers (graph rebuilt in
    place, atoms / graph norm / targets / the readout's graph -> nodes CSR copied), ``step()`` is one graph launch.

    At the reference's batch size (128 molecules) the eager step is host-bound (~5 ms for ~1 ms of GPU work: per-parameter autograd and
    optimizer bookkeeping for the ~120 per-tower tensors of the reference's ``state_dict`` layout); the replay is GPU-bound.

    Padding: node rows beyond the batch are isolated (``PaddedBatch``); graph rows beyond the batch are empty, and the padding nodes
    are shared out among the last PAD_ROWS graph rows, whose loss terms are masked -- every slot of the readout CSR stays referenced, so its backward defines
    (zero) gradients for the padding rows.  The loss is the mean absolute error over the real graphs (``nets.DGNNet.loss``).

    An ``optimizer`` passed in must be capturable (``capturable=True`` for Adam-type optimizers; plain SGD is) and must have been built
    on parameters that never took part in a default-stream autograd pass (``rewrap_parameters`` first).  Unless an ``optimizer`` is
    passed in, the net's Parameter objects are re-created on construction (``rewrap_parameters``: eager
    training steps on the default stream before a capture are otherwise fatal) and a capturable Adam is built on the new ones."""

    PAD_ROWS = 16      # readout rows behind the real graphs that share the padding nodes

    def __init__(self, net, n_cap: int, e_cap: int, g_cap: int, eig_dim: int, lr: float = 1e-3, optimizer=None, device=None,
                 max_graph_nodes: Optional[int] = None, max_graph_edges: Optional[int] = None):
        """``max_graph_nodes`` / ``max_graph_edges``: the largest graph (nodes, directed edges) of the dataset.  Given both, the padded graph
        gets a static block table (``DGNGraph.set_block_capacity``) and the captured step runs its layers on the graph-block route -- five
        launches per layer and step instead of ~28."""
        from .graph import DGNGraph
        dev = torch.device(device if device is not None else next(net.parameters()).device)
        if getattr(net, "edge_feat", False):
            raise ValueError("CapturedNetStep: nets with edge_feat=True are not supported (the captured step has no static bond-type "
                             "buffer); run them eagerly or build the net with edge_feat=False")
        # graph rows of the readout: g_cap - 1 real graphs at most, then PAD_ROWS rows that share the padding nodes (ONE padding row was a
        # single wave walking ~10 % of the batch's nodes in sequence: 40 us per direction of a 0.77 ms step)
        self.g_cap = int(g_cap)
        self.net, self.device = net, dev
        g_cap = self.g_rows = self.g_cap - 1 + self.PAD_ROWS
        self.pb = PaddedBatch(n_cap, e_cap, dev, eig_dim)
        if max_graph_nodes and max_graph_edges:
            self.pb.graph.set_block_capacity(g_cap, max_graph_nodes, max_graph_edges)
        self.atoms = torch.zeros(n_cap, dtype=torch.int64, device=dev)
        self.snorm = self.pb.add_node_tensor("snorm", 1)
        self.targets = torch.zeros(g_cap, 1, device=dev)
        self.gmask = torch.zeros(g_cap, 1, device=dev)
        self.n_graphs = torch.ones(1, device=dev)
        self.loss = torch.zeros((), device=dev)
        self._h_sizes = torch.zeros(g_cap, dtype=torch.int64).pin_memory()
        self._d_sizes = torch.zeros(g_cap, dtype=torch.int64, device=dev)
        self._g_ids = torch.arange(g_cap, device=dev).unsqueeze(1)
        # the readout's CSR: rows = graphs (the last one collects the padding nodes), slots = nodes in order
        # (built from an even split: no row may look like a hub row -- more than 2048 slots -- at construction, the hub description of
        #  a DGNGraph is static)
        indptr = (torch.arange(g_cap + 1, dtype=torch.int64, device=dev) * n_cap) // g_cap
        if int((indptr[1:] - indptr[:-1]).max()) > 2048:
            raise ValueError("capacity of more than 2048 nodes per graph row")
        rg = DGNGraph.from_csr(indptr, torch.arange(n_cap, dtype=torch.int32, device=dev), num_src=n_cap)
        assert rg.n_hub == 0
        i32 = lambda n: torch.arange(n, dtype=torch.int32, device=dev)
        rg.csc_ptr, rg.csc_pos = i32(n_cap + 1), i32(n_cap)                 # node j sits in slot j of exactly one row
        rg._c.csc_ptr, rg._c.csc_pos = rg.csc_ptr.data_ptr(), rg.csc_pos.data_ptr()
        rg._csc_ready = True
        rg.sizes = rg.in_degree
        self.rg = rg
        self.pb.graph._dgn_readout = rg
        if optimizer is None:
            rewrap_parameters(net)
            try:                   # one multi-tensor kernel for all ~120 parameter tensors (the per-tensor form is ~1 ms of tiny kernels per step)
                optimizer = torch.optim.Adam(net.parameters(), lr=lr, capturable=True, fused=True)
            except Exception:
                optimizer = torch.optim.Adam(net.parameters(), lr=lr, capturable=True)
        self.opt = optimizer
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    @torch.no_grad()
    def load(self, src, dst, num_nodes: int, eig, atoms, snorm, sizes, targets) -> None:
        """sizes: nodes per graph (host list or tensor), targets [n_graphs, 1]."""
        n_cap, g_cap, dev = self.pb.n_cap, self.g_rows, self.device
        sizes = torch.as_tensor(sizes, dtype=torch.int64)
        G = sizes.numel()
        if G >= self.g_cap:
            raise ValueError(f"{G} graphs need a capacity of at least {G + 1} graph rows (rows behind the batch collect the padding)")
        pad, R = n_cap - int(num_nodes), self.PAD_ROWS
        if pad > 2048 * R:
            raise ValueError(f"more than {2048 * R} padding nodes: pick a smaller capacity bucket (a padding row must not be a hub row)")
        self.pb.load(src, dst, num_nodes, eig, node={"snorm": snorm}, graph_sizes=sizes)
        self.atoms[:num_nodes].copy_(atoms, non_blocking=True)
        self.atoms[num_nodes:].zero_()
        # graph sizes -> the readout CSR, on the device: ONE copy from a pinned staging buffer (a pageable host tensor copied
        # "non_blocking" right in front of a graph launch stalled the launch by ~20 ms on this runtime), everything else device ops
        h = self._h_sizes
        h.zero_()
        h[:G] = sizes
        h[g_cap - R:] = pad // R                                  # the padding rows
        h[g_cap - R:g_cap - R + pad % R] += 1
        self._d_sizes.copy_(h, non_blocking=True)
        rg = self.rg
        rg.indptr[1:].copy_(torch.cumsum(self._d_sizes, 0))
        rg.in_degree.copy_(self._d_sizes)
        rg.log_deg.copy_(torch.log((self._d_sizes + 1).double()))
        self.targets.zero_()
        self.targets[:G].copy_(targets, non_blocking=True)
        self.gmask.copy_((self._g_ids < G).to(self.gmask.dtype))
        self.n_graphs.fill_(float(G))

    def _step(self) -> None:
        self.opt.zero_grad(set_to_none=True)
        self.pb.graph.invalidate_caches()
        scores = self.net(self.pb.graph, self.atoms, None, self.snorm, None)
        loss = ((scores - self.targets).abs() * self.gmask).sum() / self.n_graphs
        loss.backward()
        self.opt.step()
        self.loss.copy_(loss.detach().reshape(()))
        # nothing attached to this step's autograd graph may outlive the step (the net parks the node features on the graph object:
        # released inside the NEXT capture region it would crash capture_end, see capture())
        self.pb.graph.ndata.pop("h", None)
        del scores, loss

    def capture(self, warmup: int = 3) -> None:
        """Capture the step on the batch currently loaded (the warm-up steps DO update the parameters)."""
        self.graph = capture(self._step, warmup=warmup)

    def step(self) -> torch.Tensor:
        """One training step on the loaded batch; returns the (device) loss of that step."""
        if self.graph is None:
            self._step()
        else:
            self.graph.replay()
        return self.loss
