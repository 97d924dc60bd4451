"""Graph batch in the layout the HIP kernels sweep: CSR by destination.

The reference hands a (batched) DGLGraph to every layer and lets DGL 0.4 bucket the
destinations by in-degree on every call (realworld_benchmark/nets/dgn_layer.py:183-186;
``dgl.batch`` in data/molecules.py:229).  Here the batch is converted ONCE into

    indptr [N+1] int32, src [E] int32   slots of a destination in ascending edge id
    eid    [E]   int64                  CSR slot -> original edge id (for edge features)
    log_deg [N]  fp32                   (float) log((double)(in_degree + 1))  (scalers.py:13)
    hub rows / slices                   rows longer than ``hub_threshold`` are cut in slices

and the per-edge directional weights derived from ``eig`` are cached on the graph, so
that the L layers (and all towers) of a forward pass share them.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import functools

import torch

from . import _lib
from .spec import EPS, AggPlan, Channel

HUB_THRESHOLD = 2048
HUB_CHUNK = 1024
# True: a graph given as CUDA edge lists is prepared by dgn_graph_build* (a handful of kernels behind one C call each);
# False: the same arrays from ~40 torch ops (what CPU tensors -- the gloo tests -- always use).  Same results.
NATIVE_BUILD = True
DC_CLASSES, DC_UNIT = 32, 64      # include/dgn_hip.h: DGN_DC_CLASSES, DGN_DC_UNIT
GRAPH_BLOCK_MAX_ROWS = 512       # largest graph (nodes) of a batch for which the graph backward is attached (the C side checks the LDS per list / width)
BLOCK_MAX_GAP = 52         # largest graph of a batch (nodes) for which the block backward is tried (a wave's block is at most 56 rows: csrc/dgn_agg_block.hpp; the C side checks the LDS budget per F)
DEFERRED_STATS = True      # DGNGraph.rebuild: the batch's (max in-degree, hub rows) are checked at the next load instead of with a host sync


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class DGNGraph:
    def __init__(self, src: torch.Tensor, dst: torch.Tensor, num_nodes: int, eig: Optional[torch.Tensor] = None,
                 hub_threshold: int = HUB_THRESHOLD, hub_chunk: int = HUB_CHUNK):
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src/dst must be 1-D tensors of equal length")
        device = src.device
        E = src.numel()
        if src.is_cuda and NATIVE_BUILD:
            self._build_native(src, dst, int(num_nodes), hub_threshold, hub_chunk)
        else:
            dst64 = dst.long()
            # stable sort by destination keeps ascending edge id inside every row (DGL mailbox order)
            _, perm = torch.sort(dst64, stable=True)
            deg = torch.bincount(dst64, minlength=num_nodes)
            indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=device)
            indptr[1:] = torch.cumsum(deg, 0)
            self._init_csr(indptr, src.long()[perm], perm, num_nodes, E, deg, hub_threshold, hub_chunk)
        self.ndata: Dict[str, torch.Tensor] = {}
        self.edata: Dict[str, torch.Tensor] = {}
        if eig is not None:
            self.ndata["eig"] = eig

    @classmethod
    def from_csr(cls, indptr: torch.Tensor, src_csr: torch.Tensor, eid: Optional[torch.Tensor] = None,
                 eig: Optional[torch.Tensor] = None, hub_threshold: int = HUB_THRESHOLD,
                 hub_chunk: int = HUB_CHUNK, num_src: Optional[int] = None, row_base: int = 0) -> "DGNGraph":
        """Adopt an existing destination-major CSR (no sort).  ``num_src``: number of source nodes of a bipartite
        CSR (rows = e.g. the graphs of a batch, sources = its nodes); default: the same node set.  ``row_base``:
        global id of row 0 when the CSR is a destination-range shard of a larger graph (``dist.shard_rows``)."""
        self = cls.__new__(cls)
        n = indptr.numel() - 1
        deg = (indptr[1:] - indptr[:-1]).long()
        self._init_csr(indptr.long(), src_csr, eid, n, src_csr.numel(), deg, hub_threshold, hub_chunk, num_src)
        self.row_base = int(row_base)
        self._c.row_base = self.row_base
        self.ndata, self.edata = {}, {}
        if eig is not None:
            self.ndata["eig"] = eig
        return self

    # ---- a batch held at a fixed capacity (shape-bucketed HIP-graph replay) --------------------------------------------------
    @classmethod
    def padded(cls, n_cap: int, e_cap: int, device, eig_dim: int = 0) -> "DGNGraph":
        """An EMPTY graph with static device arrays for up to ``n_cap`` nodes and ``e_cap`` edges; ``rebuild`` fills it with a
        batch.  Every array keeps its address and its capacity shape for the life of the object, the C-side description says
        ``n_nodes = n_cap``, ``n_edges = e_cap`` (rows beyond the batch are isolated, slots beyond its edges are never pointed
        at), and ``n_valid`` (a device int64 scalar) tells BatchNorm how many rows are real: a HIP graph captured on this object
        is valid for every batch that fits.  Molecule-like batches only (no hub rows)."""
        self = cls.__new__(cls)
        dev = torch.device(device)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        self._pad = dict(n_cap=int(n_cap), e_cap=int(e_cap))
        indptr, src_csr = i32(n_cap + 1), i32(e_cap)
        self.dst_csr = i32(e_cap)
        eid = torch.zeros(e_cap, dtype=torch.int64, device=dev)
        deg = torch.zeros(n_cap, dtype=torch.int64, device=dev)
        log_deg = torch.zeros(n_cap, dtype=torch.float32, device=dev)
        self._stats = i32(4)
        self._init_csr(indptr, src_csr, eid, n_cap, e_cap, deg, HUB_THRESHOLD, HUB_CHUNK, log_deg=log_deg, max_in_degree=0, n_hub_hint=0)
        self.n_valid = torch.zeros(1, dtype=torch.int64, device=dev)
        self.csc_ptr, self.csc_pos, self._csc_order = i32(n_cap + 1), i32(e_cap), i32(e_cap)
        self._c.csc_ptr, self._c.csc_pos = self.csc_ptr.data_ptr(), self.csc_pos.data_ptr()
        self._csc_ready = True
        lib = _lib.load()
        self._pad["ws_bytes"] = lib.dgn_graph_build_workspace_bytes(n_cap, e_cap)
        self._pad["ws"] = torch.empty(self._pad["ws_bytes"], dtype=torch.uint8, device=dev)
        self.ndata, self.edata = {}, {}
        if eig_dim:
            self.ndata["eig"] = torch.zeros(n_cap, eig_dim, dtype=torch.float32, device=dev)
        self.batch_nodes = self.batch_edges = 0
        return self

    def rebuild(self, src: torch.Tensor, dst: torch.Tensor, num_nodes: int, eig: Optional[torch.Tensor] = None, graph_sizes=None) -> None:
        """Load a batch into a ``padded`` graph, in place (dgn_graph_build + dgn_graph_build_csc into the static arrays; no host
        sync: the batch's statistics are checked at the next load, see check_deferred).  Cached per-graph tables (edge weights, scaler tables) are dropped: a step function
        captured on this object must recompute them INSIDE the captured region (they then replay with every batch)."""
        lib = _lib.load()
        pad = self._pad
        # The previous batch's deferred statistics are looked at BEFORE anything is overwritten: an error then leaves the object in the
        # (complete) state of the batch it is about, not half-way into the next one.  Callers run check_deferred() after the LAST batch.
        self.check_deferred(final=False)
        if self.__dict__.get("_blk_static") is not None and graph_sizes is None:
            # (a step captured on this object replays the block kernels whatever Python thinks: a stale table would be silent garbage)
            raise ValueError("rebuild: this padded graph has a static block table (set_block_capacity): pass the batch's graph_sizes")
        E, N = src.numel(), int(num_nodes)
        if N > pad["n_cap"] or E > pad["e_cap"]:
            raise ValueError(f"batch ({N} nodes, {E} edges) exceeds the capacity ({pad['n_cap']}, {pad['e_cap']})")
        dev = self.device
        src64, dst64 = src.to(dev).long().contiguous(), dst.to(dev).long().contiguous()
        stream = _lib.stream_ptr(dev)
        n_cap = pad["n_cap"]
        _lib.check(lib.dgn_graph_build(n_cap, E, _ptr(src64), _ptr(dst64), self.indptr.data_ptr(), self.src.data_ptr(), self.dst_csr.data_ptr(),
                                       self.eid.data_ptr(), self.log_deg.data_ptr(), self.in_degree.data_ptr(), self._stats.data_ptr(),
                                       int(self.hub_threshold), pad["ws"].data_ptr(), pad["ws_bytes"], stream), "dgn_graph_build")
        _lib.check(lib.dgn_graph_build_csc(n_cap, E, self.src.data_ptr(), self.csc_ptr.data_ptr(), self.csc_pos.data_ptr(),
                                           self._csc_order.data_ptr(), pad["ws"].data_ptr(), pad["ws_bytes"], stream), "dgn_graph_build_csc")
        self.n_valid.fill_(N)
        if graph_sizes is not None and self.__dict__.get("_blk_static") is not None:
            self.load_block_sizes(graph_sizes)
        if E < pad["e_cap"]:
            # slots beyond the batch's edges: no row points at them, but per-edge tensors are e_cap rows long (to_slot_order gathers through
            # eid, the edge-feature Linear runs over all rows) -- keep the tail a valid, fixed gather of edge 0 instead of the previous batch's ids
            self.eid[E:].zero_()
        if eig is not None:
            buf = self.ndata["eig"]
            buf[:N].copy_(eig, non_blocking=True)
            buf[N:].zero_()
        # No host sync: (max in-degree, hub rows) of THIS batch go to pinned memory asynchronously and are looked at when the NEXT batch
        # is loaded (or by check_deferred()).  A padded graph carries no hub tables, so a batch with rows beyond hub_threshold is still
        # computed correctly -- by the row kernels, slowly -- and the error arrives one batch late (top of this function) instead of
        # stalling every load.
        if DEFERRED_STATS:
            if getattr(self, "_stats_host", None) is None:
                self._stats_host = torch.zeros(4, dtype=torch.int32).pin_memory()
                self._stats_event = torch.cuda.Event()
            self._stats_host.copy_(self._stats, non_blocking=True)
            self._stats_event.record(torch.cuda.current_stream(dev))
            self._stats_pending = True
            self.max_in_degree = 0                                                  # (unknown: no launch is skipped on its account)
        else:
            max_deg, n_hub = self._stats[:2].tolist()                               # one host sync per load
            if n_hub:
                raise _lib.DgnError("padded graphs take batches without hub rows (in-degree <= hub_threshold) only")
            self.max_in_degree = int(max_deg)
        self.batch_nodes, self.batch_edges = N, E
        self.invalidate_caches()

    def check_deferred(self, final: bool = True) -> None:
        """Look at the statistics of the batch loaded last (``rebuild`` without a host sync): raises if it had hub rows, or if a step on
        a static block table met a graph beyond its capacity.  ``final`` (the explicit call after the last batch): the block-overflow
        flag is fetched as every step so far left it (one host sync); ``rebuild`` only looks at the copy made one load earlier."""
        if self.__dict__.get("_blk_static") is not None:
            ent = self._blk_static
            self._check_block_overflow(ent)
            if final:
                ent["overflow_host"].copy_(ent["overflow"], non_blocking=True)
                ent["overflow_event"].record(torch.cuda.current_stream(self.device))
                ent["overflow_pending"] = True
                self._check_block_overflow(ent)
        if not getattr(self, "_stats_pending", False):
            return
        self._stats_event.synchronize()
        self._stats_pending = False
        max_deg, n_hub = int(self._stats_host[0]), int(self._stats_host[1])
        self.max_in_degree = max_deg
        if n_hub:
            raise _lib.DgnError("padded graphs take batches without hub rows (in-degree <= hub_threshold) only "
                                f"(the batch loaded before this call had {n_hub}, largest in-degree {max_deg})")

    def invalidate_caches(self) -> None:
        """Drop everything derived from the graph's content (edge weights, scaler tables, slot -> destination map)."""
        self._wcache.clear()
        for k in ("_scale_cache", "_dst_slots", "_eig_norm", "_slot_types", "_dc", "_dc_split", "_dc_scale", "_blk", "_blk_tables"):
            self.__dict__.pop(k, None)
        if hasattr(self, "_c"):
            self._c.blk_cut, self._c.blk_gap = None, 0      # (the cut tensor is gone with "_blk": never leave its address behind)
            self._c.gblk_desc, self._c.n_gblk, self._c.gblk_rows, self._c.dst_csr = None, 0, 0, None

    def _build_native(self, src, dst, num_nodes, hub_threshold, hub_chunk):
        """CSR by destination through dgn_graph_build: one C call, one read-back of (max in-degree, hub rows)."""
        lib = _lib.load()
        dev, E = src.device, src.numel()
        if num_nodes >= 2 ** 31 - 1 or E >= 2 ** 31 - 1:
            raise ValueError("graph exceeds the int32 CSR range")
        src64, dst64 = src.long().contiguous(), dst.long().contiguous()
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        indptr, src_csr, dst_csr = i32(num_nodes + 1), i32(E), i32(E)
        eid = torch.empty(E, dtype=torch.int64, device=dev)
        log_deg = torch.empty(num_nodes, dtype=torch.float32, device=dev)
        deg = torch.empty(num_nodes, dtype=torch.int64, device=dev)
        stats = i32(4)
        nbytes = lib.dgn_graph_build_workspace_bytes(num_nodes, E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stream = _lib.stream_ptr(dev)
        _lib.check(lib.dgn_graph_build(num_nodes, E, _ptr(src64), _ptr(dst64), indptr.data_ptr(), _ptr(src_csr), _ptr(dst_csr), _ptr(eid),
                                       log_deg.data_ptr(), deg.data_ptr(), stats.data_ptr(), int(hub_threshold), ws.data_ptr(), nbytes,
                                       stream), "dgn_graph_build")
        self.dst_csr, self._stats = dst_csr, stats
        max_deg, n_hub = stats[:2].tolist()                                      # the one host sync of a graph build
        self._init_csr(indptr, src_csr, eid, num_nodes, E, deg, hub_threshold, hub_chunk, log_deg=log_deg, max_in_degree=max_deg,
                       n_hub_hint=n_hub)

    def _init_csr(self, indptr, src_csr, eid, num_nodes, E, deg, hub_threshold, hub_chunk, num_src=None, log_deg=None,
                  max_in_degree=None, n_hub_hint=None):
        if num_nodes >= 2 ** 31 - 1 or E >= 2 ** 31 - 1 or (num_src or 0) >= 2 ** 31 - 1:
            raise ValueError("graph exceeds the int32 CSR range")
        device = indptr.device
        self.device = device
        self.num_nodes, self.num_edges = int(num_nodes), int(E)
        self.num_src = int(num_src) if num_src is not None else int(num_nodes)
        self.row_base = 0
        self.indptr = indptr.to(torch.int32).contiguous()
        self.src = src_csr.to(torch.int32).contiguous()
        self.eid = eid  # None = identity (messages already in slot order)
        self.in_degree = deg
        self.log_deg = log_deg if log_deg is not None else torch.log((deg + 1).double()).float().contiguous()
        self.hub_threshold, self.hub_chunk = int(hub_threshold), int(hub_chunk)
        if max_in_degree is None:
            max_in_degree = int(deg.max().item()) if self.num_nodes else 0       # the one host sync of a graph build
        self.max_in_degree = int(max_in_degree)
        has_hubs = self.max_in_degree > hub_threshold if n_hub_hint is None else n_hub_hint > 0
        hub_rows = torch.nonzero(deg > hub_threshold).flatten() if has_hubs else deg.new_empty(0)
        self.n_hub = int(hub_rows.numel())           # (a second sync only for graphs that do have hub rows)
        self._keep = []
        c = _lib.DgnGraph()
        c.n_nodes, c.n_edges = self.num_nodes, self.num_edges
        c.indptr, c.src = self.indptr.data_ptr(), self.src.data_ptr()
        c.hub_threshold, c.hub_chunk = self.hub_threshold, self.hub_chunk
        c.max_in_degree = self.max_in_degree
        c.n_src = self.num_src if self.num_src != self.num_nodes else 0
        c.n_hub, c.n_chunks = 0, 0
        if self.n_hub:
            n_sl = (deg[hub_rows] + hub_chunk - 1) // hub_chunk
            ptr = torch.zeros(self.n_hub + 1, dtype=torch.int64, device=device)
            ptr[1:] = torch.cumsum(n_sl, 0)
            chunk_hub = torch.repeat_interleave(torch.arange(self.n_hub, device=device), n_sl)
            hub_rows_i, ptr_i, chunk_hub_i = hub_rows.int().contiguous(), ptr.int().contiguous(), chunk_hub.int().contiguous()
            self._keep += [hub_rows_i, ptr_i, chunk_hub_i]
            c.n_hub, c.n_chunks = self.n_hub, int(chunk_hub.numel())
            c.hub_rows, c.hub_chunk_ptr, c.chunk_hub = hub_rows_i.data_ptr(), ptr_i.data_ptr(), chunk_hub_i.data_ptr()
        self.n_chunks = int(c.n_chunks)
        self._c = c
        self._wcache: Dict[tuple, torch.Tensor] = {}

    # ---- degree classes (dgn_dc_kernels.hpp): nodes stably sorted by in-degree, every class padded to DC_UNIT rows ------------------
    def degree_classes(self):
        """``None`` when an in-degree reaches ``DC_CLASSES`` (or the graph is a padded / bipartite one), else a dict with the virtual
        row space of the degree-class posttrans products: ``vperm`` int32 [64 n_units] (node of a virtual row, -1: padding),
        ``unit_class`` int32 [n_units], ``present`` int32 [32] (rows per class), ``rep`` int64 [32] (one node of each class, 0 where
        absent: its row of a per-node scaler table is the class's row).  Built once per graph (one host read-back), torch ops."""
        if "_dc" in self.__dict__:
            return self._dc
        dc = None
        N = self.num_nodes
        if (N > 0 and self.in_degree is not None and getattr(self, "_pad", None) is None
                and self.num_src == self.num_nodes and self.max_in_degree < DC_CLASSES):
            dev = self.device
            key = self.in_degree.long()
            skey, order = torch.sort(key, stable=True)
            counts = torch.zeros(DC_CLASSES, dtype=torch.int64, device=dev).scatter_add_(0, key, torch.ones_like(key))
            padded = (counts + DC_UNIT - 1) // DC_UNIT * DC_UNIT
            seg_end = torch.cumsum(padded, 0)
            seg_start, first = seg_end - padded, torch.cumsum(counts, 0) - counts
            n_units = int(seg_end[-1].item()) // DC_UNIT                      # the one host read-back
            pos = seg_start[skey] + (torch.arange(N, device=dev) - first[skey])
            vperm = torch.full((n_units * DC_UNIT,), -1, dtype=torch.int32, device=dev)
            vperm[pos] = order.int()
            uc = torch.searchsorted(seg_end, torch.arange(n_units, device=dev) * DC_UNIT, right=True)
            rep = torch.where(counts > 0, order[first.clamp(max=N - 1)], torch.zeros_like(first))
            dc = dict(n_units=n_units, vperm=vperm, unit_class=uc.int().contiguous(), present=counts.int().contiguous(), rep=rep)
        self._dc = dc
        return dc

    def degree_classes_split(self):
        """The virtual row space of ``degree_classes()`` for a graph WITH hub rows (power-law graphs: C5 has 6 % of its rows at an
        in-degree >= ``DC_CLASSES``): the rows below that in-degree are sorted into the class units, the hub rows are left OUT of it
        and returned as ``hub_rows`` (int64, ascending) -- their posttrans takes the folded product on the gathered rows
        (ops.dc_posttrans_split).  ``None`` for padded / bipartite graphs.  Built once per graph (two host read-backs), torch ops."""
        if "_dc_split" in self.__dict__:
            return self._dc_split
        dc = None
        N = self.num_nodes
        if N > 0 and self.in_degree is not None and getattr(self, "_pad", None) is None and self.num_src == self.num_nodes:
            dev = self.device
            key = self.in_degree.long().clamp(max=DC_CLASSES)               # (class DC_CLASSES: the hubs, sorted last, not placed)
            skey, order = torch.sort(key, stable=True)
            counts = torch.zeros(DC_CLASSES + 1, dtype=torch.int64, device=dev).scatter_add_(0, key, torch.ones_like(key))
            n_hub = int(counts[DC_CLASSES].item())
            counts = counts[:DC_CLASSES]
            n_low = N - n_hub
            padded = (counts + DC_UNIT - 1) // DC_UNIT * DC_UNIT
            seg_end = torch.cumsum(padded, 0)
            seg_start, first = seg_end - padded, torch.cumsum(counts, 0) - counts
            n_units = int(seg_end[-1].item()) // DC_UNIT
            lkey, lorder = skey[:n_low], order[:n_low]
            pos = seg_start[lkey] + (torch.arange(n_low, device=dev) - first[lkey])
            vperm = torch.full((max(n_units, 1) * DC_UNIT,), -1, dtype=torch.int32, device=dev)
            vperm[pos] = lorder.int()
            uc = torch.searchsorted(seg_end, torch.arange(n_units, device=dev) * DC_UNIT, right=True)
            rep = torch.where(counts > 0, order[first.clamp(max=N - 1)], torch.zeros_like(first))
            dc = dict(n_units=n_units, vperm=vperm, unit_class=uc.int().contiguous(), present=counts.int().contiguous(), rep=rep,
                      hub_rows=order[n_low:].contiguous())                   # (stable sort: ascending node ids)
        self._dc_split = dc
        return dc

    def n_hub_rows_dc(self) -> int:
        """Rows whose in-degree is outside the degree classes (>= DC_CLASSES): 0 for graphs ``degree_classes()`` takes whole."""
        dc = self.degree_classes_split()
        return 0 if dc is None else int(dc["hub_rows"].numel())

    def ensure_csc(self) -> None:
        """Transposed view for the atomic-free backward (built on first use, one extra sort per graph)."""
        if getattr(self, "_csc_ready", False):
            return
        E, dev = self.num_edges, self.device
        if self.src.is_cuda and NATIVE_BUILD and self.num_src == self.num_nodes and E > 0:
            self._csc_native()
            self._csc_ready = True
            return
        order = torch.sort(self.src.long(), stable=True)[1]                 # slots ordered by (source, slot)
        pos = torch.empty(E, dtype=torch.int64, device=dev)
        pos[order] = torch.arange(E, device=dev)
        out_deg = torch.bincount(self.src.long(), minlength=self.num_src) if E else torch.zeros(self.num_src, dtype=torch.int64, device=dev)
        ptr = torch.zeros(self.num_src + 1, dtype=torch.int64, device=dev)
        ptr[1:] = torch.cumsum(out_deg, 0)
        self.csc_ptr, self.csc_pos, self.csc_order = ptr.int().contiguous(), pos.int().contiguous(), order.int().contiguous()
        self._c.csc_ptr, self._c.csc_pos = self.csc_ptr.data_ptr(), self.csc_pos.data_ptr()
        self._csc_ready = True

    def _csc_native(self) -> None:
        """Transposed view through dgn_graph_build_csc (one C call, no read-back)."""
        lib = _lib.load()
        N, E, dev = self.num_nodes, self.num_edges, self.device
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        stream = _lib.stream_ptr(dev)
        nbytes = lib.dgn_graph_build_workspace_bytes(N, E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.csc_ptr, self.csc_pos, order = i32(N + 1), i32(E), i32(E)
        self.csc_order = order                                  # (rank -> slot: the graph backward walks a source's out-edges with it)
        _lib.check(lib.dgn_graph_build_csc(N, E, self.src.data_ptr(), self.csc_ptr.data_ptr(), self.csc_pos.data_ptr(), order.data_ptr(),
                                           ws.data_ptr(), nbytes, stream), "dgn_graph_build_csc")
        self._c.csc_ptr, self._c.csc_pos = self.csc_ptr.data_ptr(), self.csc_pos.data_ptr()

    def _closed_cuts(self):
        """(blk_cut int32 [N + 1], largest gap) of dgn_graph_build_cuts, cached: blk_cut[i] = the last CLOSED cut <= i -- a cut is closed
        when no edge crosses it (the graph boundaries of a dgl.batch).  Five kernels and ONE read-back (the largest gap)."""
        if "_blk" not in self.__dict__:
            lib = _lib.load()
            N, E, dev = self.num_nodes, self.num_edges, self.device
            dst_csr = getattr(self, "dst_csr", None)
            if dst_csr is None:
                deg = (self.indptr[1:] - self.indptr[:-1]).long()
                dst_csr = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int32), deg)
                self.dst_csr = dst_csr
            cut = torch.empty(N + 1, dtype=torch.int32, device=dev)
            gap = torch.zeros(1, dtype=torch.int32, device=dev)
            nbytes = lib.dgn_graph_build_workspace_bytes(N, E)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.dgn_graph_build_cuts(N, E, self.src.data_ptr(), dst_csr.data_ptr(), cut.data_ptr(), gap.data_ptr(),
                                                ws.data_ptr(), nbytes, _lib.stream_ptr(dev)), "dgn_graph_build_cuts")
            self._blk = (cut, int(gap.item()))                                    # the one read-back
        return self._blk

    # ---- block table of the graph-block layer (csrc/dgn_blk_layer.hip): whole graphs per workgroup -------------------------------------
    def block_table(self, graph_sizes=None, target_rows=None):
        """DgnBlockTable of this batch, or None (hub rows, a bipartite / sharded / padded CSR, no edges; inside a stream capture unless
        built before).  Blocks = runs of whole graphs: the closed cuts (``graph_sizes``: the batch's node counts when the caller knows
        them -- ``dgl.batch``'s ``batch_num_nodes`` --, else found by dgn_graph_build_cuts) greedily grouped so that a block has about
        N / 512 rows -- at the reference's batch of 128 graphs one graph per workgroup.  Built once per batch: a few small kernels and two
        read-backs (cuts, row pointers), on the loader's side of the step like the CSR itself.  Cached; the dict also carries
        ``max_rows`` / ``max_edges`` (what decides whether a block fits the LDS)."""
        if getattr(self, "_pad", None) is not None:
            # a padded graph: the static table (set_block_capacity + load_block_sizes per batch), or none
            return self.__dict__.get("_blk_static") if target_rows is None else None
        tables = self.__dict__.setdefault("_blk_tables", {})
        ent = tables.get(target_rows)
        if ent is not None:
            return ent or None
        if not (self.src.is_cuda and self.num_src == self.num_nodes and self.n_hub == 0 and getattr(self, "_pad", None) is None
                and self.row_base == 0 and self.num_edges > 0 and self.num_nodes > 0):
            tables[target_rows] = {}
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        import numpy as np
        N = self.num_nodes
        if graph_sizes is None:
            # dgl.batch's own bookkeeping when the caller left it on the graph (``g.batch_num_nodes``: list / tensor, or DGL >= 0.5's method)
            graph_sizes = getattr(self, "batch_num_nodes", None)
            if callable(graph_sizes):
                graph_sizes = graph_sizes()
        known = graph_sizes is not None
        if known:
            sz = torch.as_tensor(graph_sizes).cpu().numpy().astype(np.int64).reshape(-1)
            cuts = np.concatenate([[0], np.cumsum(sz)])
            if cuts[-1] != N:
                raise ValueError("block_table: graph_sizes do not add up to the node count")
        else:
            cut, _ = self._closed_cuts()
            cuts = torch.unique_consecutive(cut).cpu().numpy().astype(np.int64)
        target = max(1, -(-N // 512)) if target_rows is None else int(target_rows)      # (target_rows = 1: one graph per block)
        prev, bounds = 0, [0]
        for c in cuts[1:]:
            # the graph [prev, c) would take the running block [bounds[-1], prev) past the target: close the block first
            # (a block always holds at least one whole graph, however large)
            if c - bounds[-1] > target and prev != bounds[-1]:
                bounds.append(int(prev))
            prev = c
        if bounds[-1] != N:
            bounds.append(N)
        b = np.asarray(bounds, dtype=np.int64)
        max_rows = int((b[1:] - b[:-1]).max())
        if known and self.max_in_degree > 0:
            # no read-back at all: the kernels take a block's slot range from the row pointers (negative slot fields), and the LDS plan
            # is made for the bound rows x largest in-degree (molecules: 4 x 37 slots where the largest block has ~84)
            desc = np.stack([b[:-1], b[1:], np.full_like(b[:-1], -1), np.full_like(b[:-1], -1)], axis=1).astype(np.int32)
            max_edges = int(min(self.num_edges, max_rows * int(self.max_in_degree)))
        else:
            indptr = self.indptr.cpu().numpy().astype(np.int64)
            desc = np.stack([b[:-1], b[1:], indptr[b[:-1]], indptr[b[1:]]], axis=1).astype(np.int32)
            max_edges = int((desc[:, 3] - desc[:, 2]).max())
        t = torch.from_numpy(np.ascontiguousarray(desc)).to(self.device)
        st = _lib.DgnBlockTable(n_blocks=int(desc.shape[0]), max_rows=max_rows, max_edges=max_edges, desc=t.data_ptr())
        ent = dict(struct=st, desc=t, n_blocks=st.n_blocks, max_rows=st.max_rows, max_edges=st.max_edges)
        tables[target_rows] = ent
        return ent

    def set_block_capacity(self, g_cap: int, max_rows: int, max_edges: int) -> None:
        """Padded graphs: a STATIC block table of ``g_cap`` entries -- one graph of the batch per block -- so that a step captured on this
        object runs its layers on the graph-block route (csrc/dgn_blk_layer.hip) for every batch that fits.  ``max_rows`` / ``max_edges``:
        the largest graph (nodes, directed edges) any batch will hold: the kernels' LDS plan is made for them once.  ``load_block_sizes``
        writes a batch's table; a graph beyond the capacity is skipped by the kernels and reported by ``check_deferred`` (late, like the
        hub statistics: no host sync per load)."""
        if getattr(self, "_pad", None) is None:
            raise ValueError("set_block_capacity is for padded graphs (DGNGraph.padded); others build their table from the batch")
        dev, g_cap = self.device, int(g_cap)
        desc = torch.zeros(g_cap, 4, dtype=torch.int32, device=dev)
        st = _lib.DgnBlockTable(n_blocks=g_cap, max_rows=int(max_rows), max_edges=int(max_edges), desc=desc.data_ptr())
        self._blk_static = dict(struct=st, desc=desc, n_blocks=g_cap, max_rows=int(max_rows), max_edges=int(max_edges),
                                host=torch.zeros(g_cap, 4, dtype=torch.int32).pin_memory(), copied=torch.cuda.Event(),
                                overflow=torch.zeros(1, dtype=torch.int32, device=dev), overflow_host=torch.zeros(1, dtype=torch.int32).pin_memory(),
                                overflow_event=torch.cuda.Event(), overflow_pending=False)

    def load_block_sizes(self, sizes) -> None:
        """The node counts of the loaded batch's graphs (host list / tensor, in batch order) -> the static block table: rows from the
        sizes, slots left to the kernels (they read the row pointers).  One pinned-memory copy, no host sync."""
        ent = self.__dict__.get("_blk_static")
        if ent is None:
            raise ValueError("load_block_sizes: call set_block_capacity first")
        import numpy as np
        sz = np.asarray(torch.as_tensor(sizes).cpu().numpy() if not isinstance(sizes, np.ndarray) else sizes, dtype=np.int64).reshape(-1)
        G = int(sz.size)
        if G > ent["n_blocks"]:
            raise ValueError(f"{G} graphs exceed the block table's capacity ({ent['n_blocks']})")
        if G and int(sz.max()) > ent["max_rows"]:
            raise ValueError(f"a graph of {int(sz.max())} nodes exceeds the block capacity ({ent['max_rows']} rows)")
        self._check_block_overflow(ent)
        ent["copied"].synchronize()                       # (the previous batch's copy has left the pinned buffer long ago)
        host = ent["host"].numpy()
        host[:] = 0
        cuts = np.concatenate([[0], np.cumsum(sz)])
        host[:G, 0], host[:G, 1], host[:G, 2:] = cuts[:-1], cuts[1:], -1
        ent["desc"].copy_(ent["host"], non_blocking=True)
        ent["copied"].record(torch.cuda.current_stream(self.device))
        # the overflow flag as the steps on the PREVIOUS batch left it travels to the host now and is looked at one load later
        ent["overflow_host"].copy_(ent["overflow"], non_blocking=True)
        ent["overflow_event"].record(torch.cuda.current_stream(self.device))
        ent["overflow_pending"] = True

    def _check_block_overflow(self, ent) -> None:
        if ent.get("overflow_pending"):
            ent["overflow_event"].synchronize()
            ent["overflow_pending"] = False
            if int(ent["overflow_host"][0]):
                ent["overflow"].zero_()
                raise _lib.DgnError(f"a batch held a graph beyond the block capacity ({ent['max_rows']} rows / {ent['max_edges']} edges): its rows "
                                    "were skipped by the graph-block layer kernels (set_block_capacity with larger bounds)")

    # ---- block description for the LDS-accumulating backward (csrc/dgn_agg_block.hpp) -----------------------------------------------
    def ensure_blocks(self, enabled: bool = True) -> bool:
        """Attach (or detach) DgnGraph.blk_cut / blk_gap: the closed cuts of a batch of small graphs (dgn_graph_build_cuts, built on
        first use: five kernels and ONE read-back of the largest gap).  Graphs without usable cuts (hub rows, a bipartite CSR, a padded
        batch, more than 3 edges per node on average -- the four-rows-per-wave kernels do not run there --, a component of more than
        BLOCK_MAX_GAP nodes) never get them: the staged backward runs.  Returns whether attached."""
        ok = False
        # (the C side takes the block kernels from DGN_BLK_MIN_NODES nodes on -- default 131 072, below it one wave per graph under-fills
        #  the chip --: smaller batches do not pay for the cut build and its read-back either)
        min_nodes = int(_lib.options.blk_min_nodes)
        if enabled and self.src.is_cuda and self.num_src == self.num_nodes and self.n_hub == 0 and getattr(self, "_pad", None) is None \
                and self.num_nodes >= max(1, min_nodes) and 0 < self.num_edges <= 3 * self.num_nodes and self.row_base == 0:
            if "_blk" not in self.__dict__ and torch.cuda.is_current_stream_capturing():
                enabled = False        # (the build reads the largest gap back: not inside a capture -- the staged backward is captured instead)
            if enabled and "_blk" not in self.__dict__:
                self._closed_cuts()
            if enabled:
                cut, gap = self._blk
                ok = 0 < gap <= BLOCK_MAX_GAP
        if ok:
            self._c.blk_cut, self._c.blk_gap = cut.data_ptr(), gap
        else:
            self._c.blk_cut, self._c.blk_gap = None, 0
        self._ensure_graph_blocks(enabled)
        return ok

    def _ensure_graph_blocks(self, enabled: bool = True) -> bool:
        """Attach (or detach) DgnGraph.gblk_desc / csc_order / dst_csr: one block per graph for the graph backward of batches whose graphs
        are too large for a wave's LDS block (k-NN superpixels, SBM: more than 3 edges per node, graphs of up to GRAPH_BLOCK_MAX_ROWS nodes;
        csrc/dgn_agg_graph.hpp).  Built on first use (the closed cuts: a few kernels and two read-backs), never inside a capture."""
        ok = False
        if enabled and self.src.is_cuda and self.num_src == self.num_nodes and self.n_hub == 0 and getattr(self, "_pad", None) is None \
                and self.row_base == 0 and self.num_edges > 3 * self.num_nodes:
            t = self.block_table(target_rows=1)
            if t is not None and t["max_rows"] <= GRAPH_BLOCK_MAX_ROWS and getattr(self, "_csc_ready", False) and hasattr(self, "csc_order"):
                self._closed_cuts()                      # (dst_csr)
                self._c.gblk_desc, self._c.n_gblk, self._c.gblk_rows = t["desc"].data_ptr(), t["n_blocks"], t["max_rows"]
                self._c.csc_order, self._c.dst_csr = self.csc_order.data_ptr(), self.dst_csr.data_ptr()
                ok = True
        if not ok:
            self._c.gblk_desc, self._c.n_gblk, self._c.gblk_rows, self._c.csc_order, self._c.dst_csr = None, 0, 0, None, None
        return ok

    # ---- DGL-flavoured accessors used by the nets (duck typing) ----
    def number_of_nodes(self) -> int:
        return self.num_nodes

    def number_of_edges(self) -> int:
        return self.num_edges

    @property
    def c_graph(self) -> _lib.DgnGraph:
        return self._c

    def to_slot_order(self, per_edge: torch.Tensor) -> torch.Tensor:
        """[E, ...] in original edge-id order -> CSR slot order (differentiable)."""
        return per_edge if self.eid is None else per_edge.index_select(0, self.eid)

    # ---- per-edge directional weights (cached per eig tensor version) ----
    def _normalised_eig(self, eig: torch.Tensor) -> torch.Tensor:
        """``eig`` on this graph's device as contiguous fp32 -- converted ONCE per (source tensor, version): the
        reference keeps eig on the CPU (dgn_layer.py:157-159) and the nets hand the same tensor to every layer, so the
        H2D copy / cast must not be repeated per layer.  The source tensor is held by the cache entry (identity check,
        not an address), so a freed-and-reused allocation can never alias an entry."""
        ent = getattr(self, "_eig_norm", None)
        if ent is not None and ent[0] is eig and ent[1] == eig._version:
            return ent[2]
        e = eig
        if e.device != self.device:
            e = e.to(self.device)
        if e.dtype != torch.float32 or e.dim() != 2 or e.stride(-1) != 1:
            e = e.float().contiguous()
        self._eig_norm = (eig, eig._version, e)
        return e

    def edge_weights(self, plan: AggPlan, eig: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        if plan.n_channels == 0:
            return None
        eig = self._normalised_eig(self.ndata["eig"] if eig is None else eig)
        # the entry keeps a strong reference to the tensor it was computed from: while the entry lives, that tensor's
        # storage cannot be freed and handed to another eig, so (id, version) identifies the VALUES
        key = (id(eig), eig._version, tuple(eig.shape), plan.channels)
        ent = self._wcache.get(key)
        if ent is None or ent[0] is not eig:
            w = compute_edge_weights(self, plan.channels, eig=eig)
            if len(self._wcache) >= 8:
                self._wcache.clear()
            self._wcache[key] = ent = (eig, w)
        return ent[1]


@functools.lru_cache(maxsize=64)
def _channel_array(channels: Tuple[Channel, ...]):
    arr = (_lib.DgnChannel * len(channels))()
    for i, (kind, col, alpha) in enumerate(channels):
        arr[i].kind, arr[i].eig_col, arr[i].alpha, arr[i].eps = kind, col, alpha, EPS
    return arr


def compute_edge_weights(graph: DGNGraph, channels: Tuple[Channel, ...], eig: Optional[torch.Tensor] = None,
                         eig_s_edge: Optional[torch.Tensor] = None, eig_d_edge: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w [len(channels), E] fp32 in CSR slot order (dgn_edge_weights of the C ABI)."""
    lib = _lib.load()
    ref = eig if eig is not None else eig_s_edge
    if ref is None or not ref.is_cuda:
        raise _lib.DgnError("dgn_amd runs on the GPU only: eig must be a CUDA tensor")
    E = graph.num_edges
    w = torch.empty((len(channels), max(E, 1)), dtype=torch.float32, device=ref.device)
    ld = ref.stride(0) if ref.dim() == 2 and ref.shape[0] > 0 else ref.shape[-1]
    stream = _lib.stream_ptr(ref.device)
    for c0 in range(0, len(channels), _lib.DGN_MAX_CH):
        chunk = channels[c0:c0 + _lib.DGN_MAX_CH]
        for ch in chunk:
            if ch[1] >= ref.shape[-1]:
                raise IndexError(f"aggregator needs eig column {ch[1]} but eig has {ref.shape[-1]} columns")
        nbytes = lib.dgn_edge_weights_workspace_bytes(C.byref(graph.c_graph), len(chunk))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device) if nbytes else None
        rc = lib.dgn_edge_weights(C.byref(graph.c_graph), _ptr(eig), _ptr(eig_s_edge), _ptr(eig_d_edge), ld, len(chunk),
                                  _channel_array(chunk), w[c0:].data_ptr(), w.stride(0), _ptr(ws), nbytes, stream)
        _lib.check(rc, "dgn_edge_weights")
    return w


def as_dgn_graph(g, device: Optional[torch.device] = None) -> DGNGraph:
    """Accept a DGNGraph, or anything DGL-shaped (``edges()``/``all_edges()``, ``number_of_nodes()``,
    ``ndata['eig']``): the converted batch is cached on the object, per device.  ``device``: where the features
    live (the layers pass ``h.device``); default: the current CUDA device.  The cached conversion always carries the
    caller's CURRENT ``g.ndata['eig']`` (the reference's train loops reassign it per batch for the sign-flip /
    rotation augmentations, train_molecules_graph_regression.py:29-33)."""
    if isinstance(g, DGNGraph):
        return g
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    cached = getattr(g, "_dgn_graph", None)
    if cached is not None and cached.device == dev:
        cached.ndata["eig"] = g.ndata["eig"]
        return cached
    edges = g.all_edges(order="eid") if hasattr(g, "all_edges") else g.edges()
    src, dst = edges[0], edges[1]
    out = DGNGraph(src.to(dev), dst.to(dev), g.number_of_nodes(), eig=g.ndata["eig"])
    if getattr(g, "batch_num_nodes", None) is not None:
        out.batch_num_nodes = g.batch_num_nodes         # (dgl.batch's graph sizes: the block table is then built without a read-back)
    try:
        g._dgn_graph = out
    except Exception:
        pass
    return out
