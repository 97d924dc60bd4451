"""``directional_aggregate``: the fused aggregation as a torch.autograd.Function over the C ABI.

Replaces ``g.apply_edges(...)`` + ``g.update_all(message_func, reduce_func)`` of the
reference layers (realworld_benchmark/nets/dgn_layer.py:183-186, :112-115, :261-264).
GPU only; no CPU path exists in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from .graph import DGNGraph, _ptr
from .spec import EPS, AggPlan


# Backward scatter of d x_src: True = two-phase, atomic-free, bitwise reproducible (needs an [E, F] staging buffer
# and the graph's transposed view, built on first use); False = hardware fp32 atomics; "auto" = two-phase when the
# sweep can use >= 8-byte lanes (even F: measured 8-14 % faster on the molecule configs), atomics otherwise (odd F,
# e.g. hidden 75: 4-byte staging rows make the two-phase path 40 % slower than atomics).
DETERMINISTIC_BACKWARD = "auto"
# Block backward (csrc/dgn_agg_block.hpp): on batches of small graphs (molecules) one wave owns whole graphs and accumulates d x_src in
# its own LDS rows -- one kernel, no [E, F] staging round trip (1.2-1.7x the algorithmic traffic on the measured configs), the adds in
# the staged path's own order (run-to-run reproducible).  True (default): used wherever the graph and the aggregator list have such a
# kernel (the C side decides per launch); False: always the staged two-phase scatter.
BLOCK_BACKWARD = os.environ.get("DGN_BLOCK_BACKWARD", "1") != "0"

# ---- padded batches -------------------------------------------------------------------------------------------------------------
# A batch held at a fixed row capacity (shape-bucketed HIP-graph replay: dgn_amd/hipgraph.py::PaddedBatch) carries a DEVICE scalar
# ``n_valid``: rows >= n_valid are padding (zero features, no edges).  The sweep, the Linears and the elementwise kernels treat them
# as ordinary isolated rows; only BatchNorm must know (statistics over the valid rows, zero gradient for the others).  The layers
# announce the scalar for the duration of their forward; every BatchNorm node picks it up and keeps it for its backward.
_N_VALID = None


class padded_rows:
    def __init__(self, n_valid):
        self.n_valid = n_valid

    def __enter__(self):
        global _N_VALID
        self.prev, _N_VALID = _N_VALID, self.n_valid
        return self

    def __exit__(self, *exc):
        global _N_VALID
        _N_VALID = self.prev
        return False


def _spec_structs(plan: AggPlan, n_towers: int, avg_log: float, tower_stride: int = 0):
    # (tower_stride = N * K of the batch at hand: keying the cache on it would add a permanent entry per distinct batch node count over
    # a training run, and patching the cached structs would alias two callers holding them at once -- so the cache keeps stride-less
    # templates and every call gets its own ~200-byte copies.)
    key = (n_towers, float(avg_log))
    cache = plan.__dict__.setdefault("_spec_cache", {})
    if key not in cache:
        specs = []
        for l in plan.launches:
            s = _lib.DgnAggSpec()
            s.n_agg = len(l.ops)
            for i, (op, ch) in enumerate(zip(l.ops, l.chs)):
                s.agg_op[i], s.agg_ch[i] = op, ch
            s.n_ch = len(l.channels)
            s.n_scalers = plan.n_scalers
            for i, k in enumerate(plan.applied_scalers):
                s.scaler[i] = k
            s.avg_log, s.eps, s.n_towers = float(avg_log), EPS, n_towers
            s.agg_total, s.agg_offset = plan.n_agg, l.agg_offset
            s.tower_stride = 0
            specs.append(s)
        cache[key] = specs
    out = []
    for t in cache[key]:
        s = _lib.DgnAggSpec.from_buffer_copy(t)
        s.tower_stride = int(tower_stride)
        out.append(s)
    return out


def _check(t: Optional[torch.Tensor], name: str, rows: int, F: int):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.DgnError(f"{name} must be a CUDA tensor: dgn_amd has no CPU path")
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[0] != rows or t.shape[1] != F or (F > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected fp32 [{rows}, {F}] with unit inner stride, got {tuple(t.shape)} {t.dtype} "
                         f"strides {t.stride()}")


def _ld(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else (t.stride(0) if t.shape[0] > 1 else t.shape[1])


MAX_EDGE_TABLE = 8192      # include/dgn_hip.h: DGN_MAX_EDGE_TABLE (floats of an edge-type table)


def _msg_struct(F, x_src, x_dst, m_edge, x_in, edge_type=None, f_valid=0):
    m = _lib.DgnMsg()
    m.F = F
    m.f_valid = int(f_valid)
    m.x_src, m.ld_src = _ptr(x_src), _ld(x_src)
    m.x_dst, m.ld_dst = _ptr(x_dst), _ld(x_dst)
    m.m_edge, m.ld_edge = _ptr(m_edge), _ld(m_edge)
    m.x_in, m.ld_in = _ptr(x_in), _ld(x_in)
    if edge_type is not None:
        m.edge_type, m.n_edge_types = edge_type.data_ptr(), m_edge.shape[0]
    return m


def _out_layout(t: torch.Tensor):
    """(tower_stride, row stride) of an output / upstream-gradient tensor: [N, W] node-major (tower blocks inside
    the row, tower_stride 0 = default) or [T, N, K] tower-major."""
    if t.dim() == 3:
        return t.stride(0), t.stride(1)
    return 0, t.stride(0)


def agg_aux_bytes(graph: DGNGraph, plan: AggPlan, n_towers: int, F: int, x_src, x_dst, m_edge, x_in, edge_type=None, f_valid: int = 0) -> int:
    """Bytes of the aux table of a forward / backward pair over this message (0: none; see dgn_agg_forward_aux in include/dgn_hip.h)."""
    if len(plan.launches) != 1:
        return 0
    lib = _lib.load()
    spec = _spec_structs(plan, n_towers, 1.0, 0)[0]
    msg = _msg_struct(F, x_src, x_dst, m_edge, x_in, edge_type, f_valid)
    g = graph.c_graph
    return int(lib.dgn_agg_aux_bytes(C.byref(g), C.byref(spec), C.byref(msg)))


def odd_direct_supported(graph: DGNGraph, plan: AggPlan) -> bool:
    """Whether the simple layer at an ODD hidden size runs the sweep on the un-padded rows (DgnMsg.f_valid, the `Cfg::ODD` kernels):
    csrc/dgn_layers.hip odd_direct() -- no hub rows, the library option on, a list with odd-width kernels."""
    if graph.n_hub or len(plan.launches) != 1 or not _lib.options.odd_direct:
        return False
    spec = _spec_structs(plan, 1, 1.0, 0)[0]
    return bool(_lib.load().dgn_agg_f_valid_supported(C.byref(spec)))


def launch_forward(graph: DGNGraph, plan: AggPlan, n_towers: int, avg_log: float, w, x_src, x_dst, m_edge, x_in, out, edge_type=None, aux=None,
                   f_valid: int = 0):
    """Enqueue dgn_agg_forward (one call per launch group of the plan) on the current stream.  ``edge_type`` (int32 [E], CSR slot
    order): ``m_edge`` is a [K, F] table and slot j adds row ``edge_type[j]``.  ``f_valid`` (DgnMsg.f_valid): x_src / x_in hold rows of
    that ODD width, the sweep runs at F = f_valid + 1 (what the simple layer launches at an odd hidden size)."""
    lib = _lib.load()
    ref = x_src if x_src is not None else (x_dst if x_dst is not None else m_edge)
    F = f_valid + 1 if f_valid else ref.shape[1]
    stream = _lib.stream_ptr(ref.device)
    tower_stride, ld_out = _out_layout(out)
    specs = _spec_structs(plan, n_towers, avg_log, tower_stride)
    msg = _msg_struct(F, x_src, x_dst, m_edge, x_in, edge_type, f_valid)
    g = graph.c_graph
    for spec, l in zip(specs, plan.launches):
        nbytes = lib.dgn_agg_workspace_bytes(C.byref(g), C.byref(spec), F) if graph.n_hub else 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device) if nbytes else None
        wl = w[l.ch_offset:] if (w is not None and l.channels) else None
        rc = lib.dgn_agg_forward_aux(C.byref(g), C.byref(spec), C.byref(msg), _ptr(wl), w.stride(0) if w is not None else 0,
                                     graph.log_deg.data_ptr(), out.data_ptr(), ld_out, _ptr(aux), _ptr(ws), nbytes, stream)
        _lib.check(rc, "dgn_agg_forward")


def launch_backward(graph: DGNGraph, plan: AggPlan, n_towers: int, avg_log: float, w, x_src, x_dst, m_edge, x_in, g_out,
                    g_src, g_dst, g_edge, g_in, accumulate: bool = True, edge_type=None, aux=None, f_valid: int = 0):
    """Enqueue dgn_agg_backward.  ``accumulate=False``: the sinks g_src/g_dst/g_in may be uninitialised, the first
    launch of the plan defines them and later launches add; ``True``: every launch adds.  g_edge is overwritten.
    ``f_valid``: as launch_forward (the gradient sinks are F = f_valid + 1 wide)."""
    lib = _lib.load()
    ref = x_src if x_src is not None else (x_dst if x_dst is not None else m_edge)
    F = f_valid + 1 if f_valid else ref.shape[1]
    dev = g_out.device
    grads = _lib.DgnMsgGrad()
    grads.g_src, grads.ld_src = _ptr(g_src), _ld(g_src)
    grads.g_dst, grads.ld_dst = _ptr(g_dst), _ld(g_dst)
    grads.g_edge, grads.ld_edge = _ptr(g_edge), _ld(g_edge)
    grads.g_in, grads.ld_in = _ptr(g_in), _ld(g_in)
    msg = _msg_struct(F, x_src, x_dst, m_edge, x_in, edge_type, f_valid)
    stream = _lib.stream_ptr(dev)
    tower_stride, ld_gout = _out_layout(g_out)
    specs = _spec_structs(plan, n_towers, avg_log, tower_stride)
    g = graph.c_graph
    first = True
    deterministic = (F % 2 == 0) if DETERMINISTIC_BACKWARD == "auto" else bool(DETERMINISTIC_BACKWARD)
    deterministic = deterministic and g_src is not None
    if edge_type is not None and not deterministic:
        raise _lib.DgnError("edge-type table: the backward needs the two-phase scatter (even F, a gradient for x_src)")
    if deterministic:
        graph.ensure_csc()
    graph.ensure_blocks(bool(BLOCK_BACKWARD) and deterministic and not accumulate and len(plan.launches) == 1)
    for spec, l in zip(specs, plan.launches):
        nbytes = lib.dgn_agg_backward_workspace_bytes(C.byref(g), C.byref(spec), F, 1 if deterministic else 0)
        if edge_type is not None:
            nbytes += lib.dgn_agg_edge_table_workspace_bytes(F, m_edge.shape[0])
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
        wl = w[l.ch_offset:] if (w is not None and l.channels) else None
        tmp = None
        if g_edge is not None and not first:
            # g_edge is overwritten by every launch: accumulate the slices on the host side
            tmp = torch.empty_like(g_edge)
            grads.g_edge = tmp.data_ptr()
        grads.accumulate = 1 if (accumulate or not first) else 0
        rc = lib.dgn_agg_backward_aux(C.byref(g), C.byref(spec), C.byref(msg), _ptr(wl), w.stride(0) if w is not None else 0,
                                      graph.log_deg.data_ptr(), g_out.data_ptr(), ld_gout, _ptr(aux), C.byref(grads),
                                      _ptr(ws), nbytes, stream)
        _lib.check(rc, "dgn_agg_backward")
        if tmp is not None:
            g_edge += tmp
            grads.g_edge = g_edge.data_ptr()
        first = False


def _empty_rows(x: torch.Tensor) -> torch.Tensor:
    """Uninitialised contiguous [N, F] gradient buffer for x (which may be a strided view)."""
    return torch.empty(x.shape, dtype=x.dtype, device=x.device)


class _DirectionalAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph: DGNGraph, plan: AggPlan, n_towers: int, avg_log: float, w, x_src, x_dst, m_edge, x_in,
                xin_is_src: bool, tower_major: bool = False, x_pair=None, edge_type=None):
        lib = _lib.load()
        if x_pair is not None:       # [N, 2F] = x_src | x_dst in one tensor: one gradient tensor comes back
            half = x_pair.shape[1] // 2
            x_src, x_dst = x_pair[:, :half], x_pair[:, half:]
        ref = x_src if x_src is not None else (x_dst if x_dst is not None else m_edge)
        if ref is None:
            raise ValueError("the message needs at least one of x_src / x_dst / m_edge")
        F = ref.shape[1]
        N, E = graph.num_nodes, graph.num_edges
        _check(x_src, "x_src", graph.num_src, F)
        _check(x_dst, "x_dst", N, F)
        if edge_type is not None:
            if m_edge is None or x_src is None:
                raise ValueError("edge_type needs the table (m_edge [K, F]) and x_src")
            if hasattr(graph, "_pad"):
                raise _lib.DgnError("edge-type table on a padded graph: the table-gradient reduction runs over all e_cap slots; "
                                    "pass the gathered rows (EdgeTypeFeatures does so by itself on padded graphs)")
            if edge_type.dtype != torch.int32 or edge_type.shape != (E,) or not edge_type.is_contiguous() or not edge_type.is_cuda:
                raise ValueError(f"edge_type: expected a contiguous CUDA int32 [{E}] in CSR slot order")
            if m_edge.shape[0] * F > MAX_EDGE_TABLE:
                raise ValueError(f"edge-type table of {m_edge.shape[0]} x {F} floats > {MAX_EDGE_TABLE}: pass the gathered rows instead")
            _check(m_edge, "m_edge (edge-type table)", m_edge.shape[0], F)
        else:
            _check(m_edge, "m_edge", E, F)
        if xin_is_src:
            x_in = x_src
        _check(x_in, "x_in", N, F)
        if plan.needs_x_in() and x_in is None:
            raise ValueError("dx aggregators need x_in (h_in of reduce_func)")
        if ref.device != graph.device:
            raise ValueError(f"features on {ref.device} but graph on {graph.device}")
        if tower_major:
            out = torch.empty((n_towers, N, plan.out_width(F) // n_towers), dtype=torch.float32, device=ref.device)
        else:
            out = torch.empty((N, plan.out_width(F)), dtype=torch.float32, device=ref.device)
        # training: the forward leaves what the backward would recompute from the messages in a byte table (dgn_agg_forward_aux)
        aux = None
        if AGG_AUX and any(ctx.needs_input_grad):
            n_aux = agg_aux_bytes(graph, plan, n_towers, F, x_src, x_dst, m_edge, x_in, edge_type)
            aux = torch.empty(n_aux, dtype=torch.uint8, device=ref.device) if n_aux else None
        launch_forward(graph, plan, n_towers, avg_log, w, x_src, x_dst, m_edge, x_in, out, edge_type, aux=aux)
        ctx.edge_type, ctx.aux = edge_type, aux
        ctx.graph, ctx.plan, ctx.n_towers, ctx.avg_log, ctx.xin_is_src, ctx.F = graph, plan, n_towers, avg_log, xin_is_src, F
        ctx.paired = x_pair is not None
        if ctx.paired:
            ctx.save_for_backward(w, x_pair, None, m_edge, None if xin_is_src else x_in)
        else:
            ctx.save_for_backward(w, x_src, x_dst, m_edge, None if xin_is_src else x_in)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        graph, plan, F = ctx.graph, ctx.plan, ctx.F
        w, x_src, x_dst, m_edge, x_in = ctx.saved_tensors
        g_pair = None
        if ctx.paired:
            x_pair, half = x_src, x_src.shape[1] // 2
            x_src, x_dst = x_pair[:, :half], x_pair[:, half:]
        if ctx.xin_is_src:
            x_in = x_src
        g_out = g_out.contiguous()
        need_src, need_dst, need_edge, need_in = ctx.needs_input_grad[5:9]
        dev = g_out.device
        if ctx.paired:
            # one [N, 2F] gradient for P | Q (views into it go to the kernel): no slice-backward zero-fill + copy + add
            g_pair = torch.empty_like(x_pair) if ctx.needs_input_grad[11] else None
            g_src = g_pair[:, :half] if g_pair is not None else None
            g_dst = g_pair[:, half:] if g_pair is not None else None
            need_src = need_dst = False
        else:
            g_src = _empty_rows(x_src) if (x_src is not None and (need_src or ctx.xin_is_src)) else None
            g_dst = _empty_rows(x_dst) if (x_dst is not None and need_dst) else None
        # (a padded graph has e_cap slots of which the sweep touches the batch's: the untouched gradient rows must read as zero)
        g_edge = (torch.zeros_like(m_edge) if hasattr(graph, "_pad") else torch.empty_like(m_edge)) if (m_edge is not None and need_edge) else None
        if ctx.xin_is_src:
            g_in = g_src
        else:
            g_in = _empty_rows(x_in) if (x_in is not None and need_in and plan.needs_x_in()) else None
        # the sinks are DEFINED by the call (accumulate = 0): no zero-fill, every row is written exactly once
        if ctx.edge_type is not None and g_src is None:
            g_src = _empty_rows(x_src)               # (the table's gradient is a reduction of the staged per-edge rows)
        launch_backward(graph, plan, ctx.n_towers, ctx.avg_log, w, x_src, x_dst, m_edge, x_in, g_out, g_src, g_dst, g_edge, g_in,
                        accumulate=False, edge_type=ctx.edge_type, aux=ctx.aux)
        if x_in is not None and not ctx.xin_is_src and need_in and g_in is None:
            g_in = torch.zeros_like(x_in)
        if ctx.paired:
            return (None, None, None, None, None, None, None, g_edge, None if ctx.xin_is_src else g_in, None, None, g_pair, None)
        return (None, None, None, None, None, g_src if (need_src or ctx.xin_is_src) else None, g_dst, g_edge,
                None if ctx.xin_is_src else g_in, None, None, None, None)


def directional_aggregate(graph: DGNGraph, plan: AggPlan, avg_log, x_src: Optional[torch.Tensor] = None,
                          x_dst: Optional[torch.Tensor] = None, m_edge: Optional[torch.Tensor] = None,
                          x_in: Optional[torch.Tensor] = None, eig: Optional[torch.Tensor] = None,
                          n_towers: int = 1, weights: Optional[torch.Tensor] = None, tower_major: bool = False,
                          x_pair: Optional[torch.Tensor] = None, edge_type: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out [N, T*S*A*(F/T)] (or, with ``tower_major``, [T, N, S*A*(F/T)] so that the per-tower batched GEMMs that
    follow read contiguous matrices): every aggregator of ``plan`` x every applied scaler over the messages
    ``m_j = x_src[src_j] + x_dst[i] + m_edge[j]`` (``m_edge`` in CSR slot order, see
    ``DGNGraph.to_slot_order``).  ``x_in`` is ``h_in`` of the reference's reduce_func; if it is the
    same tensor as ``x_src`` (simple layer) both gradients land in one buffer.  ``x_pair [N, 2F]`` gives
    ``x_src | x_dst`` as the column halves of one tensor (the P|Q GEMM output of the complex/towers layers) and
    gets ONE gradient tensor back.  ``edge_type`` (int32 [E], CSR slot order): ``m_edge`` is a table [K, F] and slot j adds row
    ``edge_type[j]`` (edge features that are an embedding lookup: no [E, F] tensor is ever formed; the table gets its gradient)."""
    avg = float(avg_log.item()) if torch.is_tensor(avg_log) else float(avg_log)
    w = weights if weights is not None else graph.edge_weights(plan, eig)
    if x_pair is not None:
        if x_src is not None or x_dst is not None:
            raise ValueError("x_pair replaces x_src and x_dst")
        return _DirectionalAggregate.apply(graph, plan, n_towers, avg, w, None, None, m_edge, x_in, False, tower_major, x_pair, edge_type)
    xin_is_src = x_in is not None and x_in is x_src
    return _DirectionalAggregate.apply(graph, plan, n_towers, avg, w, x_src, x_dst, m_edge,
                                       None if xin_is_src else x_in, xin_is_src, tower_major, None, edge_type)


class _ScaleCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, scale, bias, row_scale):
        lib = _lib.load()
        if not z.is_cuda:
            raise _lib.DgnError("scale_combine: CUDA tensors only (dgn_amd has no CPU path)")
        T, N, W = z.shape
        S = 1 if scale is None else scale.shape[1]
        fo = W // S
        z = z.contiguous()
        y = torch.empty((N, T * fo), dtype=torch.float32, device=z.device)
        stream = _lib.stream_ptr(z.device)
        rc = lib.dgn_scale_combine_forward(N, T, S, fo, z.data_ptr(), _ptr(scale), _ptr(bias), _ptr(row_scale), y.data_ptr(),
                                           y.stride(0), stream)
        _lib.check(rc, "dgn_scale_combine_forward")
        ctx.save_for_backward(scale, row_scale)
        ctx.dims = (T, N, S, fo, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        lib = _lib.load()
        scale, row_scale = ctx.saved_tensors
        T, N, S, fo, has_bias = ctx.dims
        g_y = g_y.contiguous()
        g_z = torch.empty((T, N, S * fo), dtype=torch.float32, device=g_y.device)
        g_b = torch.zeros(T * fo, dtype=torch.float32, device=g_y.device) if (has_bias and ctx.needs_input_grad[2]) else None
        ws_bytes = lib.dgn_scale_combine_backward_workspace_bytes(N, T, fo) if g_b is not None else 0
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=g_y.device) if ws_bytes else None
        stream = _lib.stream_ptr(g_y.device)
        rc = lib.dgn_scale_combine_backward(N, T, S, fo, g_y.data_ptr(), g_y.stride(0), _ptr(scale), _ptr(row_scale),
                                            g_z.data_ptr(), _ptr(g_b), _ptr(ws), ws_bytes, None, stream)
        _lib.check(rc, "dgn_scale_combine_backward")
        return g_z, None, g_b, None


def scale_combine(z: torch.Tensor, scale: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                  row_scale: Optional[torch.Tensor]) -> torch.Tensor:
    """y[n, t*fo+o] = row_scale[n] * (bias[t*fo+o] + sum_s scale[n,s] * z[t, n, s*fo+o]).

    ``z [T, N, S*fo]`` is the output of the (batched) post-aggregation GEMM on the scaler-free sweep output,
    ``scale [N, S]`` the degree-scaler table (None: single identity), ``row_scale [N]`` the graph-norm factor
    snorm_n (None: no graph norm).  One streaming kernel instead of mul + sum + add + mul + re-layout."""
    if scale is not None:
        scale = scale.contiguous()
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).contiguous()
    if bias is not None:
        bias = bias.reshape(-1).contiguous()
    return _ScaleCombine.apply(z, scale, bias, row_scale)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        lib = _lib.load()
        x = x.contiguous()
        n = x.numel()
        y = torch.empty_like(x)
        mask = torch.empty(lib.dgn_dropout_mask_bytes(n), dtype=torch.uint8, device=x.device)
        _lib.check(lib.dgn_dropout_forward(n, x.data_ptr(), float(p), seed.data_ptr(), int(offset), y.data_ptr(), mask.data_ptr(),
                                           _lib.stream_ptr(x.device)), "dgn_dropout_forward")
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        ctx.mark_non_differentiable(mask)
        return y, mask

    @staticmethod
    def backward(ctx, g_y, _g_mask):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        g_y = g_y.contiguous()
        g_x = torch.empty_like(g_y)
        _lib.check(lib.dgn_dropout_backward(g_y.numel(), g_y.data_ptr(), mask.data_ptr(), ctx.p, g_x.data_ptr(), _lib.stream_ptr(g_y.device)),
                   "dgn_dropout_backward")
        return g_x, None, None, None


LAST_DROPOUT_MASK = None     # (tests) the keep-bit tensor of the most recent dropout() call: bit i of byte g = element 8 g + i


def dropout(x: torch.Tensor, p: float, training: bool, seed: Optional[torch.Tensor] = None, offset: int = 0) -> torch.Tensor:
    """``F.dropout(x, p, training)`` of the reference's layer tails (nets/dgn_layer.py:130, :201, :275) as one kernel per direction with a
    BIT mask saved for the backward (N F / 8 bytes instead of autograd's fp32 product operands).  ``seed``: a device int64 scalar (default:
    drawn from torch's generator of the device, so ``torch.manual_seed`` reproduces the masks and the draw is capturable)."""
    global LAST_DROPOUT_MASK
    if not training or p == 0.0:
        return x
    if p >= 1.0:
        return x * 0.0
    if not x.is_cuda or x.dtype != torch.float32:
        raise _lib.DgnError("dropout: CUDA fp32 tensors only (dgn_amd has no CPU path)")
    if seed is None:
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=x.device)
    y, mask = _Dropout.apply(x, p, seed, offset)
    LAST_DROPOUT_MASK = mask
    return y


class _BNTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, residual):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.DgnError("bn_tail: CUDA tensors only (dgn_amd has no CPU path)")
        x = x.contiguous()
        N, F = x.shape
        if residual is not None:
            residual = residual.contiguous()
        y = torch.empty_like(x)
        save_mean = torch.empty(F, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(F, dtype=torch.float32, device=x.device)
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F) if training else 0
        ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=x.device) if ws_bytes else None
        stream = _lib.stream_ptr(x.device)
        ctx.n_valid = _N_VALID
        rc = lib.dgn_bn_tail_forward(N, F, x.data_ptr(), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                     float(momentum), float(eps), 1 if training else 0, 1 if relu else 0, _ptr(residual), y.data_ptr(),
                                     save_mean.data_ptr(), save_invstd.data_ptr(), _ptr(ws), ws_bytes, _ptr(ctx.n_valid), stream)
        _lib.check(rc, "dgn_bn_tail_forward")
        ctx.save_for_backward(x, gamma, beta, save_mean, save_invstd)
        ctx.relu, ctx.has_res = relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        lib = _lib.load()
        x, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        g_y = g_y.contiguous()
        N, F = x.shape
        g_x = torch.empty_like(x)
        g_gamma = torch.empty(F, dtype=torch.float32, device=x.device) if gamma is not None else None
        g_beta = torch.empty(F, dtype=torch.float32, device=x.device) if beta is not None else None
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=x.device)
        stream = _lib.stream_ptr(x.device)
        rc = lib.dgn_bn_tail_backward(N, F, g_y.data_ptr(), x.data_ptr(), x.stride(0), _ptr(gamma), _ptr(beta), save_mean.data_ptr(),
                                      save_invstd.data_ptr(), 1 if ctx.relu else 0, g_x.data_ptr(), _ptr(g_gamma), _ptr(g_beta),
                                      None, ws.data_ptr(), ws_bytes, _ptr(ctx.n_valid), stream)
        _lib.check(rc, "dgn_bn_tail_backward")
        return g_x, g_gamma, g_beta, None, None, None, None, None, None, (g_y if ctx.has_res else None)


class _CombineBNTail(torch.autograd.Function):
    """scale_combine followed by the training-mode BatchNorm tail as ONE autograd node: the backward forms the
    combine's upstream gradient from the tail's inputs on the fly (DgnBnGrad), so the [N, F] gradient between the two
    steps is never written or re-read."""

    @staticmethod
    def forward(ctx, z, scale, bias, row_scale, gamma, beta, running_mean, running_var, momentum, eps, relu, residual):
        lib = _lib.load()
        if not z.is_cuda:
            raise _lib.DgnError("combine_bn_tail: CUDA tensors only (dgn_amd has no CPU path)")
        T, N, W = z.shape
        S = 1 if scale is None else scale.shape[1]
        fo = W // S
        F = T * fo
        z = z.contiguous()
        dev = z.device
        stream = _lib.stream_ptr(dev)
        y = torch.empty((N, F), dtype=torch.float32, device=dev)
        rc = lib.dgn_scale_combine_forward(N, T, S, fo, z.data_ptr(), _ptr(scale), _ptr(bias), _ptr(row_scale), y.data_ptr(), y.stride(0), stream)
        _lib.check(rc, "dgn_scale_combine_forward")
        if residual is not None:
            residual = residual.contiguous()
        out = torch.empty_like(y)
        save_mean = torch.empty(F, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(F, dtype=torch.float32, device=dev)
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=dev)
        ctx.n_valid = _N_VALID
        rc = lib.dgn_bn_tail_forward(N, F, y.data_ptr(), y.stride(0), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                     float(momentum), float(eps), 1, 1 if relu else 0, _ptr(residual), out.data_ptr(),
                                     save_mean.data_ptr(), save_invstd.data_ptr(), ws.data_ptr(), ws_bytes, _ptr(ctx.n_valid), stream)
        _lib.check(rc, "dgn_bn_tail_forward")
        ctx.save_for_backward(scale, row_scale, y, gamma, beta, save_mean, save_invstd)
        ctx.dims = (T, N, S, fo, bias is not None, relu, residual is not None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        scale, row_scale, y, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        T, N, S, fo, has_bias, relu, has_res = ctx.dims
        F = T * fo
        dev = y.device
        g_out = g_out.contiguous()
        stream = _lib.stream_ptr(dev)
        g_gamma = torch.empty(F, dtype=torch.float32, device=dev) if gamma is not None else None
        g_beta = torch.empty(F, dtype=torch.float32, device=dev) if beta is not None else None
        sums = torch.empty(2 * F, dtype=torch.float32, device=dev)
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=dev)
        rc = lib.dgn_bn_tail_backward(N, F, g_out.data_ptr(), y.data_ptr(), y.stride(0), _ptr(gamma), _ptr(beta), save_mean.data_ptr(),
                                      save_invstd.data_ptr(), 1 if relu else 0, None, _ptr(g_gamma), _ptr(g_beta), sums.data_ptr(),
                                      ws.data_ptr(), ws_bytes, _ptr(ctx.n_valid), stream)
        _lib.check(rc, "dgn_bn_tail_backward")
        bn = _lib.DgnBnGrad(g_out=g_out.data_ptr(), y=y.data_ptr(), ld=y.stride(0), gamma=_ptr(gamma), beta=_ptr(beta),
                            mean=save_mean.data_ptr(), invstd=save_invstd.data_ptr(), sums=sums.data_ptr(), relu=1 if relu else 0,
                            n_valid=_ptr(ctx.n_valid))
        g_z = torch.empty((T, N, S * fo), dtype=torch.float32, device=dev)
        g_b = torch.zeros(F, dtype=torch.float32, device=dev) if (has_bias and ctx.needs_input_grad[2]) else None
        ws2_bytes = lib.dgn_scale_combine_backward_workspace_bytes(N, T, fo) if g_b is not None else 0
        ws2 = torch.empty(ws2_bytes // 4, dtype=torch.float32, device=dev) if ws2_bytes else None
        rc = lib.dgn_scale_combine_backward(N, T, S, fo, None, 0, _ptr(scale), _ptr(row_scale), g_z.data_ptr(), _ptr(g_b), _ptr(ws2),
                                            ws2_bytes, C.byref(bn), stream)
        _lib.check(rc, "dgn_scale_combine_backward")
        return g_z, None, g_b, None, g_gamma, g_beta, None, None, None, None, None, (g_out if has_res else None)


def combine_bn_tail(z, scale, bias, row_scale, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                    relu: bool = False, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[relu](BatchNorm(scale_combine(z, scale, bias, row_scale))) [+ residual]`` in TRAINING mode as one autograd node
    (see _CombineBNTail); running statistics are updated in place.  Needs n_towers * f_out <= 1024."""
    if scale is not None:
        scale = scale.contiguous()
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).contiguous()
    if bias is not None:
        bias = bias.reshape(-1).contiguous()
    out = _CombineBNTail.apply(z, scale, bias, row_scale, gamma, beta, running_mean, running_var, momentum, eps, relu, residual)
    if num_batches_tracked is not None:
        with torch.no_grad():
            num_batches_tracked.add_(1)
    return out


_ACT_CODES = {"none": 0, "relu": 1, "leaky_relu": 2}


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, act, slope, residual):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.DgnError("bias_act: CUDA tensors only (dgn_amd has no CPU path)")
        x = x.contiguous()
        N, F = x.shape
        if residual is not None:
            residual = residual.contiguous()
        if bias is not None:
            bias = bias.contiguous()
        y = torch.empty_like(x)
        stream = _lib.stream_ptr(x.device)
        rc = lib.dgn_bias_act_forward(N, F, x.data_ptr(), x.stride(0), _ptr(bias), act, float(slope), _ptr(residual), y.data_ptr(), stream)
        _lib.check(rc, "dgn_bias_act_forward")
        ctx.save_for_backward(x, bias)
        ctx.act, ctx.slope, ctx.has_res = act, float(slope), residual is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        lib = _lib.load()
        x, bias = ctx.saved_tensors
        N, F = x.shape
        g_y = g_y.contiguous()
        g_x = torch.empty_like(x)
        g_b = torch.empty(F, dtype=torch.float32, device=x.device) if (bias is not None and ctx.needs_input_grad[1]) else None
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=x.device)
        stream = _lib.stream_ptr(x.device)
        rc = lib.dgn_bias_act_backward(N, F, g_y.data_ptr(), x.data_ptr(), x.stride(0), _ptr(bias), ctx.act, ctx.slope, g_x.data_ptr(),
                                       _ptr(g_b), ws.data_ptr(), ws_bytes, stream)
        _lib.check(rc, "dgn_bias_act_backward")
        return g_x, g_b, None, None, (g_y if ctx.has_res else None)


def bias_act(x: torch.Tensor, bias: Optional[torch.Tensor], act: str = "none", slope: float = 0.01,
             residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(x + bias) [+ residual]`` on the bias-free GEMM output ``x [N, F]`` in one pass (the tail of an FCLayer,
    layers.py:101-112; the towers' mixing network with the layer's residual, dgn_layer.py:319-324); the backward
    produces ``g_x`` and the bias gradient together.  ``act``: none | relu | leaky_relu.  F <= 1024."""
    return _BiasAct.apply(x, bias, _ACT_CODES[act], slope, residual)


def bn_tail_fused(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor, running_var: torch.Tensor,
                  num_batches_tracked: Optional[torch.Tensor], momentum: float, eps: float, training: bool, relu: bool = False,
                  residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The tail on ALREADY concatenated per-channel tensors (the towers layer keeps its BatchNorm parameters in the
    fused operand buffer and its running statistics in one tensor the per-tower modules view): no cat before and no
    scatter after the kernels.  Running statistics are updated in place."""
    y = _BNTail.apply(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, residual)
    if training and num_batches_tracked is not None:
        with torch.no_grad():
            num_batches_tracked.add_(1)
    return y


def _spans_ranks(bn, training: bool) -> bool:
    """A dist.SyncBatchNorm1d in training mode: its statistics come from an all-reduce, not from the local rows the fused kernels see."""
    return training and getattr(bn, "dgn_sync", False) and torch.distributed.is_available() and torch.distributed.is_initialized()


def bn_tail_supported(bns, x: torch.Tensor, training: bool, width: Optional[int] = None) -> bool:
    """What the fused tail kernels cover: affine BatchNorm with running statistics, F <= 1024 (``width``, default
    ``x.shape[1]``), and -- with gradients -- training mode."""
    simple = all(b.affine and b.track_running_stats and b.momentum is not None and not _spans_ranks(b, training) for b in bns)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for b in bns for p in b.parameters()))
    return simple and (x.shape[1] if width is None else width) <= 1024 and (training or not needs_grad)


def bn_tail(x: torch.Tensor, bns, training: bool, relu: bool = False, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[relu](BatchNorm1d(x)) [+ residual]`` in one pass family (dgn_layer.py:123-128, :194-199, :272-273).

    ``bns``: one ``nn.BatchNorm1d`` or a list of them covering consecutive column blocks of ``x`` (the towers'
    per-tower BatchNorm1d modules: BatchNorm is per channel, so T modules of width fo are one BatchNorm of width
    T*fo).  Running statistics and ``num_batches_tracked`` of every module are updated as torch does."""
    if not isinstance(bns, (list, tuple)):
        bns = [bns]
    b0 = bns[0]
    simple = all(b.affine and b.track_running_stats and b.momentum is not None and not _spans_ranks(b, training) for b in bns)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for b in bns for p in b.parameters()))
    if not simple or x.shape[1] > 1024 or (not training and needs_grad):   # (fused: training-mode backward, F <= 1024)
        # configurations the fused kernels do not cover: plain torch modules
        if _N_VALID is not None and training:
            raise _lib.DgnError("padded batch (n_valid): this BatchNorm configuration runs on plain torch modules, which would count the "
                                "padding rows in the batch statistics; use affine BatchNorm with running statistics, width <= 1024")
        w = x.shape[1] // len(bns)
        if len(bns) > 1 and all(_spans_ranks(b, training) and b.affine and b.track_running_stats and b.momentum is not None for b in bns):
            # the towers' SyncBatchNorm1d modules as ONE BatchNorm of width T * fo: one all-reduce per direction instead of T
            from .dist import sync_batch_norm
            rm, rv = torch.cat([b.running_mean for b in bns]), torch.cat([b.running_var for b in bns])
            y = sync_batch_norm(x, torch.cat([b.weight for b in bns]), torch.cat([b.bias for b in bns]), rm, rv, b0.momentum, b0.eps, b0.process_group)
            with torch.no_grad():
                torch._foreach_copy_([b.running_mean for b in bns], list(rm.split(w)))
                torch._foreach_copy_([b.running_var for b in bns], list(rv.split(w)))
                torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
        else:
            y = torch.cat([b(x[:, i * w:(i + 1) * w]) for i, b in enumerate(bns)], dim=1) if len(bns) > 1 else b0(x)
        y = torch.relu(y) if relu else y
        return y + residual if residual is not None else y
    cat = (lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts))
    gamma, beta = cat([b.weight for b in bns]), cat([b.bias for b in bns])
    rm, rv = cat([b.running_mean for b in bns]), cat([b.running_var for b in bns])
    y = _BNTail.apply(x, gamma, beta, rm, rv, b0.momentum, b0.eps, training, relu, residual)
    if training:
        with torch.no_grad():
            if len(bns) > 1:      # scatter the updated statistics back to the per-tower modules (2 + 1 launches)
                w = x.shape[1] // len(bns)
                torch._foreach_copy_([b.running_mean for b in bns], list(rm.split(w)))
                torch._foreach_copy_([b.running_var for b in bns], list(rv.split(w)))
            torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
    return y


# ---- tall-skinny Linear (dgn_linear.hip) -------------------------------------------------------------------------------

def _lin_fwd(lib, a, w, w_is_kn, bias, n):
    """a [T, M, k] dense, w [T, n, k] (or [T, k, n] with w_is_kn) -> [T, M, n]"""
    T, M, k = a.shape
    c = torch.empty(T, M, n, dtype=torch.float32, device=a.device)
    stream = _lib.stream_ptr(a.device)
    rc = lib.dgn_linear_forward(M, k, n, T, a.data_ptr(), k, a.stride(0), w.data_ptr(), w.stride(1), w.stride(0), int(w_is_kn),
                                _ptr(bias), bias.stride(0) if bias is not None else 0, c.data_ptr(), n, M * n, stream)
    _lib.check(rc, "dgn_linear_forward")
    return c


class _TsLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.DgnError("linear: CUDA tensors only (dgn_amd has no CPU path)")
        x, w = x.contiguous(), w.contiguous()
        bias = bias.contiguous() if bias is not None else None
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return _lin_fwd(lib, x, w, False, bias, w.shape[1])

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        T, M, k = x.shape
        n = w.shape[1]
        g = g.contiguous()
        g_x = _lin_fwd(lib, g, w, True, None, k) if ctx.needs_input_grad[0] else None
        g_w = g_b = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            g_w, g_b = _lin_wgrad(lib, g, x, want_b)
        elif want_b:
            g_b = g.sum(dim=1)
        return g_x, g_w, g_b


_LIN_OK = {}


def linear_supported(k: int, n: int) -> bool:
    """Whether ``linear`` takes an [*, k] x [n, k] product: even widths up to 160 (forward, input gradient, and a
    weight gradient of at most 45 16x16 tiles).  Otherwise use ``F.linear``."""
    key = (int(k), int(n))
    if key not in _LIN_OK:
        lib = _lib.load()
        _LIN_OK[key] = bool(lib.dgn_linear_supported(key[0], key[1], 1)) and bool(lib.dgn_linear_supported(key[1], key[0], 0))
    return _LIN_OK[key]


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear(x, weight, bias)`` for node-count-tall operands on the streaming MFMA kernels, exact fp32.

    ``x [M, k]`` with ``weight [n, k]`` (``bias [n]``), or batched over towers: ``x [T, M, k]``, ``weight [T, n, k]``
    (``bias [T, n]``) = ``torch.bmm(x, weight.transpose(1, 2))``.  Forward, input gradient and weight gradient each are
    one pass over the rows with the weights resident in LDS (``include/dgn_hip.h: dgn_linear_*``)."""
    if x.dim() == 2:
        y = _TsLinear.apply(x.unsqueeze(0), weight.unsqueeze(0), bias.unsqueeze(0) if bias is not None else None)
        return y.squeeze(0)
    return _TsLinear.apply(x, weight, bias)


# Below this many rows a product goes to the library GEMM: the streaming kernels spend ~10 us staging the weights in LDS,
# which only pays off once there are enough 16-row strips to stream (a 128-molecule batch has ~3 000 rows: 9 products
# of a captured towers step cost 0.26 ms on these kernels, 0.20 ms on the library's).
LINEAR_MIN_ROWS = int(os.environ.get("DGN_LINEAR_MIN_ROWS", "8192"))


def node_linear_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Whether ``node_linear`` runs this product on the streaming kernels (fp32 on the GPU, even widths up to 160, at least
    ``LINEAR_MIN_ROWS`` rows)."""
    return bool(x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == weight.dim() and x.dim() in (2, 3)
                and x.shape[-2] >= LINEAR_MIN_ROWS and os.environ.get("DGN_LIBRARY_GEMM") != "1"
                and linear_supported(x.shape[-1], weight.shape[-2]))


class _WideLinear(torch.autograd.Function):
    """``F.linear(x, w, b)`` on the wide GEMM kernels (dgn_gemm_*): forward, input gradient and weight gradient each one kernel
    family of this library (no library GEMM, no shape-keyed tuning)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.DgnError("wide_linear: CUDA tensors only (dgn_amd has no CPU path)")
        if x.stride(1) != 1:
            x = x.contiguous()
        w = w.contiguous()
        M, k = x.shape
        n = w.shape[0]
        c = torch.empty((M, n), dtype=torch.float32, device=x.device)
        stream = _lib.stream_ptr(x.device)
        rc = lib.dgn_gemm_forward(M, k, n, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), 0, _ptr(bias.contiguous() if bias is not None else None),
                                  c.data_ptr(), n, stream)
        _lib.check(rc, "dgn_gemm_forward")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return c

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        M, k = x.shape
        n = w.shape[0]
        g = g.contiguous()
        stream = _lib.stream_ptr(x.device)
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty((M, k), dtype=torch.float32, device=x.device)
            wt = w.t().contiguous()          # [k, n] as the "weight" of g_x = g . wt^T: the row-major staging path (5-10 % faster than w_is_kn)
            _lib.check(lib.dgn_gemm_forward(M, n, k, g.data_ptr(), n, wt.data_ptr(), n, 0, None, g_x.data_ptr(), k, stream), "dgn_gemm_forward (input gradient)")
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            # (the bias gradient is a column of ones appended to x inside the weight-gradient kernel: no reduction pass of its own)
            g_w = torch.empty((n, k), dtype=torch.float32, device=x.device)
            g_b = torch.empty(n, dtype=torch.float32, device=x.device) if want_b else None
            nbytes = lib.dgn_gemm_wgrad_workspace_bytes(M, k, n)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
            _lib.check(lib.dgn_gemm_wgrad(M, k, n, g.data_ptr(), n, x.data_ptr(), x.stride(0), g_w.data_ptr(), k, _ptr(g_b), ws.data_ptr(), nbytes, stream),
                       "dgn_gemm_wgrad")
        return g_x, g_w, g_b


# Wide products (k or n beyond the streaming kernels' 160 columns, or odd widths) go to the dgn_gemm_* kernels from this many rows on.
WIDE_MIN_ROWS = int(os.environ.get("DGN_WIDE_MIN_ROWS", "4096"))
# The whole simple / complex layer as one C call per direction has NO row threshold of that kind: below 4096 rows the alternative is not
# "the library's GEMM" but the per-op route with its torch glue (13 copies, 11 fills and two library GEMMs per step on the shipped ZINC
# json layer at batch 128): measured 0.77 -> 0.39 ms eager and 0.229 -> 0.186 ms in a captured step (round 4).
WHOLE_LAYER_MIN_ROWS = int(os.environ.get("DGN_WHOLE_LAYER_MIN_ROWS", "0"))


def wide_linear_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return bool(x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and weight.dim() == 2
                and x.shape[0] >= WIDE_MIN_ROWS and 4 <= weight.shape[0] <= 4096 and os.environ.get("DGN_LIBRARY_GEMM") != "1"
                and 4 <= x.shape[1] <= 4096)


def wide_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _WideLinear.apply(x, weight, bias)


def node_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear`` / batched ``bmm(x, weight^T)`` of the layers' node-count-tall operands: the streaming kernels where
    they apply (even widths up to 160), the wide GEMM kernels for the simple / complex layers' posttrans shapes, the library
    GEMM only for small batches and CPU glue in the tests."""
    if node_linear_supported(x, weight):
        return linear(x, weight, bias)
    if wide_linear_supported(x, weight):
        return wide_linear(x, weight, bias)
    if x.dim() == 2:
        return torch.nn.functional.linear(x, weight, bias)
    y = torch.bmm(x, weight.transpose(1, 2))
    return y if bias is None else y + bias.unsqueeze(1)


# ---- block-diagonal P|Q Linear of the towers layer (dgn_linear_bd.hip) --------------------------------------------------------------

BLOCK_DIAGONAL = os.environ.get("DGN_NO_BD") is None      # False / DGN_NO_BD=1: the dense streaming kernels multiply the zero blocks too

_BD_OK = {}


def pair_linear_supported(x: torch.Tensor, n_towers: int, f_in: int) -> bool:
    """Whether ``pair_linear`` takes this shape: an instantiated (towers, f_in) pair, fp32 CUDA rows, at least LINEAR_MIN_ROWS of them."""
    key = (int(n_towers), int(f_in))
    if key not in _BD_OK:
        _BD_OK[key] = bool(_lib.load().dgn_linear_bd_supported(*key))
    return bool(BLOCK_DIAGONAL and _BD_OK[key] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= LINEAR_MIN_ROWS
                and os.environ.get("DGN_LIBRARY_GEMM") != "1")


class _PairLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, T, fi):
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        M, Fm = x.shape
        c = torch.empty((M, 2 * Fm), dtype=torch.float32, device=x.device)
        stream = _lib.stream_ptr(x.device)
        _lib.check(lib.dgn_linear_bd_forward(M, T, fi, x.data_ptr(), w.data_ptr(), w.stride(0), _ptr(bias.contiguous() if bias is not None else None),
                                             c.data_ptr(), stream), "dgn_linear_bd_forward")
        ctx.save_for_backward(x, w)
        ctx.dims, ctx.has_bias = (T, fi), bias is not None
        return c

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        T, fi = ctx.dims
        M, Fm = x.shape
        g = g.contiguous()
        stream = _lib.stream_ptr(x.device)
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(x)
            _lib.check(lib.dgn_linear_bd_backward_input(M, T, fi, g.data_ptr(), w.data_ptr(), w.stride(0), None, None, g_x.data_ptr(), stream),
                       "dgn_linear_bd_backward_input")
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            g_w = torch.empty_like(w)
            g_b = torch.empty(2 * Fm, dtype=torch.float32, device=x.device) if want_b else None
            nbytes = lib.dgn_linear_bd_wgrad_workspace_bytes(M, T, fi)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
            _lib.check(lib.dgn_linear_bd_wgrad(M, T, fi, g.data_ptr(), x.data_ptr(), g_w.data_ptr(), g_w.stride(0), _ptr(g_b), ws.data_ptr(), nbytes,
                                               stream), "dgn_linear_bd_wgrad")
        return g_x, g_w, g_b, None, None


def pair_linear(x: torch.Tensor, w_sd: torch.Tensor, bias_sd: Optional[torch.Tensor], n_towers: int, f_in: int) -> torch.Tensor:
    """``F.linear(x, w_sd, bias_sd)`` for the towers layer's BLOCK-DIAGONAL P | Q weights (``DGNLayerTower._assemble``: ``w_sd``
    [2 Fm, Fm] whose only non-zero entries are the towers' [f_in, f_in] blocks, nets/dgn_layer.py:226-231 under divide_input): the
    structural zeros are neither multiplied nor -- in the weight gradient, which comes back dense with zero off-diagonal blocks -- formed."""
    return _PairLinear.apply(x, w_sd, bias_sd, int(n_towers), int(f_in))


def _lin_wgrad(lib, g, x, want_bias):
    """g [T, M, n], x [T, M, k] dense -> (dW [T, n, k], dbias [T, n] | None)"""
    T, M, n = g.shape
    k = x.shape[2]
    g_w = torch.empty(T, n, k, dtype=torch.float32, device=x.device)
    g_b = torch.empty(T, n, dtype=torch.float32, device=x.device) if (want_bias and k % 16 != 0) else None
    ws_bytes = lib.dgn_linear_wgrad_workspace_bytes(M, k, n, T)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=x.device)
    stream = _lib.stream_ptr(x.device)
    rc = lib.dgn_linear_wgrad(M, k, n, T, g.data_ptr(), n, g.stride(0), x.data_ptr(), k, x.stride(0), g_w.data_ptr(), k, n * k,
                              _ptr(g_b), n, ws.data_ptr(), ws_bytes, stream)
    _lib.check(rc, "dgn_linear_wgrad")
    if want_bias and g_b is None:
        g_b = g.sum(dim=1)
    return g_w, g_b


class _LinCombineBNTail(torch.autograd.Function):
    """The towers' posttrans Linear, the scale-combine and the training-mode BatchNorm tail as ONE autograd node: the
    forward never writes the [T, N, S*fo] product (dgn_linear_combine_forward), the backward forms its gradient from
    the tail's inputs (see _CombineBNTail) and runs the input / weight gradients on the streaming kernels."""

    @staticmethod
    def forward(ctx, aggx, w, scale, bias, row_scale, gamma, beta, running_mean, running_var, momentum, eps, relu, residual):
        lib = _lib.load()
        if not aggx.is_cuda:
            raise _lib.DgnError("linear_combine_bn_tail: CUDA tensors only (dgn_amd has no CPU path)")
        aggx, w = aggx.contiguous(), w.contiguous()
        T, N, k = aggx.shape
        S = 1 if scale is None else scale.shape[1]
        fo = w.shape[1] // S
        F = T * fo
        dev = aggx.device
        stream = _lib.stream_ptr(dev)
        y = torch.empty((N, F), dtype=torch.float32, device=dev)
        rc = lib.dgn_linear_combine_forward(N, k, T, S, fo, aggx.data_ptr(), aggx.stride(0), w.data_ptr(), w.stride(1), w.stride(0),
                                            _ptr(scale), _ptr(bias), _ptr(row_scale), y.data_ptr(), y.stride(0), stream)
        _lib.check(rc, "dgn_linear_combine_forward")
        if residual is not None:
            residual = residual.contiguous()
        out = torch.empty_like(y)
        save_mean = torch.empty(F, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(F, dtype=torch.float32, device=dev)
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=dev)
        ctx.n_valid = _N_VALID
        rc = lib.dgn_bn_tail_forward(N, F, y.data_ptr(), y.stride(0), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                     float(momentum), float(eps), 1, 1 if relu else 0, _ptr(residual), out.data_ptr(),
                                     save_mean.data_ptr(), save_invstd.data_ptr(), ws.data_ptr(), ws_bytes, _ptr(ctx.n_valid), stream)
        _lib.check(rc, "dgn_bn_tail_forward")
        ctx.save_for_backward(scale, row_scale, y, gamma, beta, save_mean, save_invstd, aggx, w)
        ctx.dims = (T, N, S, fo, bias is not None, relu, residual is not None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        scale, row_scale, y, gamma, beta, save_mean, save_invstd, aggx, w = ctx.saved_tensors
        T, N, S, fo, has_bias, relu, has_res = ctx.dims
        F, k = T * fo, aggx.shape[2]
        dev = y.device
        g_out = g_out.contiguous()
        stream = _lib.stream_ptr(dev)
        g_gamma = torch.empty(F, dtype=torch.float32, device=dev) if gamma is not None else None
        g_beta = torch.empty(F, dtype=torch.float32, device=dev) if beta is not None else None
        sums = torch.empty(2 * F, dtype=torch.float32, device=dev)
        ws_bytes = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=dev)
        rc = lib.dgn_bn_tail_backward(N, F, g_out.data_ptr(), y.data_ptr(), y.stride(0), _ptr(gamma), _ptr(beta), save_mean.data_ptr(),
                                      save_invstd.data_ptr(), 1 if relu else 0, None, _ptr(g_gamma), _ptr(g_beta), sums.data_ptr(),
                                      ws.data_ptr(), ws_bytes, _ptr(ctx.n_valid), stream)
        _lib.check(rc, "dgn_bn_tail_backward")
        bn = _lib.DgnBnGrad(g_out=g_out.data_ptr(), y=y.data_ptr(), ld=y.stride(0), gamma=_ptr(gamma), beta=_ptr(beta),
                            mean=save_mean.data_ptr(), invstd=save_invstd.data_ptr(), sums=sums.data_ptr(), relu=1 if relu else 0,
                            n_valid=_ptr(ctx.n_valid))
        # g_yr = row_scale * (BatchNorm backward of g_out), tower-major [T, N, fo]: the combine backward run with ONE scaler and
        # no scale table; the per-scaler expansion happens inside the two products below
        g_yr = torch.empty((T, N, fo), dtype=torch.float32, device=dev)
        g_b = torch.zeros(F, dtype=torch.float32, device=dev) if (has_bias and ctx.needs_input_grad[3]) else None
        ws2_bytes = lib.dgn_scale_combine_backward_workspace_bytes(N, T, fo) if g_b is not None else 0
        ws2 = torch.empty(ws2_bytes // 4, dtype=torch.float32, device=dev) if ws2_bytes else None
        rc = lib.dgn_scale_combine_backward(N, T, 1, fo, None, 0, None, _ptr(row_scale), g_yr.data_ptr(), _ptr(g_b), _ptr(ws2),
                                            ws2_bytes, C.byref(bn), stream)
        _lib.check(rc, "dgn_scale_combine_backward")
        g_aggx = g_w = None
        if ctx.needs_input_grad[0]:
            g_aggx = torch.empty_like(aggx)
            rc = lib.dgn_linear_combine_backward_input(N, T, S, fo, k, g_yr.data_ptr(), g_yr.stride(0), _ptr(scale), w.data_ptr(), w.stride(1),
                                                       w.stride(0), g_aggx.data_ptr(), g_aggx.stride(0), stream)
            _lib.check(rc, "dgn_linear_combine_backward_input")
        if ctx.needs_input_grad[1]:
            g_w = torch.empty_like(w)
            ws3_bytes = lib.dgn_linear_wgrad_workspace_bytes(N, k, S * fo, T)
            ws3 = torch.empty(max(ws3_bytes // 4, 1), dtype=torch.float32, device=dev)
            rc = lib.dgn_linear_combine_backward_weight(N, T, S, fo, k, g_yr.data_ptr(), g_yr.stride(0), _ptr(scale), aggx.data_ptr(), aggx.stride(0),
                                                        g_w.data_ptr(), k, S * fo * k, ws3.data_ptr(), ws3_bytes, stream)
            _lib.check(rc, "dgn_linear_combine_backward_weight")
        return g_aggx, g_w, None, g_b, None, g_gamma, g_beta, None, None, None, None, None, (g_out if has_res else None)


def linear_combine_supported(aggx: torch.Tensor, w: torch.Tensor, n_scalers: int) -> bool:
    """Whether ``linear_combine_bn_tail`` takes this posttrans product: the streaming kernels apply, at most 3 scalers,
    an even per-tower output width."""
    return node_linear_supported(aggx, w) and 1 <= n_scalers <= 3 and w.shape[-2] % n_scalers == 0 and (w.shape[-2] // n_scalers) % 2 == 0


def linear_combine_bn_tail(aggx, w, scale, bias, row_scale, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                           eps, relu: bool = False, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[relu](BatchNorm(scale_combine(bmm(aggx, w^T), scale, bias, row_scale))) [+ residual]`` in TRAINING mode as one
    autograd node; ``aggx [T, N, k]`` (the sweep's tower-major output), ``w [T, S*fo, k]``.  Running statistics are
    updated in place.  Needs ``linear_supported(k, S*fo)`` and ``T*fo <= 1024``."""
    if scale is not None:
        scale = scale.contiguous()
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).contiguous()
    if bias is not None:
        bias = bias.reshape(-1).contiguous()
    out = _LinCombineBNTail.apply(aggx, w, scale, bias, row_scale, gamma, beta, running_mean, running_var, momentum, eps, relu, residual)
    if num_batches_tracked is not None:
        with torch.no_grad():
            num_batches_tracked.add_(1)
    return out


# ---- the whole towers layer as ONE autograd node over two C calls (dgn_towers.hip) --------------------------------------------

# True: DGNLayerTower runs its fused configuration (single-affine pretrans / posttrans, divide_input, scalers folded, training-mode
# BatchNorm, mixing network, no edge features, no dropout) through dgn_towers_layer_forward / _backward: every kernel of the
# layer is enqueued by one call per direction.  At the reference's batch size the layer is host-bound otherwise (~48 launches,
# each its own Python / autograd node).  False: the per-kernel route (same kernels, same results).
WHOLE_LAYER = os.environ.get("DGN_WHOLE_LAYER", "1") != "0"
FUSE_BN_MIXING = os.environ.get("DGN_FUSE_BN", "1") != "0"      # whole-layer path: BatchNorm apply inside the mixing Linear's operand staging

_TOWERS_OK = {}


def towers_layer_supported(n_towers: int, f_in: int, f_out: int, n_scalers: int, n_agg_total: int) -> bool:
    key = (n_towers, f_in, f_out, n_scalers, n_agg_total)
    if key not in _TOWERS_OK:
        _TOWERS_OK[key] = bool(_lib.load().dgn_towers_layer_supported(*key))
    return _TOWERS_OK[key]


def _carve(sizes, device):
    """one allocation, views of the given element counts (each view starts 256-byte aligned)"""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) & ~63
    buf = torch.empty(max(total, 1), dtype=torch.float32, device=device)
    return buf, [buf[o:o + n] for o, n in zip(offs, sizes)]


def _carve_ptrs(sizes, device, buf=None):
    """_carve for callers that only pass addresses on: (buffer, [address or None per view]) -- no view tensors are created"""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) & ~63
    if buf is None:
        buf = torch.empty(max(total, 1), dtype=torch.float32, device=device)
    base = buf.data_ptr()
    return buf, [base + 4 * o if n else None for o, n in zip(offs, sizes)]


class _TowersLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, plan, avg_log, w_edge, cfg, h, snorm, scale, running_mean, running_var, nbt,
                w_sd, bias_sd, w_post, b_post, gamma, beta, w_mix, b_mix):
        lib = _lib.load()
        global LAST_DROPOUT_MASK
        T, fi, fo, S, residual, momentum, eps, slope, drop = cfg[:9]
        if not h.is_cuda:
            raise _lib.DgnError("towers_layer: CUDA tensors only (dgn_amd has no CPU path)")
        N, Fm, Fo = h.shape[0], T * fi, T * fo
        K = plan.n_agg * fi
        dev = h.device
        h, w_sd, bias_sd, w_post, b_post = h.contiguous(), w_sd.contiguous(), bias_sd.contiguous(), w_post.contiguous(), b_post.contiguous()
        gamma, beta, w_mix, b_mix = gamma.contiguous(), beta.contiguous(), w_mix.contiguous(), b_mix.contiguous()
        # y1 = BatchNorm(y0) is not materialised (FUSE_BN_MIXING): the mixing Linear and its weight gradient normalise y0 while they
        # stage their strips (dgn_linear_forward_bn / dgn_linear_wgrad_bn) -- one pass and N * Fo saved floats per layer less
        # (with the towers' dropout the normalised rows ARE written: the mask is applied to them in place)
        n_y1 = 0 if FUSE_BN_MIXING and drop is None else N * Fo
        # the mixing network's pre-activation is only needed for the sign of (z + b): where the fused kernels run it is kept as a byte
        # mask (DgnTowersLayer.zmask, 1/8 of the bytes) in the slot z would take
        use_mask = n_y1 == 0 and (h.data_ptr() & 15) == 0 and bool(lib.dgn_towers_layer_zmask_supported(T, fo))
        n_z = (lib.dgn_linear_act_mask_bytes(N, Fo) + 3) // 4 if use_mask else N * Fo
        saved_buf, (pq, aggx, y0, y1, z, mean, invstd) = _carve([N * 2 * Fm, T * N * K, N * Fo, n_y1, n_z, Fo, Fo], dev)
        ctx.n_y1, ctx.n_z, ctx.use_mask = n_y1, n_z, use_mask
        out = torch.empty((N, Fo), dtype=torch.float32, device=dev)
        spec = _spec_structs(plan, T, avg_log, N * K)[0]
        L = _lib.DgnTowersLayer()
        cg = graph.c_graph
        L.graph, L.spec = C.pointer(cg), C.pointer(spec)
        L.w, L.ld_w, L.log_deg = _ptr(w_edge), (w_edge.stride(0) if w_edge is not None else 0), graph.log_deg.data_ptr()
        L.n_towers, L.f_in, L.f_out, L.n_scalers, L.residual = T, fi, fo, S, int(residual)
        L.momentum, L.eps, L.slope = float(momentum), float(eps), float(slope)
        L.h, L.snorm, L.scale = h.data_ptr(), _ptr(snorm), _ptr(scale)
        L.w_sd, L.bias_sd, L.w_post, L.b_post = w_sd.data_ptr(), bias_sd.data_ptr(), w_post.data_ptr(), b_post.data_ptr()
        L.bn_gamma, L.bn_beta, L.running_mean, L.running_var = gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr()
        L.w_mix, L.b_mix = w_mix.data_ptr(), b_mix.data_ptr()
        L.pq, L.aggx, L.y0, L.y1 = pq.data_ptr(), aggx.data_ptr(), y0.data_ptr(), (y1.data_ptr() if n_y1 else None)
        L.z, L.zmask = (None, z.data_ptr()) if use_mask else (z.data_ptr(), None)
        L.save_mean, L.save_invstd, L.out = mean.data_ptr(), invstd.data_ptr(), out.data_ptr()
        if nbt is not None:                              # (the counters ride in the statistics' finalize kernel: ABI 28)
            L.num_batches_tracked, L.n_nbt = nbt.data_ptr(), nbt.numel()
        drop_mask = None
        if drop is not None:
            drop_mask = torch.empty(lib.dgn_dropout_mask_bytes(N * Fo), dtype=torch.uint8, device=dev)
            L.drop_p, L.drop_seed, L.drop_offset, L.drop_mask = drop[0], drop[1].data_ptr(), int(drop[2]), drop_mask.data_ptr()
            LAST_DROPOUT_MASK = drop_mask
        # what the backward sweep would recompute from the messages (first max / min slot, dx signs): one byte per (row, feature)
        n_aux = int(lib.dgn_towers_layer_agg_aux_bytes(C.byref(L))) if AGG_AUX else 0
        aux = torch.empty(n_aux, dtype=torch.uint8, device=dev) if n_aux else None
        L.agg_aux = _ptr(aux)
        nbytes = lib.dgn_towers_layer_forward_workspace_bytes(C.byref(L))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        L.ws, L.ws_bytes = ws.data_ptr(), nbytes
        ctx.n_valid = _N_VALID
        L.n_valid = _ptr(ctx.n_valid)
        stream = _lib.stream_ptr(dev)
        _lib.check(lib.dgn_towers_layer_forward(C.byref(L), stream), "dgn_towers_layer_forward")
        ctx.save_for_backward(w_edge, h, snorm, scale, w_sd, bias_sd, w_post, b_post, gamma, beta, w_mix, b_mix, saved_buf, aux, drop_mask)
        ctx.graph, ctx.plan, ctx.avg_log, ctx.cfg = graph, plan, avg_log, cfg
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        w_edge, h, snorm, scale, w_sd, bias_sd, w_post, b_post, gamma, beta, w_mix, b_mix, saved_buf, aux, drop_mask = ctx.saved_tensors
        graph, plan = ctx.graph, ctx.plan
        T, fi, fo, S, residual, momentum, eps, slope, drop = ctx.cfg[:9]
        id_slot = ctx.cfg[9] if len(ctx.cfg) > 9 else None
        N, Fm, Fo = h.shape[0], T * fi, T * fo
        K = plan.n_agg * fi
        dev = h.device
        sizes = [N * 2 * Fm, T * N * K, N * Fo, ctx.n_y1, ctx.n_z, Fo, Fo]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) & ~63
        pq, aggx, y0, y1, z, mean, invstd = (saved_buf[o:o + n] for o, n in zip(offs, sizes))
        g_out = g_out.contiguous()
        graph.ensure_csc()
        graph.ensure_blocks(bool(BLOCK_BACKWARD))
        spec = _spec_structs(plan, T, ctx.avg_log, N * K)[0]
        L = _lib.DgnTowersLayer()
        cg = graph.c_graph
        L.graph, L.spec = C.pointer(cg), C.pointer(spec)
        L.w, L.ld_w, L.log_deg = _ptr(w_edge), (w_edge.stride(0) if w_edge is not None else 0), graph.log_deg.data_ptr()
        L.n_towers, L.f_in, L.f_out, L.n_scalers, L.residual = T, fi, fo, S, int(residual)
        L.momentum, L.eps, L.slope = float(momentum), float(eps), float(slope)
        L.h, L.snorm, L.scale = h.data_ptr(), _ptr(snorm), _ptr(scale)
        L.w_sd, L.bias_sd, L.w_post, L.b_post = w_sd.data_ptr(), bias_sd.data_ptr(), w_post.data_ptr(), b_post.data_ptr()
        L.bn_gamma, L.bn_beta = gamma.data_ptr(), beta.data_ptr()
        L.w_mix, L.b_mix = w_mix.data_ptr(), b_mix.data_ptr()
        L.pq, L.aggx, L.y0, L.y1 = pq.data_ptr(), aggx.data_ptr(), y0.data_ptr(), (y1.data_ptr() if ctx.n_y1 else None)
        L.z, L.zmask = (None, z.data_ptr()) if ctx.use_mask else (z.data_ptr(), None)
        L.save_mean, L.save_invstd = mean.data_ptr(), invstd.data_ptr()
        L.agg_aux = _ptr(aux)
        L.id_slot1 = 0 if id_slot is None else int(id_slot) + 1
        if drop is not None:
            L.drop_p, L.drop_mask = drop[0], drop_mask.data_ptr()
        nbytes = lib.dgn_towers_layer_backward_workspace_bytes(C.byref(L))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        L.ws, L.ws_bytes = ws.data_ptr(), nbytes
        L.n_valid = _ptr(ctx.n_valid)
        g_h = torch.empty((N, Fm), dtype=torch.float32, device=dev)
        # the six operand gradients back to back (no padding) in the order of DGNLayerTower._assemble's flat operand buffer: _AssembleOperands.backward
        # finds them adjacent and gathers the parameter gradients out of this buffer without a concatenation
        n_ops = [2 * Fm * Fm, 2 * Fm, Fo, Fo, Fo, T * S * fo * K]
        n_flat = (sum(n_ops) + 63) & ~63
        gbuf = torch.empty(n_flat + ((Fo * Fo + 63) & ~63) + Fo, dtype=torch.float32, device=dev)
        g_w_sd, g_bias_sd, g_b_post, g_gamma, g_beta, g_w_post = gbuf[:sum(n_ops)].split(n_ops)
        g_w_mix, g_b_mix = gbuf[n_flat:n_flat + Fo * Fo], gbuf[n_flat + ((Fo * Fo + 63) & ~63):]
        G = _lib.DgnTowersGrads(g_out=g_out.data_ptr(), g_h=g_h.data_ptr(), g_w_sd=g_w_sd.data_ptr(), g_bias_sd=g_bias_sd.data_ptr(),
                                g_w_post=g_w_post.data_ptr(), g_b_post=g_b_post.data_ptr(), g_gamma=g_gamma.data_ptr(),
                                g_beta=g_beta.data_ptr(), g_w_mix=g_w_mix.data_ptr(), g_b_mix=g_b_mix.data_ptr())
        stream = _lib.stream_ptr(dev)
        _lib.check(lib.dgn_towers_layer_backward(C.byref(L), C.byref(G), stream), "dgn_towers_layer_backward")
        return (None, None, None, None, None, g_h, None, None, None, None, None, g_w_sd.view(2 * Fm, Fm), g_bias_sd, g_w_post.view(T, S * fo, K),
                g_b_post, g_gamma, g_beta, g_w_mix.view(Fo, Fo), g_b_mix)


def towers_layer(graph: DGNGraph, plan: AggPlan, avg_log: float, w_edge, h, snorm, scale, running_mean, running_var, num_batches_tracked,
                 w_sd, bias_sd, w_post, b_post, gamma, beta, w_mix, b_mix, n_towers: int, f_in: int, f_out: int, residual: bool,
                 momentum: float, eps: float, slope: float, dropout=None, id_slot=None) -> torch.Tensor:
    """``DGNLayerTower.forward`` (nets/dgn_layer.py:309-325) of the fused configuration as one autograd node: see
    ``include/dgn_hip.h: DgnTowersLayer`` for the operand layouts (those of ``DGNLayerTower._assemble``) and the sequence of
    kernels.  Training mode; the BatchNorm running statistics and ``num_batches_tracked`` are updated in place.
    ``dropout``: None, or ``(p, key tensor, offset)`` of the towers' F.dropout (:275) as ``ops.dropout`` takes them.  ``id_slot``: the
    index of the identity scaler among the columns of ``scale`` (None: unknown; DgnTowersLayer.id_slot1)."""
    S = 1 if scale is None else scale.shape[1]
    if snorm is not None:
        snorm = snorm.reshape(-1).contiguous()
    if scale is not None:
        scale = scale.contiguous()
    drop = None if dropout is None else (float(dropout[0]), dropout[1], int(dropout[2]))
    cfg = (n_towers, f_in, f_out, S, bool(residual), float(momentum), float(eps), float(slope), drop, (0 if S == 1 else id_slot))
    nbt = num_batches_tracked
    if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64 and nbt.is_contiguous() and nbt.numel() <= 256):
        with torch.no_grad():
            nbt.add_(1)
        nbt = None
    return _TowersLayer.apply(graph, plan, float(avg_log), w_edge, cfg, h, snorm, scale, running_mean, running_var, nbt,
                              w_sd, bias_sd, w_post.contiguous(), b_post, gamma, beta, w_mix, b_mix)


# ---- the whole simple / complex layer as ONE autograd node over two C calls (dgn_layers.hip) ----------------------------------------

_DENSE_OK = {}


def dense_layer_supported(type_net: int, f_in: int, f_out: int, n_scalers: int, n_agg: int) -> bool:
    key = (type_net, f_in, f_out, n_scalers, n_agg)
    if key not in _DENSE_OK:
        _DENSE_OK[key] = bool(_lib.load().dgn_dense_layer_supported(*key))
    return _DENSE_OK[key]


# True: simple / complex layers with several degree scalers run posttrans as ONE f_out-column product per in-degree class
# (dgn_dc_kernels.hpp) instead of the folded S * f_out-column product + scale-combine: a third of the MFMA flops in the forward, the
# input gradient and the weight gradient.  Graphs with an in-degree >= 32 (and padded graphs) keep the folded route.
DC_POSTTRANS = os.environ.get("DGN_DC_POSTTRANS", "1") != "0"
# ... from this many nodes on: building the virtual row space costs a sort, a dozen small launches and one read-back per GRAPH, which a
# training loop pays per batch -- at the reference's batch 128 (3 000 nodes) the products it speeds up are ~10 us each
DC_MIN_NODES = int(os.environ.get("DGN_DC_MIN_NODES", "16384"))


def _degree_classes(graph, scale, fo, K):
    """(graph.degree_classes(), class scaler table [32, S]) or None."""
    if not DC_POSTTRANS or scale is None or not scale.is_cuda or graph.num_nodes < DC_MIN_NODES:
        return None
    lib = _lib.load()
    if not (lib.dgn_dc_supported(K, fo) and lib.dgn_dc_supported(fo, K) and lib.dgn_dc_wgrad_supported(K, fo)):
        return None
    dc = graph.degree_classes()
    if dc is None:
        return None
    cache = graph.__dict__.setdefault("_dc_scale", {})
    key = scale.data_ptr()
    if key not in cache:
        cache[key] = (scale, scale.index_select(0, dc["rep"]).contiguous())      # (the per-node table is kept alive with its class rows)
    return dc, cache[key][1]


def _dc_struct(dc):
    if dc is None:
        return None
    g, cls_scale = dc
    return _lib.DgnDegreeClasses(n_units=g["n_units"], vperm=g["vperm"].data_ptr(), unit_class=g["unit_class"].data_ptr(),
                                 present=g["present"].data_ptr(), scale=cls_scale.data_ptr())


# Graphs WITH hub rows (in-degree >= DGN_DC_CLASSES: power-law graphs), no gradients recorded: the rows below that in-degree take the
# degree-class product, the hub rows the folded product on their gathered aggregate rows.  C5 (10 M rows, 6 % hubs, 8 aggregators x 3
# scalers, hidden 128): 2 N_low K f_out + 2 N_hub K S f_out flops instead of 2 N K S f_out (0.37x).
DC_SPLIT = os.environ.get("DGN_DC_SPLIT", "1") != "0"


def dc_posttrans_split_supported(graph: DGNGraph, agg: torch.Tensor, fo: int, S: int) -> bool:
    """Inference (no gradients recorded): any graph with degree classes.  Training (round 6): graphs WITH hub rows -- the whole-layer call
    refuses those and its folded product is three times the flops; graphs without hubs keep the whole-layer degree-class call."""
    if not (DC_SPLIT and DC_POSTTRANS and S > 1 and agg.is_cuda and agg.dtype == torch.float32
            and graph.num_nodes >= DC_MIN_NODES and getattr(graph, "_pad", None) is None and graph.num_src == graph.num_nodes):
        return False
    lib = _lib.load()
    K = agg.shape[1]
    if not lib.dgn_dc_supported(K, fo):
        return False
    if torch.is_grad_enabled() and agg.requires_grad:
        return bool(DC_SPLIT_TRAINING and graph.n_hub_rows_dc() > 0 and lib.dgn_dc_supported(fo, K) and lib.dgn_dc_wgrad_supported(K, fo))
    return True


DC_SPLIT_TRAINING = os.environ.get("DGN_DC_SPLIT_TRAINING", "1") != "0"


def split_training_route(graph: DGNGraph, S: int) -> bool:
    """Whether a TRAINING step of a simple / complex layer leaves the whole-layer call for the per-op route with the split degree-class
    posttrans: a graph with hub rows (the whole-layer call would run the folded S f_out-column product on every row)."""
    return bool(DC_SPLIT and DC_SPLIT_TRAINING and DC_POSTTRANS and S > 1 and graph.num_nodes >= DC_MIN_NODES and getattr(graph, "_pad", None) is None
                and graph.num_src == graph.num_nodes and graph.degree_classes() is None and graph.n_hub_rows_dc() > 0)


class _DcClassRows(torch.autograd.Function):
    """The class rows' share of ``dc_posttrans_split`` with its backward (round 6: training on graphs with hub rows): forward
    dgn_dc_fold + dgn_dc_gemm; backward d agg = (row_scale g) W_class (dgn_dc_gemm on the transposed class weights), d W in the
    reference's layout by dgn_dc_wgrad, d bias = column sums.  Hub rows of y are left to the caller (index_copy), their rows of d agg
    are zero here."""

    @staticmethod
    def forward(ctx, graph, agg, weight, bias, cls_scale, row_scale, n_agg, f_in, id_slot):
        lib = _lib.load()
        dc = graph.degree_classes_split()
        N, K = agg.shape
        fo, S = weight.shape[0], cls_scale.shape[1]
        cx = id_slot >= 0
        f_pad = K // (n_agg + (1 if cx else 0))
        stream = _lib.stream_ptr(agg.device)
        y = torch.empty(N, fo, dtype=torch.float32, device=agg.device)
        s = _lib.DgnDegreeClasses(n_units=dc["n_units"], vperm=dc["vperm"].data_ptr(), unit_class=dc["unit_class"].data_ptr(),
                                  present=dc["present"].data_ptr(), scale=cls_scale.data_ptr())
        wc = torch.empty(2, _lib.DGN_DC_CLASSES, fo * K, dtype=torch.float32, device=agg.device)
        lay = _lib.DgnDcLayout(n_agg=n_agg, f_pad=f_pad, f_in=f_in, h_off=f_in if cx else 0, id_slot=id_slot if cx else -1, ld=weight.stride(0))
        _lib.check(lib.dgn_dc_fold(C.byref(s), S, fo, K, 1, weight.data_ptr(), C.byref(lay), wc[0].data_ptr(), wc[1].data_ptr(), stream), "dgn_dc_fold")
        _lib.check(lib.dgn_dc_gemm(C.byref(s), K, fo, 1, agg.data_ptr(), agg.stride(0), 0, wc[0].data_ptr(), K, fo * K, 0, _ptr(bias), _ptr(row_scale),
                                   y.data_ptr(), fo, 0, 0, stream), "dgn_dc_gemm")
        ctx.save_for_backward(agg, weight, cls_scale, row_scale, wc)
        ctx.meta = (graph, n_agg, f_in, id_slot, f_pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        lib = _lib.load()
        agg, weight, cls_scale, row_scale, wc = ctx.saved_tensors
        graph, n_agg, f_in, id_slot, f_pad, has_bias = ctx.meta
        dc = graph.degree_classes_split()
        N, K = agg.shape
        fo, S = weight.shape[0], cls_scale.shape[1]
        cx = id_slot >= 0
        stream = _lib.stream_ptr(agg.device)
        g = g_y.contiguous() if row_scale is None else (g_y * row_scale.unsqueeze(1)).contiguous()      # (hub rows arrive as zeros: index_copy's adjoint)
        s = _lib.DgnDegreeClasses(n_units=dc["n_units"], vperm=dc["vperm"].data_ptr(), unit_class=dc["unit_class"].data_ptr(),
                                  present=dc["present"].data_ptr(), scale=cls_scale.data_ptr())
        g_agg = g_w = g_b = None
        if ctx.needs_input_grad[1]:
            g_agg = torch.empty(N, K, dtype=torch.float32, device=agg.device)
            hub = dc["hub_rows"]
            if hub.numel():
                g_agg.index_fill_(0, hub, 0.0)                  # (dgn_dc_gemm writes the rows the virtual row space names)
            _lib.check(lib.dgn_dc_gemm(C.byref(s), fo, K, 1, g.data_ptr(), fo, 0, wc[1].data_ptr(), fo, fo * K, 0, None, None, g_agg.data_ptr(), K, 0, 0,
                                       stream), "dgn_dc_gemm")
        if ctx.needs_input_grad[2]:
            g_w = torch.zeros_like(weight)
            lay = _lib.DgnDcLayout(n_agg=n_agg, f_pad=f_pad, f_in=f_in, h_off=f_in if cx else 0, id_slot=id_slot if cx else -1, ld=g_w.stride(0))
            nbytes = lib.dgn_dc_wgrad_workspace_bytes(dc["n_units"], K, fo)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=agg.device)
            _lib.check(lib.dgn_dc_wgrad(C.byref(s), S, K, fo, g.data_ptr(), fo, agg.data_ptr(), agg.stride(0), g_w.data_ptr(), g_w.stride(0), C.byref(lay),
                                        ws.data_ptr(), nbytes, stream), "dgn_dc_wgrad")
        if has_bias and ctx.needs_input_grad[3]:
            g_b = g.sum(0)
        return None, g_agg, g_w, g_b, None, None, None, None, None


def dc_posttrans_split(graph: DGNGraph, agg: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], scale: torch.Tensor,
                       row_scale: Optional[torch.Tensor], n_agg: int, f_in: int, id_slot: int = -1) -> torch.Tensor:
    """``snorm * posttrans(cat_s(scale_s * agg))`` (nets/dgn_layer.py:186-193 with the scalers of scalers.py:7-18) WITHOUT gradients on a
    graph that may hold hub rows: ``agg [N, n_agg * f_pad]`` the scaler-free aggregates, ``weight [f_out, S * n_agg * f_in]`` the
    posttrans Linear in the reference's layout, ``scale [N, S]`` the per-node scaler table.  Rows of in-degree < DGN_DC_CLASSES:
    dgn_dc_fold + dgn_dc_gemm (one f_out-column product per in-degree class); hub rows: gathered, folded product (S f_out columns),
    scale-combine, written back by row index.
    ``id_slot >= 0``: the complex layer (:116-122): ``agg [N, (n_agg + 1) * f_pad]`` ends with the h_in pass-through block, ``weight`` is
    ``[f_out, f_in + S * n_agg * f_in]`` and its h columns act through the identity scaler's slot ``id_slot``."""
    lib = _lib.load()
    cx = id_slot >= 0
    dc = graph.degree_classes_split()
    if dc is None:
        raise RuntimeError("dc_posttrans_split: the graph has no degree classes (padded or bipartite)")
    N, K = agg.shape
    fo, S = weight.shape[0], scale.shape[1]
    f_pad = K // (n_agg + (1 if cx else 0))
    weight = weight.contiguous()
    row_scale = None if row_scale is None else row_scale.reshape(-1).contiguous()
    training = torch.is_grad_enabled() and (agg.requires_grad or weight.requires_grad)
    if training:
        # training (round 6): the class rows through _DcClassRows (its own backward), the hub rows through the differentiable folded product
        hub = dc["hub_rows"]
        y = None
        if dc["n_units"] > 0:
            cls_scale = scale.index_select(0, dc["rep"]).contiguous()
            y = _DcClassRows.apply(graph, agg.contiguous(), weight, bias, cls_scale, row_scale, n_agg, f_in, id_slot)
        if hub.numel():
            y_hub = _split_hub_rows(agg, weight, bias, scale, row_scale, hub, n_agg, f_in, f_pad, id_slot, fo, S, K)
            y = y_hub.new_zeros(N, fo).index_copy(0, hub, y_hub) if y is None else y.index_copy(0, hub, y_hub)
        return y
    y = torch.empty(N, fo, dtype=torch.float32, device=agg.device)
    stream = _lib.stream_ptr(agg.device)
    if dc["n_units"] > 0:
        cls_scale = scale.index_select(0, dc["rep"]).contiguous()
        s = _lib.DgnDegreeClasses(n_units=dc["n_units"], vperm=dc["vperm"].data_ptr(), unit_class=dc["unit_class"].data_ptr(),
                                  present=dc["present"].data_ptr(), scale=cls_scale.data_ptr())
        wc = torch.empty(2, _lib.DGN_DC_CLASSES, fo * K, dtype=torch.float32, device=agg.device)
        lay = _lib.DgnDcLayout(n_agg=n_agg, f_pad=f_pad, f_in=f_in, h_off=f_in if cx else 0, id_slot=id_slot if cx else -1, ld=weight.stride(0))
        _lib.check(lib.dgn_dc_fold(C.byref(s), S, fo, K, 1, weight.data_ptr(), C.byref(lay), wc[0].data_ptr(), wc[1].data_ptr(), stream), "dgn_dc_fold")
        _lib.check(lib.dgn_dc_gemm(C.byref(s), K, fo, 1, agg.data_ptr(), agg.stride(0), 0, wc[0].data_ptr(), K, fo * K, 0, _ptr(bias), _ptr(row_scale),
                                   y.data_ptr(), fo, 0, 0, stream), "dgn_dc_gemm")
    hub = dc["hub_rows"]
    if hub.numel():
        y.index_copy_(0, hub, _split_hub_rows(agg, weight, bias, scale, row_scale, hub, n_agg, f_in, f_pad, id_slot, fo, S, K))
    return y


def _split_hub_rows(agg, weight, bias, scale, row_scale, hub, n_agg, f_in, f_pad, id_slot, fo, S, K):
    """The hub rows of dc_posttrans_split: gathered aggregate rows, the folded product (S f_out columns) and the scale-combine -- all
    differentiable ops (index_select, node_linear, scale_combine)."""
    cx = id_slot >= 0
    pad = (lambda t: torch.nn.functional.pad(t, (0, f_pad - f_in)) if f_pad != f_in else t)
    w = pad(weight[:, f_in if cx else 0:].reshape(fo, S * n_agg, f_in)).reshape(fo, S, n_agg * f_pad).permute(1, 0, 2)      # [S, fo, A f_pad]
    if cx:      # the h columns in the identity scaler's block, zero elsewhere (dgn_layer._folded_weight)
        hcols = [torch.zeros(fo, f_pad, dtype=w.dtype, device=w.device) for _ in range(S)]
        hcols[id_slot] = pad(weight[:, :f_in])
        w = torch.cat([w, torch.stack(hcols, dim=0)], dim=2)
    w = w.reshape(S * fo, K).contiguous()
    z = node_linear(agg.index_select(0, hub), w)
    return scale_combine(z.unsqueeze(0), scale.index_select(0, hub), bias, None if row_scale is None else row_scale.index_select(0, hub))


def _dense_sizes(cfg, N, dc=False):
    type_net, F0, fo, S, A = cfg[:5]
    Fp = F0 + (F0 & 1)
    K = (A + (1 if type_net == 1 else 0)) * Fp
    return [N * Fp if Fp != F0 else 0, N * 2 * Fp if type_net == 1 else 0, N * K, N * fo, (2 * S + (2 * _lib.DGN_DC_CLASSES if dc else 0)) * fo * K,
            (4 * Fp * Fp + 2 * Fp) if type_net == 1 else 0, fo, fo], Fp, K


def _dense_struct(graph, plan, avg_log, w_edge, cfg, h, snorm, scale, w_pre, b_pre, w_post, b_post, gamma, beta, bufs, n_valid, dc=None):
    type_net, F0, fo, S, A, id_slot, residual, momentum, eps = cfg
    hp, pq, agg, y, wf, wsd, mean, invstd = bufs
    spec = _spec_structs(plan, 1, avg_log, 0)[0]
    L = _lib.DgnDenseLayer()
    cg = graph.c_graph
    L.graph, L.spec = C.pointer(cg), C.pointer(spec)
    L.w, L.ld_w, L.log_deg = _ptr(w_edge), (w_edge.stride(0) if w_edge is not None else 0), graph.log_deg.data_ptr()
    L.type, L.f_in, L.f_out, L.n_scalers, L.n_agg, L.id_slot, L.residual = type_net, F0, fo, S, A, max(id_slot, 0), int(residual)
    L.momentum, L.eps = float(momentum), float(eps)
    L.h, L.snorm, L.scale = h.data_ptr(), _ptr(snorm), _ptr(scale)
    L.w_pre, L.b_pre, L.w_post, L.b_post = _ptr(w_pre), _ptr(b_pre), w_post.data_ptr(), _ptr(b_post)
    L.bn_gamma, L.bn_beta = gamma.data_ptr(), beta.data_ptr()
    L.hp, L.pq, L.agg, L.y, L.wf, L.wsd, L.save_mean, L.save_invstd = hp, pq, agg, y, wf, wsd, mean, invstd          # (addresses, _carve_ptrs)
    L.n_valid = _ptr(n_valid)
    dcs = _dc_struct(dc)
    if dcs is not None:
        L.dc = C.pointer(dcs)
    return L, (cg, spec, dcs)


class _DenseLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, plan, avg_log, w_edge, cfg, h, snorm, scale, running_mean, running_var, nbt, w_pre, b_pre, w_post, b_post, gamma, beta):
        lib = _lib.load()
        if not h.is_cuda:
            raise _lib.DgnError("dense_layer: CUDA tensors only (dgn_amd has no CPU path)")
        N, fo = h.shape[0], cfg[2]
        dev = h.device
        h, w_post, gamma, beta = h.contiguous(), w_post.contiguous(), gamma.contiguous(), beta.contiguous()
        w_pre = w_pre.contiguous() if w_pre is not None else None
        _, _, K = _dense_sizes(cfg, 0)
        dc = _degree_classes(graph, scale, fo, K)
        sizes, Fp, K = _dense_sizes(cfg, N, dc is not None)
        saved_buf, bufs = _carve_ptrs(sizes, dev)
        out = torch.empty((N, fo), dtype=torch.float32, device=dev)
        ctx.n_valid = _N_VALID
        ctx.dc = dc
        L, keep = _dense_struct(graph, plan, avg_log, w_edge, cfg, h, snorm, scale, w_pre, b_pre, w_post, b_post, gamma, beta, bufs, ctx.n_valid, dc)
        L.running_mean, L.running_var, L.out = running_mean.data_ptr(), running_var.data_ptr(), out.data_ptr()
        L.num_batches_tracked = _ptr(nbt)                # (the counter rides in the statistics' finalize kernel: ABI 28)
        n_aux = int(lib.dgn_dense_layer_agg_aux_bytes(C.byref(L))) if AGG_AUX else 0      # (see _TowersLayer)
        aux = torch.empty(n_aux, dtype=torch.uint8, device=dev) if n_aux else None
        L.agg_aux = _ptr(aux)
        nbytes = lib.dgn_dense_layer_forward_workspace_bytes(C.byref(L))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        L.ws, L.ws_bytes = ws.data_ptr(), nbytes
        stream = _lib.stream_ptr(dev)
        _lib.check(lib.dgn_dense_layer_forward(C.byref(L), stream), "dgn_dense_layer_forward")
        ctx.save_for_backward(w_edge, h, snorm, scale, w_pre, b_pre, w_post, b_post, gamma, beta, saved_buf, aux)
        ctx.graph, ctx.plan, ctx.avg_log, ctx.cfg = graph, plan, avg_log, cfg
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        w_edge, h, snorm, scale, w_pre, b_pre, w_post, b_post, gamma, beta, saved_buf, aux = ctx.saved_tensors
        cfg, graph = ctx.cfg, ctx.graph
        type_net, F0, fo = cfg[:3]
        N, dev = h.shape[0], h.device
        sizes, Fp, K = _dense_sizes(cfg, N, ctx.dc is not None)
        _, bufs = _carve_ptrs(sizes, dev, saved_buf)
        g_out = g_out.contiguous()
        graph.ensure_csc()
        graph.ensure_blocks(bool(BLOCK_BACKWARD))
        L, keep = _dense_struct(graph, ctx.plan, ctx.avg_log, w_edge, cfg, h, snorm, scale, w_pre, b_pre, w_post, b_post, gamma, beta, bufs, ctx.n_valid,
                                ctx.dc)
        L.agg_aux = _ptr(aux)
        nbytes = lib.dgn_dense_layer_backward_workspace_bytes(C.byref(L))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        L.ws, L.ws_bytes = ws.data_ptr(), nbytes
        g_h = torch.empty((N, F0), dtype=torch.float32, device=dev)
        n_pre = w_pre.numel() if w_pre is not None else 0
        _, (g_w_pre, g_b_pre, g_w_post, g_b_post, g_gamma, g_beta) = _carve([n_pre, F0 if (type_net == 1 and b_pre is not None) else 0, w_post.numel(), fo, fo, fo], dev)
        q = lambda t: t.data_ptr() if t.numel() else None
        G = _lib.DgnDenseGrads(g_out=g_out.data_ptr(), g_h=g_h.data_ptr(), g_w_pre=q(g_w_pre), g_b_pre=q(g_b_pre), g_w_post=g_w_post.data_ptr(),
                               g_b_post=g_b_post.data_ptr(), g_gamma=g_gamma.data_ptr(), g_beta=g_beta.data_ptr())
        stream = _lib.stream_ptr(dev)
        _lib.check(lib.dgn_dense_layer_backward(C.byref(L), C.byref(G), stream), "dgn_dense_layer_backward")
        return (None, None, None, None, None, g_h, None, None, None, None, None,
                g_w_pre.view_as(w_pre) if w_pre is not None else None, g_b_pre if (b_pre is not None and type_net == 1) else None,
                g_w_post.view_as(w_post), g_b_post if b_post is not None else None, g_gamma, g_beta)


def dense_layer(graph: DGNGraph, plan: AggPlan, avg_log: float, w_edge, h, snorm, scale, bn, w_pre, b_pre, w_post, b_post, type_net: int, n_agg: int,
                id_slot: int, residual: bool) -> torch.Tensor:
    """``DGNLayerSimple.forward`` / ``DGNLayerComplex.forward`` (nets/dgn_layer.py:178-202, :103-132) of the fused configuration as one
    autograd node over ``dgn_dense_layer_forward / _backward`` (``include/dgn_hip.h: DgnDenseLayer``): odd hidden sizes are padded, the
    posttrans weight is folded and its gradient un-folded inside the calls; parameters and their gradients keep the reference's layout.
    ``plan``: the sweep's list (with the h_in block last for the complex layer), identity scaler.  Training mode; the BatchNorm running
    statistics and ``num_batches_tracked`` are updated in place."""
    S = 1 if scale is None else scale.shape[1]
    if snorm is not None:
        snorm = snorm.reshape(-1).contiguous()
    if scale is not None:
        scale = scale.contiguous()
    cfg = (int(type_net), h.shape[1], w_post.shape[0], S, int(n_agg), int(id_slot), bool(residual), float(bn.momentum), float(bn.eps))
    nbt = bn.num_batches_tracked
    if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1):
        with torch.no_grad():
            nbt.add_(1)
        nbt = None
    return _DenseLayer.apply(graph, plan, float(avg_log), w_edge, cfg, h, snorm, scale, bn.running_mean, bn.running_var, nbt, w_pre, b_pre, w_post, b_post,
                             bn.weight, bn.bias)


# ---- the graph-block layer: a batch at the reference's batch size as five launches per step (csrc/dgn_blk_layer.hip) -----------------

# Batches up to this many nodes take the graph-block route where it applies (whole graphs per workgroup, everything out of LDS; the
# streaming kernels are built for batches that fill the chip).  0 switches the route off.
# Measured captured steps, route vs streaming (tools/route_crossover.py): towers x 128 / 256 / 512 / 1024 graphs (3.0 k / 5.8 k / 11.9 k / 23.8 k
# nodes) 0.103 / 0.154 / 0.320 / 0.658 vs 0.182 / 0.194 / 0.217 / 0.270 ms; ZINC json complex 45: 0.101 / 0.111 / 0.210 / 0.410 vs 0.150 / 0.162 /
# 0.184 / 0.193 -- the route's time grows with the rounds of (block, tower) workgroups, the streaming kernels' barely at all.
BLOCK_LAYER_MAX_NODES = int(os.environ.get("DGN_BLOCK_LAYER_MAX_NODES", "8192"))
# ... and only layers whose posttrans is small: a (block, tower) workgroup multiplies its rows by the tower's whole [f_out, (f_in +) S A f_in]
# weight three times per step (forward, input gradient, weight gradient) with the scalers applied on the fly, and a batch of 128 graphs is
# 128 workgroups per tower.  Measured captured steps at batch 128, block route vs streaming kernels: towers 5 x (224 x 14 = 3.1 k weights)
# 0.103 vs 0.174 ms, ZINC json complex 45 (450 x 45 = 20 k) 0.100 vs 0.150, simple 75 (450 x 75 = 34 k) 0.115 vs 0.119, HIV json simple 70 with
# five aggregators x three scalers (1050 x 70 = 74 k) 0.236 vs 0.149 -- the streaming route has the degree-class product (S times fewer
# flops) and all CUs.
BLOCK_LAYER_MAX_POST = int(os.environ.get("DGN_BLOCK_LAYER_MAX_POST", "40960"))
_BLK_DBG = None      # tests: dict that receives the aggregate rows / their gradients of the next call
# Opt-in (False by default; also DGN_DIRECT_PARAM_GRADS=1): on the graph-block route the layer's parameters are NOT inputs of the autograd
# node -- its backward assigns / accumulates their `.grad` itself.  A towers layer has 33 parameters; autograd's per-output work (gradient
# validation + one AccumulateGrad node each, ~3.7 us apiece) is 0.12 ms of a step whose GPU side is 0.10 ms.  What is given up: parameter
# hooks and anything built on AccumulateGrad (DistributedDataParallel: use dist.FlatGradAllReduce), `torch.autograd.grad(..., params)`,
# double backward.  The gradients themselves are the same tensors, bit for bit.
DIRECT_PARAM_GRADS = os.environ.get("DGN_DIRECT_PARAM_GRADS", "0") == "1"


def _block_struct(graph, table, plan, avg_log, eig, cfg, h, snorm, rm, rv, nbt, params):
    """DgnBlockLayer for one call (+ the ctypes objects it points to)."""
    type_net, T, fi, fo, residual, momentum, eps, slope = cfg[:8]
    from .graph import _channel_array
    spec = _spec_structs(plan, 1, avg_log, 0)[0]
    L = _lib.DgnBlockLayer()
    cg, tb = graph.c_graph, table["struct"]
    chans = _channel_array(plan.channels) if plan.n_channels else None
    L.graph, L.blocks, L.spec = C.pointer(cg), C.pointer(tb), C.pointer(spec)
    if chans is not None:
        L.channels = C.cast(chans, C.POINTER(_lib.DgnChannel))
        L.eig, L.ld_eig, L.n_eig_cols = eig.data_ptr(), eig.stride(0), eig.shape[1]
    L.log_deg = graph.log_deg.data_ptr()
    L.type, L.n_towers, L.f_in, L.f_out, L.residual = type_net, T, fi, fo, int(residual)
    L.momentum, L.eps, L.slope = float(momentum), float(eps), float(slope)
    L.h, L.snorm = h.data_ptr(), _ptr(snorm)
    per = 6 if type_net != 0 else 4
    arr = C.c_void_p * T
    cols = [arr(*[params[t * per + q].data_ptr() for t in range(T)]) for q in range(per)]
    if type_net != 0:
        L.w_pre, L.b_pre, L.w_post, L.b_post, L.gamma, L.beta = cols
    else:
        L.w_post, L.b_post, L.gamma, L.beta = cols
    if type_net == 2:
        L.w_mix, L.b_mix = params[T * per].data_ptr(), params[T * per + 1].data_ptr()
    L.running_mean, L.running_var = _ptr(rm), _ptr(rv)
    if nbt is not None:
        L.num_batches_tracked, L.n_nbt = nbt.data_ptr(), nbt.numel()
    # padded batches (hipgraph.PaddedBatch): the valid-row count the layer announced, the static table's overflow flag
    n_valid = _N_VALID
    if n_valid is not None:
        L.n_valid = n_valid.data_ptr()
    if table.get("overflow") is not None:
        L.overflow = table["overflow"].data_ptr()
    # (the table's descriptor tensor is kept with the struct: graph.invalidate_caches() between a forward and its backward must not
    #  free the memory the struct's raw pointer names)
    return L, (cg, tb, spec, chans, cols, n_valid, table.get("desc"), table.get("overflow"))


def _plan_signature(plan):
    """What of an AggPlan the graph-block route's LDS plan, workspace sizes and parameter-gradient layout depend on."""
    return (tuple(plan.aggregators), tuple(plan.scalers))


# Evaluation forward (eval() under no_grad) of batches ABOVE BLOCK_LAYER_MAX_NODES: taken when every (block, tower) workgroup has a CU of
# its own (one round) -- 128 k-NN / SBM graphs of 85-190 nodes (CIFAR10 / PATTERN at the json's batch size: 12-15 k nodes) --, up to this
# many nodes.  The training limit above was measured on molecule batches, whose block count grows with the node count.
BLOCK_LAYER_EVAL_MAX_NODES = int(os.environ.get("DGN_BLOCK_LAYER_EVAL_MAX_NODES", "32768"))


def block_layer_supported(graph, plan, type_net, T, fi, fo, eval_only: bool = False) -> bool:
    """Whether this (batch, layer shape) runs on the graph-block route: a block table whose largest block fits the LDS plan
    (``eval_only``: the forward plan alone, dgn_block_layer_supported with eval_mode set)."""
    if BLOCK_LAYER_MAX_NODES <= 0 or len(plan.launches) != 1 or plan.n_channels > 3:
        return False
    over = graph.num_nodes > BLOCK_LAYER_MAX_NODES
    if over and not (eval_only and graph.num_nodes <= BLOCK_LAYER_EVAL_MAX_NODES):
        return False
    if ((fi if type_net != 0 else 0) + plan.n_scalers * plan.n_agg * fi) * fo > BLOCK_LAYER_MAX_POST:
        return False
    table = graph.block_table()
    if table is None:
        return False
    if over and table["n_blocks"] * T > 256:
        return False
    # (a stable signature of the plan, not id(plan): an id can be reused after garbage collection)
    key = (_plan_signature(plan), type_net, T, fi, fo, bool(eval_only))
    ok = table.setdefault("ok", {})
    if key not in ok:
        L = _lib.DgnBlockLayer()
        L.eval_mode = 1 if eval_only else 0
        spec = _spec_structs(plan, 1, 1.0, 0)[0]
        cg, tb = graph.c_graph, table["struct"]
        L.graph, L.blocks, L.spec = C.pointer(cg), C.pointer(tb), C.pointer(spec)
        L.type, L.n_towers, L.f_in, L.f_out = type_net, T, fi, fo
        if plan.n_channels:
            from .graph import _channel_array
            L.channels = C.cast(_channel_array(plan.channels), C.POINTER(_lib.DgnChannel))
            L.eig = 1      # (presence only)
        ok[key] = bool(_lib.load().dgn_block_layer_supported(C.byref(L)))
    return ok[key]


def _block_sizes(lib, table, L, cfg, params):
    """Per (batch, layer shape): workspace bytes, parameter-gradient floats and the split of the flat gradient buffer (cached on the
    block table: they depend on the block table and the widths only)."""
    # (the aggregator / scaler counts are part of the key: ld_post, the gradient floats and the workspaces depend on A x S)
    key = ("sizes",) + tuple(cfg[:4]) + (tuple(int(p_.numel()) for p_ in params),)
    ent = table.get(key)
    if ent is None:
        type_net, T, fi, fo = cfg[:4]
        per = 6 if type_net != 0 else 4
        sizes, shapes = [], []
        for t in range(T):
            for q in range(per - 2):
                sizes.append(params[t * per + q].numel()); shapes.append(tuple(params[t * per + q].shape))
        for p_ in params[T * per:]:
            sizes.append(p_.numel()); shapes.append(tuple(p_.shape))
        n_par = int(lib.dgn_block_layer_param_grad_floats(C.byref(L)))
        assert sum(sizes) == n_par, "block_layer: parameter shapes do not match the layer's widths"
        sizes += [fo] * (2 * T)
        ent = table[key] = (int(lib.dgn_block_layer_forward_workspace_bytes(C.byref(L))), int(lib.dgn_block_layer_backward_workspace_bytes(C.byref(L))),
                            n_par, sizes, shapes)
    return ent


class _BlockLayer(torch.autograd.Function):
    # The step is launch-bound on the host once the GPU side is ~0.1 ms: the struct built by the forward is kept for the backward, sizes
    # are cached per (batch, layer shape), scratch and saved tensors share allocations, the gradients leave as ONE split of a flat buffer.
    @staticmethod
    def forward(ctx, graph, plan, avg_log, eig, cfg, h, snorm, rm, rv, nbt, direct, *params):
        lib = _lib.load()
        # `direct` (DIRECT_PARAM_GRADS): the parameters as a TUPLE -- not inputs of the autograd node; the backward assigns their .grad
        ctx.direct = direct
        if direct is not None:
            params = direct
        type_net, T, fi, fo = cfg[:4]
        N, Fo, dev = h.shape[0], T * fo, h.device
        table = graph.block_table()
        h = h.contiguous()
        params = tuple(p if p.is_contiguous() else p.contiguous() for p in params)
        L, keep = _block_struct(graph, table, plan, avg_log, eig, cfg, h, snorm, rm, rv, nbt, params)
        ws_f, ws_b, n_par, sizes, shapes = _block_sizes(lib, table, L, cfg, params)
        drop = cfg[8] if len(cfg) > 8 else None
        if drop is not None:      # the towers' dropout inside the tails: (p, key tensor, offset); the keep bits are kept for the backward
            global LAST_DROPOUT_MASK
            drop_mask = torch.empty(lib.dgn_dropout_mask_bytes(N * Fo), dtype=torch.uint8, device=dev)
            L.drop_p, L.drop_seed, L.drop_offset, L.drop_mask = float(drop[0]), drop[1].data_ptr(), int(drop[2]), drop_mask.data_ptr()
            keep = keep + (drop_mask, drop[1])
            LAST_DROPOUT_MASK = drop_mask
        n_saved = N * Fo + 2 * Fo
        saved = torch.empty(n_saved + (ws_f + 3) // 4 + 64, dtype=torch.float32, device=dev)      # [y0 | mean | invstd | (forward scratch)]
        out = torch.empty((N, Fo), dtype=torch.float32, device=dev)
        base = saved.data_ptr()
        L.y0, L.save_mean, L.save_invstd, L.out = base, base + 4 * N * Fo, base + 4 * (N * Fo + Fo), out.data_ptr()
        L.ws, L.ws_bytes = (base + 4 * n_saved + 255) & ~255, ws_f
        dbg = _BLK_DBG
        if dbg is not None:
            dbg["agg"] = torch.zeros(N, T * plan.n_agg * fi, device=dev)
            dbg["t_fwd"] = torch.zeros(T * table["n_blocks"], 16, dtype=torch.int64, device=dev)
            L.dbg_agg, L.dbg_time = dbg["agg"].data_ptr(), dbg["t_fwd"].data_ptr()
        _lib.check(lib.dgn_block_layer_forward(C.byref(L), _lib.stream_ptr(dev)), "dgn_block_layer_forward")
        L.dbg_agg, L.dbg_time = None, None
        ctx.save_for_backward(h, snorm, eig, saved, *params)
        ctx.graph, ctx.cfg, ctx.call = graph, cfg, (L, keep, ws_b, n_par, sizes, shapes)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        h = ctx.saved_tensors[0]
        graph, cfg = ctx.graph, ctx.cfg
        L, keep, ws_b, n_par, sizes, shapes = ctx.call
        type_net, T, fi, fo = cfg[:4]
        N, F_, Fo, dev = h.shape[0], T * fi, T * fo, h.device
        g_out = g_out.contiguous()
        graph.ensure_csc()
        n_flat = n_par + 2 * Fo
        g_h = torch.empty((N, F_), dtype=torch.float32, device=dev)
        flat = torch.empty(n_flat + (ws_b + 3) // 4 + 64, dtype=torch.float32, device=dev)      # [parameter gradients | gamma | beta | (scratch)]
        fb = flat.data_ptr()
        L.ws, L.ws_bytes = (fb + 4 * n_flat + 255) & ~255, ws_b
        L.running_mean, L.running_var, L.num_batches_tracked, L.n_nbt, L.out = None, None, None, 0, None
        G = _lib.DgnBlockGrads(g_out=g_out.data_ptr(), g_h=g_h.data_ptr(), g_params=fb, g_gamma=fb + 4 * n_par, g_beta=fb + 4 * (n_par + Fo))
        dbg = _BLK_DBG
        if dbg is not None:
            dbg["t_bwd"] = torch.zeros(T * graph.block_table()["n_blocks"], 16, dtype=torch.int64, device=dev)
            L.dbg_time = dbg["t_bwd"].data_ptr()
        _lib.check(lib.dgn_block_layer_backward(C.byref(L), C.byref(G), _lib.stream_ptr(dev)), "dgn_block_layer_backward")
        # the gradients, in the order of `params`: ONE split of the flat buffer (+ a view per matrix)
        parts = flat[:n_flat].split(sizes)
        per = 6 if type_net != 0 else 4
        n_w = len(shapes)
        grads, q = [], 0
        for t in range(T):
            for _ in range(per - 2):
                grads.append(parts[q].view(shapes[q]) if len(shapes[q]) != 1 else parts[q])
                q += 1
            grads.append(parts[n_w + t])
            grads.append(parts[n_w + T + t])
        while q < n_w:
            grads.append(parts[q].view(shapes[q]) if len(shapes[q]) != 1 else parts[q])
            q += 1
        if ctx.direct is not None:
            for p_, g_ in zip(ctx.direct, grads):
                if p_.requires_grad:
                    p_.grad = g_ if p_.grad is None else p_.grad + g_
            return (None, None, None, None, None, g_h, None, None, None, None, None)
        return (None, None, None, None, None, g_h, None, None, None, None, None, *grads)


@torch.no_grad()
def _block_layer_eval(graph, plan, avg_log, eig, cfg, h, snorm, rm, rv, params):
    """The evaluation-mode forward (BatchNorm with the running statistics, nothing updated, no autograd node): two launches."""
    lib = _lib.load()
    T, fo = cfg[1], cfg[3]
    N, Fo, dev = h.shape[0], T * fo, h.device
    table = graph.block_table()
    h = h.contiguous()
    params = tuple(p if p.is_contiguous() else p.contiguous() for p in params)
    L, keep = _block_struct(graph, table, plan, avg_log, eig, cfg, h, snorm, rm, rv, None, params)
    L.eval_mode = 1
    ws_f = _block_sizes(lib, table, L, cfg, params)[0]
    y0 = torch.empty(N * Fo + (ws_f + 3) // 4 + 64, dtype=torch.float32, device=dev)       # [y0 | (forward scratch)]
    out = torch.empty((N, Fo), dtype=torch.float32, device=dev)
    L.y0, L.out = y0.data_ptr(), out.data_ptr()
    L.ws, L.ws_bytes = (y0.data_ptr() + 4 * N * Fo + 255) & ~255, ws_f
    _lib.check(lib.dgn_block_layer_forward(C.byref(L), _lib.stream_ptr(dev)), "dgn_block_layer_forward")
    return out


def block_layer(graph: DGNGraph, plan: AggPlan, avg_log: float, eig, h, snorm, rm, rv, nbt, params, type_net: int, n_towers: int, f_in: int,
                f_out: int, residual: bool, momentum: float, eps: float, slope: float = 0.01, training: bool = True, dropout=None) -> torch.Tensor:
    """One DGN layer (nets/dgn_layer.py:103-132 complex, :178-202 simple, :254-276 + :309-325 towers; training mode -- ``training=False``:
    the evaluation-mode forward without gradients, BatchNorm on its running statistics) as ONE autograd node
    over ``dgn_block_layer_forward / _backward`` (``include/dgn_hip.h: DgnBlockLayer``): two launches forward, three backward.
    ``plan``: the layer's own list (aggregators x applied scalers, no pass-through block); ``params``: per tower (pretrans weight, bias --
    complex / towers --, posttrans weight, bias, BatchNorm weight, bias), then (towers) the mixing network's weight and bias, all in the
    reference's state_dict layout.  ``rm / rv / nbt``: running statistics [T * f_out] and the counters, updated in place."""
    if snorm is not None:
        snorm = snorm.reshape(-1).contiguous()
    if plan.n_channels:
        eig = graph._normalised_eig(graph.ndata["eig"] if eig is None else eig)
        for ch in plan.channels:
            if ch[1] >= eig.shape[-1]:
                raise IndexError(f"aggregator needs eig column {ch[1]} but eig has {eig.shape[-1]} columns")
    else:
        eig = None
    cfg = (int(type_net), int(n_towers), int(f_in), int(f_out), bool(residual), float(momentum), float(eps), float(slope))
    if training and dropout is not None:      # (towers: ``(p, key tensor, offset)`` as ops.dropout takes them; evaluation applies none)
        cfg = cfg + ((float(dropout[0]), dropout[1], int(dropout[2])),)
    if not training:
        return _block_layer_eval(graph, plan, float(avg_log), eig, cfg, h, snorm, rm, rv, params)
    if DIRECT_PARAM_GRADS and h.requires_grad:
        return _BlockLayer.apply(graph, plan, float(avg_log), eig, cfg, h, snorm, rm, rv, nbt, tuple(params))
    return _BlockLayer.apply(graph, plan, float(avg_log), eig, cfg, h, snorm, rm, rv, nbt, None, *params)


# ---- the posttrans product inside the sweep (dgn_fused.hip) ---------------------------------------------------------------------

# True: when no gradient is needed (inference / validation passes) the towers layer runs sweep + posttrans + scale-combine as ONE
# kernel (layer_fwd_fused) and the [N, A*F] aggregate rows never reach memory: 0.346 instead of 0.383 ms on ZINC-12k.  Training keeps
# the separate kernels: the backward needs the aggregate rows for the posttrans weight gradient, and the recompute twin of this kernel
# measured slower than reading them back.
FUSED_FORWARD = os.environ.get("DGN_FUSED_FORWARD", "1") != "0"
# True: the towers layer's forward sweep records the slots of each row's first maximum / minimum and the dx signs (one byte per row
# and feature) and the backward sweep works from that table instead of gathering the source rows again (bit-identical gradients)
AGG_AUX = os.environ.get("DGN_AGG_AUX", "1") != "0"


def fused_sweep_posttrans_supported(graph: DGNGraph, plan: AggPlan, n_towers: int, F: int, n_scalers: int, f_out: int) -> bool:
    lib = _lib.load()
    if len(plan.launches) != 1:
        return False
    spec = _spec_structs(plan, n_towers, 1.0, 0)[0]
    return bool(lib.dgn_layer_fused_supported(C.byref(graph.c_graph), C.byref(spec), F, n_scalers, f_out))


def fused_sweep_posttrans_forward(graph: DGNGraph, plan: AggPlan, n_towers: int, avg_log: float, w_edge, x_pair, x_in, weight, scale, bias,
                                  row_scale) -> torch.Tensor:
    """``scale_combine(bmm(sweep(x_pair, x_in), weight^T), scale, bias, row_scale)`` as ONE kernel (dgn_layer_fused_forward): the
    tower-major aggregate rows never reach memory.  ``x_pair [N, 2F]`` = P | Q, ``x_in [N, F]``, ``weight [T, S*fo, K]``; returns
    ``y [N, T*fo]``.  Forward only (inference, and the recompute half of the training path)."""
    lib = _lib.load()
    N, F = x_in.shape
    T = n_towers
    S = 1 if scale is None else scale.shape[1]
    fo = weight.shape[1] // S
    weight = weight.contiguous()
    spec = _spec_structs(plan, T, avg_log, 0)[0]
    msg = _msg_struct(F, x_pair[:, :F], x_pair[:, F:], None, x_in)
    y = torch.empty((N, T * fo), dtype=torch.float32, device=x_in.device)
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).contiguous()
    stream = _lib.stream_ptr(x_in.device)
    rc = lib.dgn_layer_fused_forward(C.byref(graph.c_graph), C.byref(spec), C.byref(msg), _ptr(w_edge), w_edge.stride(0) if w_edge is not None else 0,
                                     graph.log_deg.data_ptr(), weight.data_ptr(), weight.stride(1), weight.stride(0), S, fo,
                                     _ptr(scale.contiguous() if scale is not None else None), _ptr(bias), _ptr(row_scale), y.data_ptr(), y.stride(0), stream)
    _lib.check(rc, "dgn_layer_fused_forward")
    return y


class _AssembleOperands(torch.autograd.Function):
    """The towers layer's fused operand buffer from its ~30 per-tower parameters in ONE launch (dgn_assemble_params), instead of
    cat + index_select + zeros + index_put (and, backward, their four nodes plus a cat-backward of 30 slices): at batch 128 the
    parameter plumbing was a third of the step's host time.  ``maps``: (ptr_table [P] int64 on the device + the host list it mirrors,
    map_param, map_off [total] int32, inv [n_flat] int64, sizes, shapes) built once per layer."""

    @staticmethod
    def forward(ctx, maps, *params):
        lib = _lib.load()
        ptrs = [p.data_ptr() for p in params]
        if ptrs != maps["ptr_host"]:                      # (parameters are updated in place: this changes on .to() / load only)
            maps["ptr_table"].copy_(torch.tensor(ptrs, dtype=torch.int64), non_blocking=False)
            maps["ptr_host"] = ptrs
        dev = params[0].device
        out = torch.empty(maps["total"], dtype=torch.float32, device=dev)
        stream = _lib.stream_ptr(dev)
        _lib.check(lib.dgn_assemble_params(maps["total"], maps["ptr_table"].data_ptr(), maps["map_param"].data_ptr(), maps["map_off"].data_ptr(),
                                           out.data_ptr(), stream), "dgn_assemble_params")
        ctx.maps = maps
        # the operands leave as views of the one buffer (no split node: its backward would concatenate the operand gradients -- a launch
        # and a copy -- where the whole-layer backward already writes them back to back, see backward)
        return tuple(part.view(shp) for part, shp in zip(out.split(maps["op_sizes"]), maps["op_shapes"]))

    @staticmethod
    def backward(ctx, *gs):
        maps = ctx.maps
        sizes, total = maps["op_sizes"], maps["total"]
        flat = None
        if all(g is not None and g.is_contiguous() and g.dtype == torch.float32 for g in gs):
            # _TowersLayer.backward carves the operand gradients out of one buffer in this very order: use it as the flat gradient
            st, at = gs[0].untyped_storage(), gs[0].data_ptr()
            for g, n in zip(gs, sizes):
                if g.data_ptr() != at or g.untyped_storage().data_ptr() != st.data_ptr():
                    st = None
                    break
                at += 4 * n
            if st is not None and (gs[0].storage_offset() + total) * 4 <= st.nbytes():
                flat = torch.empty(0, dtype=torch.float32, device=gs[0].device).set_(st, gs[0].storage_offset(), (total,))
        if flat is None:
            dev = next(g.device for g in gs if g is not None)
            flat = torch.cat([(g.reshape(-1) if g is not None else torch.zeros(n, dtype=torch.float32, device=dev)) for g, n in zip(gs, sizes)])
        g_flat = flat.index_select(0, maps["inv"])
        return (None,) + tuple(part.view(shp) for part, shp in zip(g_flat.split(maps["sizes"]), maps["shapes"]))


def assemble_operands(maps, params):
    """the operand tensors (views of one buffer, in the order of maps["op_sizes"] / maps["op_shapes"])"""
    return _AssembleOperands.apply(maps, *params)
