"""CPU oracle for the DGN directional-aggregation hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  ``dgn_amd``
never imports it and has no CPU fallback.

It restates, in plain CPU torch (fp32 by default, fp64 on request), what the
reference computes between ``g.apply_edges`` and the scaler concat, and the
layer bodies around it.  Every function cites the reference file:line it
follows (paths relative to ``/root/reference``).  The structure is the
reference's own: per-edge message materialisation, DGL-style in-degree
bucketing, a ``[n, D, F]`` mailbox per bucket, one short chain of torch ops
per aggregator, concat, scalers, merge.  That makes it (a) the parity oracle
and (b) an honest "reference CPU path" to time on the host cores.

Parity status: PINNED for everything computed by in-tree reference code -- the
restatement is checked against golden vectors produced by importing the
reference's own functions/modules (``tests/golden/make_golden.py``, fixtures
``tests/golden/*.npz``, test ``tests/test_oracle_vs_golden.py``).
PARITY UNPINNED for two behaviours that live inside third-party DGL 0.4.2
(``realworld_benchmark/environment_gpu.yml:15``; source absent from the
reference tree, and the reference holds no tests): rows of zero-in-degree
nodes are defined as zeros, and mailbox order within a destination is
ascending edge id.  Both choices are encoded here and recorded in the
fixtures.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# realworld_benchmark/nets/aggregators.py:5
EPS = 1e-8

# ----------------------------------------------------------------------------
# mailbox-level aggregators: m [n, D, F], eig_s/eig_d [n, D, K], x [n, F]
# ----------------------------------------------------------------------------


def _delta(eig_s, eig_d, k):
    # eig_s[:, :, k] - eig_d[:, :, k]  (aggregators.py:36, :49, :56, :63)
    return eig_s[:, :, k] - eig_d[:, :, k]


def agg_mean(m, eig_s, eig_d, x):
    """aggregators.py:8-9"""
    return m.mean(dim=1)


def agg_sum(m, eig_s, eig_d, x):
    """aggregators.py:31-32"""
    return m.sum(dim=1)


def agg_max(m, eig_s, eig_d, x):
    """aggregators.py:12-13 (values only; torch routes the gradient to the index max returns)"""
    return m.max(dim=1)[0]


def agg_min(m, eig_s, eig_d, x):
    """aggregators.py:16-17"""
    return m.min(dim=1)[0]


def agg_var(m, eig_s, eig_d, x):
    """aggregators.py:24-28: relu(E[m^2] - E[m]^2) over the mailbox dimension."""
    second = (m * m).mean(dim=-2)
    first = m.mean(dim=-2)
    return torch.relu(second - first * first)


def agg_std(m, eig_s, eig_d, x):
    """aggregators.py:20-21: sqrt(var + EPS)."""
    return torch.sqrt(agg_var(m, eig_s, eig_d, x) + EPS)


def _abs_normalised(eig_s, eig_d, k):
    """delta / (sum_j |delta_j| + EPS), shape [n, D, 1]  (aggregators.py:49-50, :56-57)."""
    d = _delta(eig_s, eig_d, k)
    return (d / (d.abs().sum(dim=1, keepdim=True) + EPS)).unsqueeze(-1)


def agg_dir_av(m, eig_s, eig_d, x, k):
    """aggregators.py:35-39: sum_j |delta_j| / (sum|delta| + EPS) * m_j."""
    d = _delta(eig_s, eig_d, k).abs()
    w = (d / (d.sum(dim=1, keepdim=True) + EPS)).unsqueeze(-1)
    return (m * w).sum(dim=1)


def agg_dir_softmax(m, eig_s, eig_d, x, k, alpha):
    """aggregators.py:42-45: softmax over the mailbox of alpha*|delta|, weighted sum."""
    s = torch.softmax(alpha * _delta(eig_s, eig_d, k).abs().unsqueeze(-1), dim=1)
    return (m * s).sum(dim=1)


def agg_dir_dx_no_abs(m, eig_s, eig_d, x, k):
    """aggregators.py:55-59: sum_j w_j m_j - (sum_j w_j) x."""
    w = _abs_normalised(eig_s, eig_d, k)
    return (m * w).sum(dim=1) - w.sum(dim=1) * x


def agg_dir_dx(m, eig_s, eig_d, x, k):
    """aggregators.py:48-52: |sum_j w_j m_j - (sum_j w_j) x|."""
    return agg_dir_dx_no_abs(m, eig_s, eig_d, x, k).abs()


def agg_dir_dx_balanced(m, eig_s, eig_d, x, k):
    """aggregators.py:62-71: forward and backward fields normalised separately, averaged."""
    d = _delta(eig_s, eig_d, k)
    fwd = torch.relu(d)
    bwd = torch.relu(-d)
    fwd = (fwd / (fwd.abs().sum(dim=1, keepdim=True) + EPS)).unsqueeze(-1)
    bwd = (bwd / (bwd.abs().sum(dim=1, keepdim=True) + EPS)).unsqueeze(-1)
    w = (fwd + bwd) / 2
    return ((m * w).sum(dim=1) - w.sum(dim=1) * x).abs()


_PLAIN = {"mean": agg_mean, "sum": agg_sum, "max": agg_max, "min": agg_min,
          "std": agg_std, "var": agg_var}
_DIR_RE = re.compile(r"^dir([1-3])-(av|smooth|dx|dx-no-abs|dx-balanced|0\.1|neg-0\.1)$")


def get_aggregator(name: str):
    """Name -> mailbox function; the 24 names of aggregators.py:74-93, plus the
    ``dirK-smooth`` spelling that models/dgl/aggregators.py:76-78 and the README use
    for ``dirK-av``.  Unknown names raise KeyError like the reference's dict lookup
    (dgn_layer.py:335)."""
    if name in _PLAIN:
        return _PLAIN[name]
    mt = _DIR_RE.match(name)
    if mt is None:
        raise KeyError(name)
    k, kind = int(mt.group(1)), mt.group(2)
    if kind in ("av", "smooth"):
        return lambda m, es, ed, x: agg_dir_av(m, es, ed, x, k)
    if kind == "dx":
        return lambda m, es, ed, x: agg_dir_dx(m, es, ed, x, k)
    if kind == "dx-no-abs":
        return lambda m, es, ed, x: agg_dir_dx_no_abs(m, es, ed, x, k)
    if kind == "dx-balanced":
        return lambda m, es, ed, x: agg_dir_dx_balanced(m, es, ed, x, k)
    alpha = 0.1 if kind == "0.1" else -0.1
    return lambda m, es, ed, x: agg_dir_softmax(m, es, ed, x, k, alpha)


AGGREGATOR_NAMES = (["mean", "sum", "max", "min", "std", "var"]
                    + [f"dir{k}-{s}" for s in ("av", "0.1", "neg-0.1", "dx", "dx-no-abs", "dx-balanced")
                       for k in (1, 2, 3)])


# ----------------------------------------------------------------------------
# scalers  (realworld_benchmark/nets/scalers.py:7-18)
# ----------------------------------------------------------------------------


def scale(name: str, h: torch.Tensor, D: int, avg_log) -> torch.Tensor:
    """D is the python-int bucket degree; ``np.log(D + 1)`` is float64 and the
    division by the 0-dim fp32 tensor ``avg_d['log']`` yields an fp32 scalar
    tensor (probed in this container), which then multiplies ``h``."""
    if name == "identity":
        return h
    if not torch.is_tensor(avg_log):
        avg_log = torch.tensor(avg_log, dtype=h.dtype)
    if name == "amplification":
        return h * (np.log(D + 1) / avg_log)
    if name == "attenuation":
        return h * (avg_log / np.log(D + 1))
    raise KeyError(name)


SCALER_NAMES = ["identity", "amplification", "attenuation"]


# ----------------------------------------------------------------------------
# reduce over a whole graph: DGL-0.4-style degree bucketing
# ----------------------------------------------------------------------------


def reduce_bucket(aggs: Sequence[str], scalers: Sequence[str], m, eig_s, eig_d, x, avg_log):
    """dgn_layer.py:161-173 (``reduce_func``): aggregator concat, then the scaler
    concat ONLY IF more than one scaler is listed (:170)."""
    D = m.shape[-2]
    h = torch.cat([get_aggregator(a)(m, eig_s, eig_d, x) for a in aggs], dim=1)
    if len(scalers) > 1:
        h = torch.cat([scale(s, h, D, avg_log) for s in scalers], dim=1)
    return h


def aggregate_graph(src: torch.Tensor, dst: torch.Tensor, num_nodes: int, msg: torch.Tensor,
                    eig: torch.Tensor, x_in: torch.Tensor, aggs: Sequence[str],
                    scalers: Sequence[str], avg_log) -> torch.Tensor:
    """``apply_edges`` + ``update_all`` (dgn_layer.py:183-186) for per-edge messages
    ``msg [E, F]`` (edge-id order), node field ``eig [N, K]`` and ``x_in [N, F]``.

    Returns ``[N, A*S*F]`` (S = 1 when a single scaler is listed).  Destinations are
    grouped by in-degree; the mailbox of a destination holds its messages in
    ascending edge-id order; zero-in-degree rows are zeros (DGL-internal, unpinned).
    """
    E = src.numel()
    F_ = msg.shape[1]
    n_s = len(scalers) if len(scalers) > 1 else 1
    width = len(aggs) * n_s * F_
    out = msg.new_zeros((num_nodes, width))
    if E == 0:
        return out
    eig = eig.to(msg.dtype)
    eig_s = eig[src]                      # edges.src['eig']  (dgn_layer.py:155)
    eig_d = eig[dst]                      # edges.dst['eig']
    order = torch.sort(dst, stable=True)[1]   # ascending edge id inside each destination
    deg = torch.bincount(dst, minlength=num_nodes)
    ptr = torch.zeros(num_nodes + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(deg, 0)
    for D in torch.unique(deg).tolist():
        if D == 0:
            continue
        nodes = torch.nonzero(deg == D).flatten()
        slots = (ptr[nodes].unsqueeze(1) + torch.arange(D).unsqueeze(0)).reshape(-1)
        eids = order[slots]
        m = msg[eids].reshape(len(nodes), D, F_)
        es = eig_s[eids].reshape(len(nodes), D, -1)
        ed = eig_d[eids].reshape(len(nodes), D, -1)
        h = reduce_bucket(aggs, scalers, m, es, ed, x_in[nodes], avg_log)
        out = out.index_copy(0, nodes, h)
    return out


# ----------------------------------------------------------------------------
# layer bodies (functional, driven by a reference-layout state_dict)
# ----------------------------------------------------------------------------


def _mlp(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, last_activation=None) -> torch.Tensor:
    """nets/layers.py:146-149 + :101-112 for MLP(mid_activation='relu', last_activation='none')."""
    n = 0
    while f"{prefix}.fully_connected.{n}.linear.weight" in sd:
        n += 1
    for i in range(n):
        x = F.linear(x, sd[f"{prefix}.fully_connected.{i}.linear.weight"],
                     sd[f"{prefix}.fully_connected.{i}.linear.bias"])
        if i < n - 1:
            x = torch.relu(x)
    return x


def _bn(sd, prefix, h, training, stats_out):
    rm = sd[f"{prefix}.running_mean"].detach().clone()
    rv = sd[f"{prefix}.running_var"].detach().clone()
    h = F.batch_norm(h, rm, rv, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"],
                     training=training, momentum=0.1, eps=1e-5)
    stats_out[f"{prefix}.running_mean"] = rm
    stats_out[f"{prefix}.running_var"] = rv
    return h


def _pretrans_messages(sd, prefix, h, e, src, dst, edge_features):
    """dgn_layer.py:75-80 / :226-231: MLP([h_src || h_dst (|| ef)])."""
    z = [h[src], h[dst]]
    if edge_features:
        z.append(e)
    return _mlp(sd, prefix + "pretrans", torch.cat(z, dim=1))


def layer_forward(type_net: str, sd: Dict[str, torch.Tensor], cfg: dict, src, dst, num_nodes,
                  eig, h, e, snorm_n, training: bool = True, dropout=None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Forward of ``DGNLayer(...).model`` for a state_dict in the reference's layout.

    cfg keys: aggregators (str), scalers (str), avg_log, graph_norm, batch_norm, residual,
    edge_features, towers, divide_input; dropout is taken as 0 (parity runs) unless ``dropout = (p, keep)`` is
    given: the simple / complex layer's last op ``F.dropout(h, p, training)`` (dgn_layer.py:130, :201), every tower's last op (:275)
    with the Bernoulli draw replaced by the given keep mask [N, out_dim] -- ``h * keep / (1 - p)``, F.dropout's own
    arithmetic (torch: ``input * mask * (1 / (1 - p))``).
    Follows dgn_layer.py:178-202 (simple), :103-132 (complex), :254-276 + :309-325 (towers).
    Returns (output, updated BN running stats).
    """
    aggs = cfg["aggregators"].split()
    scalers = cfg["scalers"].split()
    avg_log = cfg["avg_log"]
    stats: Dict[str, torch.Tensor] = {}
    in_dim = h.shape[1]

    def tail(prefix, y, relu):
        if cfg["graph_norm"]:
            y = y * snorm_n
        if cfg["batch_norm"]:
            y = _bn(sd, prefix + "batchnorm_h", y, training, stats)
        if relu:
            y = torch.relu(y)
        return y

    def drop(y):
        if dropout is None or not training:
            return y
        p_drop, keep = dropout
        return y * keep.to(y.dtype) * (1.0 / (1.0 - p_drop))

    if type_net == "simple":
        agg = aggregate_graph(src, dst, num_nodes, h[src], eig, h, aggs, scalers, avg_log)
        y = tail("", _mlp(sd, "posttrans", agg), relu=True)
        if cfg["residual"] and y.shape[1] == in_dim:
            y = h + y
        return drop(y), stats

    if type_net == "complex":
        msg = _pretrans_messages(sd, "", h, e, src, dst, cfg.get("edge_features"))
        agg = aggregate_graph(src, dst, num_nodes, msg, eig, h, aggs, scalers, avg_log)
        y = tail("", _mlp(sd, "posttrans", torch.cat([h, agg], dim=1)), relu=True)
        if cfg["residual"] and y.shape[1] == in_dim:
            y = h + y
        return drop(y), stats

    if type_net == "towers":
        towers = cfg.get("towers", 5)
        divide = cfg.get("divide_input", True)
        ft = in_dim // towers if divide else in_dim
        outs = []
        for t in range(towers):
            ht = h[:, t * ft:(t + 1) * ft] if divide else h
            p = f"towers.{t}."
            msg = _pretrans_messages(sd, p, ht, e, src, dst, cfg.get("edge_features"))
            agg = aggregate_graph(src, dst, num_nodes, msg, eig, ht, aggs, scalers, avg_log)
            outs.append(tail(p, _mlp(sd, p + "posttrans", torch.cat([ht, agg], dim=1)), relu=False))
        y = drop(torch.cat(outs, dim=1))      # DGNTower's last op on every tower (dgn_layer.py:275): keep is [N, towers * out_tower]
        if towers > 1:
            # FCLayer(out_dim, out_dim, activation='LeakyReLU')  (dgn_layer.py:307, :318-319)
            y = F.leaky_relu(F.linear(y, sd["mixing_network.linear.weight"], sd["mixing_network.linear.bias"]))
        if cfg["residual"] and y.shape[1] == in_dim:
            y = h + y
        return y, stats

    raise ValueError(type_net)


# ----------------------------------------------------------------------------
# convenience: one aggregation fwd+bwd pass, used as the timed CPU baseline
# ----------------------------------------------------------------------------


def aggregate_fwd_bwd(src, dst, num_nodes, h, eig, aggs, scalers, avg_log):
    """Simple-layer message path (message = h[src]) forward + backward of the
    aggregation only; returns (out, grad_h)."""
    h = h.detach().clone().requires_grad_(True)
    out = aggregate_graph(src, dst, num_nodes, h[src], eig, h, aggs, scalers, avg_log)
    out.backward(torch.ones_like(out))
    return out.detach(), h.grad
