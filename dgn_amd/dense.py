"""The reference's DENSE DGN API (models/pytorch/{aggregators,scalers,dgn_layer}.py) on the MI355X kernels.

    DGNLayer(in_features, out_features, aggregators: list[str], scalers: list[str], NN_eig, avg_d, eigs, towers=1,
             self_loop=False, pretrans_layers=1, posttrans_layers=1, divide_input=True, device='cpu')
    forward(input [B, N, F], adj [B, N, N], eigvec [B, N, K]) -> [B, N, out_features]

The reference builds all N^2 pair messages ``X[b,i,j] = pretrans([h_i || h_j])`` and reduces them with dense
[B,N,N] weight matrices.  Here only the entries that can contribute (adj != 0, plus the diagonal, which the
directional derivative, ``identity`` and ``self_loop`` read) become CSR slots; the scalar weight matrices are
formed exactly as the reference does (they are O(B N^2) scalars), and every F-wide reduction runs on the HIP
kernels: weighted segment sums through ``dgn_agg_forward`` with explicit per-slot weights, column-wise max/min
through a second, transposed CSR.  Same names, widths, quirks and ``state_dict`` keys as the reference; parity is
pinned by tests/golden/g6_dense.npz.  This is an API/parity path (O(nnz F) work), not a benchmarked one.

Reference quirks kept: ``std``/``mean_amplified``/``mean_attenuated`` always
add self loops; ``max``/``min`` reduce over column neighbours; ``softmax``/``softmin`` raise TypeError;
``momentN`` with self_loop re-adds the loop inside its mean; the degree scalers are always applied.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .graph import DGNGraph
from .layers import MLP, FCLayer
from .ops import directional_aggregate
from .spec import make_plan

EPS = 1e-5   # models/pytorch/aggregators.py:8

_WSUM_PLAN = make_plan(["dir1-0.1"], ["identity"])     # op DGN_AGG_DIR_WSUM, one channel: weights are supplied
_MAX_PLAN = make_plan(["max"], ["identity"])
_MIN_PLAN = make_plan(["min"], ["identity"])


class DenseBatch:
    """CSR slots of a dense adjacency: every (b, i, j) with adj != 0 plus every diagonal entry; rows are
    destinations i (the reference reduces over j, dim=2), sources are j."""

    def __init__(self, adj: torch.Tensor):
        if not adj.is_cuda:
            raise RuntimeError("dgn_amd.dense runs on the GPU only")
        B, N, _ = adj.shape
        dev = adj.device
        self.B, self.N, self.adj = B, N, adj
        eye = torch.eye(N, dtype=torch.bool, device=dev).unsqueeze(0)
        b, i, j = ((adj != 0) | eye).nonzero(as_tuple=True)          # lexicographic: rows ascending, j ascending
        self.b, self.i, self.j = b, i, j
        self.row, self.col = b * N + i, b * N + j
        deg = torch.bincount(self.row, minlength=B * N)
        indptr = torch.zeros(B * N + 1, dtype=torch.int64, device=dev)
        indptr[1:] = torch.cumsum(deg, 0)
        self.graph = DGNGraph.from_csr(indptr, self.col)
        self.diag_slots = torch.nonzero(i == j).flatten()            # exactly one per row, ascending row order
        self._col_graphs: Dict[bool, tuple] = {}

    def slots(self, dense: torch.Tensor) -> torch.Tensor:
        """[B, N, N] -> per-slot values."""
        return dense[self.b, self.i, self.j]

    def wsum(self, dense_w: torch.Tensor, msg: torch.Tensor) -> torch.Tensor:
        """sum_j W[b,i,j] * msg[slot(b,i,j)]  ->  [B*N, F]   (aggregate_sum, aggregators.py:80-88)"""
        w = self.slots(dense_w).float().contiguous().unsqueeze(0)
        return directional_aggregate(self.graph, _WSUM_PLAN, 1.0, m_edge=msg.contiguous(), weights=w)

    def col_extreme(self, adj_eff: torch.Tensor, msg: torch.Tensor, plan, fill: float) -> torch.Tensor:
        """max/min over dim -3: node j collects X[b, i, j] over the rows i with adj_eff[b, i, j] > 0
        (aggregators.py:34-55); columns with no such row get +-inf like the reference."""
        sel = torch.nonzero(self.slots(adj_eff) > 0).flatten()
        g = DGNGraph(self.row[sel], self.col[sel], self.B * self.N)
        out = directional_aggregate(g, plan, 1.0, m_edge=g.to_slot_order(msg.index_select(0, sel)))
        empty = (g.in_degree == 0).unsqueeze(1)
        return torch.where(empty, torch.full_like(out, fill), out)


def _loops(adj, self_loop):
    return adj + torch.eye(adj.shape[-1], dtype=adj.dtype, device=adj.device).unsqueeze(0) if self_loop else adj


def _gradient_adjacency(adj, feat):
    """eigen_agg.py:295-379, normalization='row-abs', add_diag=True."""
    G = adj * (feat.unsqueeze(-2) - feat.unsqueeze(-1) + EPS)
    nrm = G.abs()
    nrm = nrm * (nrm > EPS)
    G = G / (nrm.sum(-1, keepdim=True) + EPS)
    eye = torch.eye(adj.shape[-1], dtype=adj.dtype, device=adj.device).unsqueeze(0)
    return G - eye * G.sum(-1, keepdim=True)


def _mean(db: DenseBatch, msg, adj_eff):
    """sum_j adj_ij msg_ij / D_i -- sum first, divide after, like aggregators.py:28-30 (the rounding of
    (x*a)/a decides the sign of near-zero central moments, so the order of operations is kept)."""
    return db.wsum(adj_eff, msg) / adj_eff.sum(-1).reshape(-1, 1)


def _scale(name, x, adj, avg_d):
    """models/pytorch/scalers.py:7-38; x [B*N, W]."""
    if name == "identity":
        return x
    D = adj.sum(-1).reshape(-1, 1)
    if name == "amplification":
        return (torch.log(D + 1) / avg_d["log"]) * x
    if name == "attenuation":
        return (avg_d["log"] / torch.log(D + 1)) * x
    if name == "linear":
        return D * x / avg_d["lin"]
    if name == "inverse_linear":
        return avg_d["lin"] * x / D
    raise KeyError(name)


_DIR_RE = re.compile(r"^dir([1-5])-(dx|smooth|both)$")
_PLAIN = ("mean", "sum", "max", "min", "identity", "std", "var", "normalised_mean", "moment3", "moment4", "moment5",
          "mean_amplified", "mean_attenuated", "softmax", "softmin")
AGGREGATOR_NAMES = _PLAIN + ("dir0",) + tuple(f"dir{k}-{s}" for s in ("dx", "smooth", "both") for k in range(1, 6))
SCALER_NAMES = ("identity", "linear", "inverse_linear", "amplification", "attenuation")


def aggregator_width(name: str) -> int:
    """Number of F-wide blocks the aggregator emits (the reference probes this with a dummy call,
    models/pytorch/dgn_layer.py:27-28)."""
    if name in _PLAIN or name == "dir0":
        return 1
    m = _DIR_RE.match(name)
    if m is None:
        raise KeyError(name)
    return int(m.group(1)) * (2 if m.group(2) == "both" else 1)


def aggregate(name: str, db: DenseBatch, msg: torch.Tensor, eigvec=None, self_loop=False, avg_d=None) -> torch.Tensor:
    """AGGREGATORS[name] of models/pytorch/aggregators.py:231-271 on slot messages ``msg [nnz, F]`` -> [B*N, W*F]."""
    adj = db.adj
    if name == "mean":
        return _mean(db, msg, _loops(adj, self_loop))
    if name == "sum":
        return db.wsum(_loops(adj, self_loop), msg)
    if name == "max":
        return db.col_extreme(_loops(adj, self_loop), msg, _MAX_PLAN, -math.inf)
    if name == "min":
        return db.col_extreme(_loops(adj, self_loop), msg, _MIN_PLAN, math.inf)
    if name == "identity":
        return msg.index_select(0, db.diag_slots)
    if name in ("var", "std"):
        a = _loops(adj, True if name == "std" else self_loop)        # std: the reference's positional-argument slip
        mean = _mean(db, msg, a)
        var = torch.relu(_mean(db, msg * msg, a) - mean * mean)
        return torch.sqrt(var + EPS) if name == "std" else var
    if name == "normalised_mean":
        a = _loops(adj, self_loop)
        r = a.sum(-1).pow(-0.5)
        return db.wsum(r.unsqueeze(-1) * a * r.unsqueeze(-2), msg)
    if name in ("moment3", "moment4", "moment5"):
        n = int(name[-1])
        a = _loops(adj, self_loop)
        mean = _mean(db, msg, _loops(a, self_loop))                  # the inner mean adds the loop again
        xn = db.wsum(a, (msg - mean.index_select(0, db.row)).pow(n)) / a.sum(-1).reshape(-1, 1)
        return torch.sign(xn) * (xn.abs() + EPS).pow(1.0 / n)
    if name == "mean_amplified":
        return _scale("amplification", _mean(db, msg, _loops(adj, True)), adj, avg_d)
    if name == "mean_attenuated":
        return _scale("attenuation", _mean(db, msg, _loops(adj, True)), adj, avg_d)
    if name in ("softmax", "softmin"):
        raise TypeError(f"aggregator '{name}' is unusable in the reference (models/pytorch/aggregators.py:117)")
    if name == "dir0":
        idx, kind = [0], "smoothing"
    else:
        m = _DIR_RE.match(name)
        if m is None:
            raise KeyError(name)
        idx = list(range(1, int(m.group(1)) + 1))
        kind = {"dx": "derivative", "smooth": "smoothing", "both": "both"}[m.group(2)]
    eigvec = eigvec.to(adj.device)
    out = []
    for ii in idx:
        if ii != 0:
            v = eigvec[..., ii]
            G = _gradient_adjacency(adj, torch.acos(v / v.abs().max()))
        else:
            G = adj / (adj.abs().sum(-1, keepdim=True) + EPS)
        if kind in ("derivative", "both") and ii != 0:
            out.append(db.wsum(_loops(G, self_loop), msg))
        if kind in ("smoothing", "both") or ii == 0:
            out.append(db.wsum(_loops(G.abs(), self_loop), msg))
    return torch.cat(out, dim=-1)


class DGNTower(nn.Module):
    """models/pytorch/dgn_layer.py:9-57"""

    def __init__(self, in_features, out_features, aggregators, scalers, avg_d, self_loop, eigs, pretrans_layers,
                 posttrans_layers, device):
        super().__init__()
        self.device, self.in_features, self.out_features = device, in_features, out_features
        self.aggregators, self.scalers = list(aggregators), list(scalers)
        self.self_loop, self.eigs, self.avg_d = self_loop, eigs, avg_d
        for a in self.aggregators:
            if a in ("mean_amplified", "mean_attenuated", "softmax", "softmin"):
                # the reference's constructor probe calls every aggregator without avg_d and fails on these
                raise TypeError(f"aggregator '{a}' cannot be used inside DGNTower (models/pytorch/dgn_layer.py:27-28)")
        width = sum(aggregator_width(a) for a in self.aggregators)
        for s in self.scalers:
            if s not in SCALER_NAMES:
                raise KeyError(s)
        self.pretrans = MLP(in_size=2 * in_features, hidden_size=in_features, out_size=in_features,
                            layers=pretrans_layers, mid_activation="relu", last_activation="none")
        self.posttrans = MLP(in_size=(width * len(self.scalers) + 1) * in_features, hidden_size=out_features,
                             out_size=out_features, layers=posttrans_layers, mid_activation="relu", last_activation="none")

    def forward(self, input, adj, eigvec, batch: Optional[DenseBatch] = None):
        B, N, _ = adj.shape
        db = batch if batch is not None else DenseBatch(adj)
        h = input.reshape(B * N, -1)
        msg = self.pretrans(torch.cat([h.index_select(0, db.row), h.index_select(0, db.col)], dim=1))
        m = torch.cat([aggregate(a, db, msg, eigvec, self.self_loop, self.avg_d) for a in self.aggregators], dim=1)
        m = torch.cat([_scale(s, m, adj, self.avg_d) for s in self.scalers], dim=1)
        return self.posttrans(torch.cat([h, m], dim=1)).reshape(B, N, -1)

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"


class DGNLayer(nn.Module):
    """models/pytorch/dgn_layer.py:60-112"""

    def __init__(self, in_features, out_features, aggregators, scalers, NN_eig, avg_d, eigs, towers=1, self_loop=False,
                 pretrans_layers=1, posttrans_layers=1, divide_input=True, device="cpu"):
        super().__init__()
        assert ((not divide_input) or in_features % towers == 0), "if divide_input is set the number of towers has to divide in_features"
        assert (out_features % towers == 0), "the number of towers has to divide the out_features"
        for a in aggregators:
            if a not in AGGREGATOR_NAMES:
                raise KeyError(a)
        self.in_features, self.out_features = in_features, out_features
        self.divide_input = divide_input
        self.input_tower = in_features // towers if divide_input else in_features
        self.output_tower = out_features // towers
        self.towers = nn.ModuleList()
        for _ in range(towers):
            self.towers.append(DGNTower(in_features=self.input_tower, out_features=self.output_tower, aggregators=aggregators,
                                        scalers=scalers, avg_d=avg_d, self_loop=self_loop, eigs=eigs,
                                        pretrans_layers=pretrans_layers, posttrans_layers=posttrans_layers, device=device))
        self.mixing_network = FCLayer(out_features, out_features, activation="LeakyReLU")

    def forward(self, input, adj, eigvec=None):
        db = DenseBatch(adj)
        if self.divide_input:
            y = torch.cat([tower(input[:, :, n * self.input_tower:(n + 1) * self.input_tower], adj, eigvec, db)
                           for n, tower in enumerate(self.towers)], dim=2)
        else:
            y = torch.cat([tower(input, adj, eigvec, db) for tower in self.towers], dim=2)
        return self.mixing_network(y)

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"
