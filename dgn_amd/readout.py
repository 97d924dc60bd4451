"""Graph-level readouts and the virtual node on the aggregation sweep.

The reference's nets end with ``dgl.sum_nodes / mean_nodes / max_nodes`` (or the two directional readouts) over the
batched graph (realworld_benchmark/nets/molecules_graph_regression/dgn_net.py:71-86, HIV :76-83, PCBA :88-95,
superpixels :64-71), and ``VirtualNode`` pools the node features per graph (nets/dgn_layer.py:12-49).  A readout is
the same segmented reduction as the layer's reduce step, keyed by graph instead of by destination node: a
bipartite CSR whose rows are the graphs of the batch and whose sources are its nodes (``DgnGraph.n_src``).  So
these functions run ``dgn_agg_forward`` / ``dgn_agg_backward`` on that CSR -- the same kernels, hub slicing for
graphs with more than 2048 nodes included -- and the nets need no DGL call at all.

    sum_nodes(g, h), mean_nodes(g, h), max_nodes(g, h)      h: [N, F] tensor or an ndata key
    readout(g, h, mode, eig=None)                           mode: sum | max | mean | directional | directional_abs
    VirtualNode(dim, dropout, batch_norm, bias, residual, vn_type)      reference ctor / forward / state_dict

``g`` is anything with ``batch_num_nodes`` (DGL 0.4 attribute, later DGL method; list or tensor): the graphs
occupy consecutive node ranges.  An empty graph's row is zeros.
"""
from __future__ import annotations

from typing import Union

import torch
import torch.nn as nn

from .graph import DGNGraph
from .layers import FCLayer
from .ops import directional_aggregate
from .spec import make_plan

_PLANS = {k: make_plan([k], ["identity"]) for k in ("sum", "mean", "max")}
_WSUM_MEAN_PLAN = make_plan(["dir1-0.1", "mean"], ["identity"])      # weighted sum with SUPPLIED weights | mean


def _sizes(g) -> torch.Tensor:
    b = g.batch_num_nodes
    b = b() if callable(b) else b
    return torch.as_tensor(b, dtype=torch.long)


def readout_graph(g, device) -> DGNGraph:
    """The batch's graph->nodes CSR (cached on ``g``): row i = graph i, slots = its nodes in order."""
    cached = getattr(g, "_dgn_readout", None)
    if cached is not None and cached.device == torch.device(device):
        return cached
    sizes = _sizes(g).to(device)
    indptr = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(sizes, 0)
    n = int(indptr[-1].item())
    rg = DGNGraph.from_csr(indptr, torch.arange(n, dtype=torch.int32, device=device), num_src=n)
    rg.sizes = sizes
    try:
        g._dgn_readout = rg
    except Exception:
        pass
    return rg


def _feat(g, h: Union[str, torch.Tensor]) -> torch.Tensor:
    return g.ndata[h] if isinstance(h, str) else h


def _reduce(g, h, how: str) -> torch.Tensor:
    h = _feat(g, h).contiguous()
    return directional_aggregate(readout_graph(g, h.device), _PLANS[how], 1.0, x_src=h)


def sum_nodes(g, h) -> torch.Tensor:
    """dgl.sum_nodes(g, 'h') -> [n_graphs, F]"""
    return _reduce(g, h, "sum")


def mean_nodes(g, h) -> torch.Tensor:
    """dgl.mean_nodes(g, 'h') -> [n_graphs, F]"""
    return _reduce(g, h, "mean")


def max_nodes(g, h) -> torch.Tensor:
    """dgl.max_nodes(g, 'h') -> [n_graphs, F]"""
    return _reduce(g, h, "max")


def readout(g, h, mode: str, eig: torch.Tensor = None) -> torch.Tensor:
    """The readout branch of the nets (molecules_graph_regression/dgn_net.py:71-86); unknown modes fall back to the
    mean like the reference.  The directional modes weigh every node by ``eig[:, 1] / |eig[:, 1]|`` resp.
    ``|eig[:, 1]| / |eig[:, 1]|`` exactly as written there (the "sum over dim 1" of a one-column slice is the
    element itself; a zero entry gives NaN there and here): one sweep delivers both halves of the result."""
    h = _feat(g, h).contiguous()
    if mode in ("sum", "max"):
        return _reduce(g, h, mode)
    if mode in ("directional", "directional_abs"):
        eig = g.ndata["eig"] if eig is None else eig
        e1 = eig[:, 1:2].to(h.device)
        num = torch.abs(e1) if mode == "directional_abs" else e1
        w = (num / torch.sum(torch.abs(e1), dim=1, keepdim=True)).reshape(1, -1).float().contiguous()   # [1, N] per-slot weights
        rg = readout_graph(g, h.device)
        both = directional_aggregate(rg, _WSUM_MEAN_PLAN, 1.0, x_src=h, weights=w)                      # [G, 2F]: sum_n w_n h_n | mean
        F_ = h.shape[1]
        d = both[:, :F_] / rg.sizes.clamp(min=1).unsqueeze(1).to(h.dtype)                                # mean_nodes(g, 'dir')
        return torch.cat([torch.abs(d) if mode == "directional" else d, both[:, F_:]], dim=1)
    return _reduce(g, h, "mean")


class VirtualNode(nn.Module):
    """nets/dgn_layer.py:12-49 with the pooling on the sweep kernels; same constructor, forward and state_dict."""

    def __init__(self, dim, dropout, batch_norm=False, bias=True, residual=True, vn_type="mean"):
        super().__init__()
        self.vn_type = vn_type.lower()
        self.fc_layer = FCLayer(in_size=dim, out_size=dim, activation="relu", dropout=dropout, b_norm=batch_norm, bias=bias)
        self.residual = residual

    def forward(self, g, h, vn_h):
        try:
            g.ndata["h"] = h
        except Exception:
            pass
        if self.vn_type == "mean":
            pool = mean_nodes(g, h)
        elif self.vn_type == "sum":
            pool = sum_nodes(g, h)
        elif self.vn_type == "logsum":
            rg = readout_graph(g, h.device)
            pool = mean_nodes(g, h) * torch.log(rg.sizes.to(h.dtype)).unsqueeze(-1)
        else:
            raise ValueError(f'Undefined input "{self.vn_type}". Accepted values are "sum", "mean", "logsum"')
        vn_h_temp = self.fc_layer(vn_h + pool)
        vn_h = vn_h + vn_h_temp if self.residual else vn_h_temp
        rg = readout_graph(g, h.device)
        h = h + torch.repeat_interleave(vn_h, rg.sizes, dim=0)       # every node receives its graph's virtual node
        return vn_h, h
