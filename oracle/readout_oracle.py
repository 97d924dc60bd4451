"""CPU oracle for the graph-level readouts and the virtual node.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as dgn_oracle.py: only tests/ may import it).

Restates, in plain CPU torch:
* ``dgl.sum_nodes / mean_nodes / max_nodes`` over the consecutive node blocks of a batched graph,
* the readout branch of the nets, ``realworld_benchmark/nets/molecules_graph_regression/dgn_net.py:71-86``
  (``sum``, ``max``, ``mean``, ``directional``, ``directional_abs``; the other nets use the first three:
  ``HIV_graph_classification/dgn_net.py:76-83``, ``PCBA_graph_classification/dgn_net.py:88-95``,
  ``superpixels_graph_classification/dgn_net.py:64-71``),
* ``VirtualNode.forward``, ``realworld_benchmark/nets/dgn_layer.py:21-49``.

Parity status: the net-level formulas and VirtualNode are PINNED by tests/golden/g8_readouts.npz (produced by
running the reference's unmodified DGNNet / VirtualNode, tests/golden/make_golden.py::g8_readouts).
PARITY UNPINNED for the three DGL functions themselves (DGL 0.4.2 is absent): they are taken to reduce each
graph's consecutive block of nodes, as DGL documents; an empty graph's row is defined as zeros.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def segment_reduce(h: torch.Tensor, sizes: Sequence[int], how: str) -> torch.Tensor:
    """dgl.{sum,mean,max}_nodes(g, 'h'): one row per graph of the batch."""
    outs, off = [], 0
    for n in sizes:
        blk = h[off:off + n]
        if n == 0:
            outs.append(h.new_zeros(h.shape[1]))
        elif how == "sum":
            outs.append(blk.sum(0))
        elif how == "mean":
            outs.append(blk.mean(0))
        elif how == "max":
            outs.append(blk.max(0)[0])
        else:
            raise ValueError(how)
        off += n
    return torch.stack(outs) if outs else h.new_zeros(0, h.shape[1])


def readout(h: torch.Tensor, sizes: Sequence[int], mode: str, eig: torch.Tensor = None) -> torch.Tensor:
    """molecules_graph_regression/dgn_net.py:71-86."""
    if mode == "sum":
        return segment_reduce(h, sizes, "sum")                                             # :71-72
    if mode == "max":
        return segment_reduce(h, sizes, "max")                                             # :73-74
    if mode == "directional_abs":                                                          # :77-80
        e1 = eig[:, 1:2]
        d = h * torch.abs(e1) / torch.sum(torch.abs(e1), dim=1, keepdim=True)
        return torch.cat([segment_reduce(d, sizes, "mean"), segment_reduce(h, sizes, "mean")], dim=1)
    if mode == "directional":                                                              # :81-84
        e1 = eig[:, 1:2]
        d = h * e1 / torch.sum(torch.abs(e1), dim=1, keepdim=True)
        return torch.cat([torch.abs(segment_reduce(d, sizes, "mean")), segment_reduce(h, sizes, "mean")], dim=1)
    return segment_reduce(h, sizes, "mean")                                                # :75-76, :85-86 (default)


def virtual_node_forward(sd: Dict[str, torch.Tensor], h: torch.Tensor, vn_h: torch.Tensor, sizes: Sequence[int],
                         vn_type: str, residual: bool, training: bool = True, momentum: float = 0.1, eps: float = 1e-5):
    """VirtualNode.forward, dgn_layer.py:21-49; ``sd`` = the module's state_dict (``fc_layer.linear.*`` and, with
    batch norm, ``fc_layer.b_norm.*``).  Returns (vn_h', h', new running stats or None)."""
    vn_type = vn_type.lower()
    if vn_type == "mean":                                                                  # :26-27
        pool = segment_reduce(h, sizes, "mean")
    elif vn_type == "sum":                                                                 # :28-29
        pool = segment_reduce(h, sizes, "sum")
    elif vn_type == "logsum":                                                              # :30-33
        pool = segment_reduce(h, sizes, "mean")
        lognum = torch.log(torch.tensor(list(sizes), dtype=h.dtype))
        pool = pool * lognum.unsqueeze(-1)
    else:
        raise ValueError(vn_type)
    # FCLayer(dim, dim, activation='relu', b_norm=batch_norm): linear -> relu -> (dropout) -> batch norm, layers.py:101-112
    t = F.relu(F.linear(vn_h + pool, sd["fc_layer.linear.weight"], sd.get("fc_layer.linear.bias")))
    stats = None
    if "fc_layer.b_norm.weight" in sd:
        rm, rv = sd["fc_layer.b_norm.running_mean"].clone(), sd["fc_layer.b_norm.running_var"].clone()
        t = F.batch_norm(t, rm, rv, sd["fc_layer.b_norm.weight"], sd["fc_layer.b_norm.bias"], training, momentum, eps)
        stats = (rm, rv)
    vn_new = vn_h + t if residual else t                                                   # :39-43
    sizes_t = torch.tensor(list(sizes), dtype=torch.long)
    h_new = h + torch.repeat_interleave(vn_new, sizes_t, dim=0)                            # :45-48
    return vn_new, h_new, stats
