"""Host-side mirror of the reference's graph-regression net (the direct caller of the layer):
``realworld_benchmark/nets/molecules_graph_regression/dgn_net.py:8-95`` (DGNNet) and ``nets/mlp_readout_layer.py:13-32`` (MLPReadout).

Same constructor dictionary, ``forward(g, h, e, snorm_n, snorm_e)``, ``loss`` and ``state_dict`` keys (a reference checkpoint loads
as is); no DGL call: the layers are ``dgn_amd.DGNLayer``, the readouts ``dgn_amd.readout``, and with ``edge_feat`` the bond-type
embedding is handed to the layers as ``EdgeTypeFeatures(embedding_e.weight, bond_type)`` -- the K x F table inside the sweep instead
of the gathered ``[E, edge_dim]`` rows (dgn_net.py:75).  ``g`` is a ``DGNGraph`` (or anything ``as_dgn_graph`` accepts) that carries
``batch_num_nodes`` for the readout and ``ndata['eig']``.  Parity: fixture G10 (tests/golden/make_golden.py::g10_net), produced by
the unmodified reference net."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .dgn_layer import DGNLayer, EdgeTypeFeatures
from .readout import readout


class MLPReadout(nn.Module):
    """``L`` hidden Linear + ReLU layers (widths halving when ``decreasing_dim``) and an output Linear; keys ``FC_layers.{i}.*``."""

    def __init__(self, input_dim: int, output_dim: int, L: int = 2, decreasing_dim: bool = True):
        super().__init__()
        widths = [input_dim // 2 ** i if decreasing_dim else input_dim for i in range(L + 1)]
        self.FC_layers = nn.ModuleList([nn.Linear(widths[i], widths[i + 1], bias=True) for i in range(L)] +
                                       [nn.Linear(widths[L], output_dim, bias=True)])
        self.L = L

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for fc in self.FC_layers[:-1]:
            x = F.relu(fc(x))
        return self.FC_layers[-1](x)


class _SmallTableEmbedding(torch.autograd.Function):
    """``nn.Embedding`` lookup (nets/molecules_graph_regression/dgn_net.py:44, :66: 28 atom types) whose weight gradient is the product
    one_hot(idx)^T g instead of torch's sort-based ``embedding_dense_backward``: on a 3 000-node batch that backward is ~100 us of sort /
    scatter kernels (149 us eager) next to 0.1 ms layers; the product is two small kernels, and deterministic."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        onehot = torch.zeros(idx.numel(), ctx.rows, dtype=g.dtype, device=g.device)
        onehot.scatter_(1, idx.reshape(-1, 1), 1.0)
        return onehot.t().mm(g.reshape(idx.numel(), -1)), None


def small_table_embedding(emb: nn.Embedding, idx: torch.Tensor) -> torch.Tensor:
    """``emb(idx)`` for small tables of plain embeddings (no padding_idx / max_norm / sparse), else the module itself."""
    if (emb.num_embeddings <= 256 and idx.dim() == 1 and idx.is_cuda and emb.padding_idx is None and emb.max_norm is None and not emb.sparse
            and not emb.scale_grad_by_freq):
        return _SmallTableEmbedding.apply(emb.weight, idx)
    return emb(idx)


class DGNNet(nn.Module):
    def __init__(self, net_params: dict):
        super().__init__()
        p = net_params
        hidden, out_dim, n_layers = p["hidden_dim"], p["out_dim"], p["L"]
        self.type_net, self.pos_enc_dim, self.readout = p["type_net"], p["pos_enc_dim"], p["readout"]
        self.edge_feat, self.device = p["edge_feat"], p["device"]
        self.in_feat_dropout = nn.Dropout(p["in_feat_dropout"])
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden)
        if self.pos_enc_dim > 0:
            self.embedding_pos_enc = nn.Linear(self.pos_enc_dim, hidden)
        if self.edge_feat:
            self.embedding_e = nn.Embedding(p["num_bond_type"], p["edge_dim"])
        make = lambda o: DGNLayer(in_dim=hidden, out_dim=o, dropout=p["dropout"], graph_norm=p["graph_norm"], batch_norm=p["batch_norm"],
                                  residual=p["residual"], aggregators=p["aggregators"], scalers=p["scalers"], avg_d=p["avg_d"],
                                  type_net=self.type_net, edge_features=self.edge_feat, edge_dim=p["edge_dim"],
                                  pretrans_layers=p["pretrans_layers"], posttrans_layers=p["posttrans_layers"]).model
        self.layers = nn.ModuleList([make(hidden) for _ in range(n_layers - 1)] + [make(out_dim)])
        directional = self.readout in ("directional", "directional_abs")
        self.MLP_layer = MLPReadout(2 * out_dim if directional else out_dim, 1)

    def forward(self, g, h, e, snorm_n, snorm_e=None):
        h = self.in_feat_dropout(small_table_embedding(self.embedding_h, h))
        if self.pos_enc_dim > 0:
            h = h + self.embedding_pos_enc(g.ndata["pos_enc"].to(h.device))
        if self.edge_feat:
            e = EdgeTypeFeatures(self.embedding_e.weight, e)
        for conv in self.layers:
            h = conv(g, h, e, snorm_n)
        g.ndata["h"] = h
        mode = self.readout if self.readout in ("sum", "max", "mean", "directional", "directional_abs") else "mean"
        return self.MLP_layer(readout(g, h, mode))

    def loss(self, scores, targets):
        return F.l1_loss(scores, targets)
