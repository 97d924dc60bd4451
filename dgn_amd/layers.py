"""FCLayer / MLP with the reference's constructor arguments, init and state_dict layout
(realworld_benchmark/nets/layers.py:21-154).  These are the dense pre/post-aggregation transforms.  The Linear inside goes
through ``ops.node_linear``: the library's own fp32-MFMA kernels (``dgn_linear_*`` up to 160 columns, ``dgn_gemm_*`` beyond) from a
few thousand rows on, ``torch.nn.functional.linear`` only for smaller batches.
"""
from __future__ import annotations

import torch
import torch.nn as nn

SUPPORTED_ACTIVATION_MAP = {"ReLU", "Sigmoid", "Tanh", "ELU", "SELU", "GLU", "LeakyReLU", "Softplus", "None"}


def get_activation(activation):
    """String (any case) or callable -> activation module, None for 'none' (layers.py:7-18)."""
    if activation and callable(activation):
        return activation
    hits = [x for x in SUPPORTED_ACTIVATION_MAP if str(activation).lower() == x.lower()]
    assert len(hits) == 1, "Unhandled activation function"
    if hits[0].lower() == "none":
        return None
    return getattr(torch.nn.modules.activation, hits[0])()


class FCLayer(nn.Module):
    """Linear -> activation -> dropout -> batch-norm (layers.py:101-112).

    Weight init is ``xavier_uniform_`` with gain ``1 / in_size`` and a zero bias
    (layers.py:94-99), reproduced so that a fresh layer under the same seed matches.
    """

    def __init__(self, in_size, out_size, activation="relu", dropout=0.0, b_norm=False, bias=True, init_fn=None,
                 device="cpu"):
        super().__init__()
        self.in_size, self.out_size, self.bias = in_size, out_size, bias
        self.linear = nn.Linear(in_size, out_size, bias=bias).to(device)
        self.dropout = nn.Dropout(p=dropout) if dropout else None
        self.b_norm = nn.BatchNorm1d(out_size).to(device) if b_norm else None
        self.activation = get_activation(activation)
        self.init_fn = nn.init.xavier_uniform_
        self.reset_parameters()

    def reset_parameters(self, init_fn=None):
        init_fn = init_fn or self.init_fn
        if init_fn is not None:
            init_fn(self.linear.weight, 1 / self.in_size)
        if self.bias:
            self.linear.bias.data.zero_()

    def forward(self, x, residual=None):
        """``residual`` (not in the reference's signature): added after the whole layer; with a plain Linear ->
        (Leaky)ReLU layer on the GPU the bias, the activation and this add are one kernel (ops.bias_act)."""
        act = self._fused_act()
        if act is not None and (act[0] != "none" or residual is not None) and x.is_cuda and x.dim() == 2 and self.out_size <= 1024:
            from .ops import bias_act, node_linear
            return bias_act(node_linear(x, self.linear.weight), self.linear.bias, act[0], act[1], residual)
        h = self._forward_modules(x)
        return h if residual is None else residual + h

    def _fused_act(self):
        """(name, slope) if the layer is Linear -> none | ReLU | LeakyReLU with no dropout / batch norm, else None"""
        if self.dropout is not None or self.b_norm is not None:
            return None
        a = self.activation
        if a is None:
            return ("none", 0.0)
        if type(a) is nn.ReLU:
            return ("relu", 0.0)
        if type(a) is nn.LeakyReLU:
            return ("leaky_relu", float(a.negative_slope))
        return None

    def _forward_modules(self, x):
        if x.is_cuda and x.dim() == 2:
            from .ops import node_linear
            h = node_linear(x, self.linear.weight, self.linear.bias)
        else:
            h = self.linear(x)
        if self.activation is not None:
            h = self.activation(h)
        if self.dropout is not None:
            h = self.dropout(h)
        if self.b_norm is not None:
            if h.shape[1] != self.out_size:
                h = self.b_norm(h.transpose(1, 2)).transpose(1, 2)
            else:
                h = self.b_norm(h)
        return h

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_size} -> {self.out_size})"


class MLP(nn.Module):
    """Stack of FCLayers under ``fully_connected`` (layers.py:120-149)."""

    def __init__(self, in_size, hidden_size, out_size, layers, mid_activation="relu", last_activation="none",
                 dropout=0.0, mid_b_norm=False, last_b_norm=False, device="cpu"):
        super().__init__()
        self.in_size, self.hidden_size, self.out_size = in_size, hidden_size, out_size
        self.fully_connected = nn.ModuleList()
        if layers <= 1:
            self.fully_connected.append(FCLayer(in_size, out_size, activation=last_activation, b_norm=last_b_norm,
                                                device=device, dropout=dropout))
        else:
            self.fully_connected.append(FCLayer(in_size, hidden_size, activation=mid_activation, b_norm=mid_b_norm,
                                                device=device, dropout=dropout))
            for _ in range(layers - 2):
                self.fully_connected.append(FCLayer(hidden_size, hidden_size, activation=mid_activation,
                                                    b_norm=mid_b_norm, device=device, dropout=dropout))
            self.fully_connected.append(FCLayer(hidden_size, out_size, activation=last_activation,
                                                b_norm=last_b_norm, device=device, dropout=dropout))

    def is_single_affine(self) -> bool:
        """True when the MLP is exactly one Linear (+bias) with nothing after it."""
        if len(self.fully_connected) != 1:
            return False
        fc = self.fully_connected[0]
        return fc.activation is None and fc.dropout is None and fc.b_norm is None

    def forward(self, x):
        for fc in self.fully_connected:
            x = fc(x)
        return x

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_size} -> {self.out_size})"
