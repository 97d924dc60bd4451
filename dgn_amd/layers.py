"""FCLayer / MLP with the reference's constructor arguments, init and state_dict layout
(realworld_benchmark/nets/layers.py:21-154).  These are the dense pre/post-aggregation transforms.  The Linear inside goes
through ``ops.node_linear``: the library's own fp32-MFMA kernels (``dgn_linear_*`` up to 160 columns, ``dgn_gemm_*`` beyond) from a
few thousand rows on, ``torch.nn.functional.linear`` only for smaller batches.
"""
from __future__ import annotations

import torch
import torch.nn as nn

_ACTIVATIONS = {n.lower(): getattr(nn, n) for n in ("ReLU", "Sigmoid", "Tanh", "ELU", "SELU", "GLU", "LeakyReLU", "Softplus")}
SUPPORTED_ACTIVATION_MAP = set(c.__name__ for c in _ACTIVATIONS.values()) | {"None"}      # (the reference's public name, layers.py:4)


def get_activation(activation):
    """Activation module for a name in any case, ``None`` for "none"; a callable passes through (layers.py:7-18).  Unknown names fail
    the same assertion as the reference's lookup."""
    if callable(activation):
        return activation
    key = str(activation).lower()
    assert key == "none" or key in _ACTIVATIONS, "Unhandled activation function"
    return None if key == "none" else _ACTIVATIONS[key]()


class FCLayer(nn.Module):
    """Linear -> activation -> dropout -> batch-norm (layers.py:101-112).

    Weight init is ``xavier_uniform_`` with gain ``1 / in_size`` and a zero bias
    (layers.py:94-99), reproduced so that a fresh layer under the same seed matches.
    """

    def __init__(self, in_size, out_size, activation="relu", dropout=0.0, b_norm=False, bias=True, init_fn=None,
                 device="cpu"):
        super().__init__()
        self.in_size, self.out_size, self.bias = in_size, out_size, bias
        # (built on the CPU and moved, as the reference does -- layers.py:80: the default init's draws come from the CPU generator, so a
        #  fresh layer under the same seed has the reference's weights also with device="cuda")
        self.linear = nn.Linear(in_size, out_size, bias=bias).to(device)
        self.activation = get_activation(activation)
        self.dropout = nn.Dropout(dropout) if dropout else None      # (the reference passes device= here and would raise: quirk #4)
        self.b_norm = nn.BatchNorm1d(out_size).to(device) if b_norm else None
        self.init_fn = nn.init.xavier_uniform_      # (layers.py:91: the reference ignores its init_fn argument; so does this)
        self.reset_parameters()

    def reset_parameters(self, init_fn=None):
        """layers.py:94-99: the init function is called with ``1 / in_size`` as its second argument (xavier's gain); zero bias."""
        with torch.no_grad():
            (init_fn or self.init_fn)(self.linear.weight, 1.0 / self.in_size)
            if self.linear.bias is not None:
                self.linear.bias.zero_()

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
        for stage in (self.activation, self.dropout):
            if stage is not None:
                h = stage(h)
        if self.b_norm is not None:
            # (BatchNorm1d normalises dim 1: node features [N, F] as they are; an [B, N, F] input -- the reference's dense models,
            #  layers.py:108-109 -- with the feature axis moved there and back)
            h = self.b_norm(h) if h.dim() == 2 else self.b_norm(h.movedim(-1, 1)).movedim(1, -1)
        return h

    def extra_repr(self):
        return f"{self.in_size} -> {self.out_size}"


class MLP(nn.Module):
    """Stack of FCLayers under ``fully_connected`` (layers.py:120-149)."""

    def __init__(self, in_size, hidden_size, out_size, layers, mid_activation="relu", last_activation="none",
                 dropout=0.0, mid_b_norm=False, last_b_norm=False, device="cpu"):
        super().__init__()
        self.in_size, self.hidden_size, self.out_size = in_size, hidden_size, out_size
        # widths of the chain: in -> hidden x (layers - 1) -> out; every stage but the last takes the "mid" settings (layers.py:127-143)
        n = max(int(layers), 1)
        dims = [in_size] + [hidden_size] * (n - 1) + [out_size]
        self.fully_connected = nn.ModuleList(
            FCLayer(dims[i], dims[i + 1], activation=mid_activation if i < n - 1 else last_activation,
                    b_norm=mid_b_norm if i < n - 1 else last_b_norm, dropout=dropout, device=device) for i in range(n))

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

    def extra_repr(self):
        return f"{self.in_size} -> {self.out_size}"
