"""Aggregator / scaler names of the reference -> op codes of the C ABI.

Names and meaning follow realworld_benchmark/nets/aggregators.py:74-93 and
scalers.py:21 of the reference; ``dirK-smooth`` (models/dgl/aggregators.py:76-78,
realworld_benchmark/README.md) is accepted as a spelling of ``dirK-av``.
Unknown names raise KeyError exactly like the reference's dict lookups
(dgn_layer.py:335-336).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

from . import _lib

EPS = 1e-8  # aggregators.py:5

# op codes (include/dgn_hip.h)
AGG_MEAN, AGG_SUM, AGG_MAX, AGG_MIN, AGG_STD, AGG_VAR, AGG_DIR_AV, AGG_DIR_WSUM, AGG_DIR_DX, AGG_DIR_DX_NO_ABS, AGG_X_IN = range(11)
X_IN_NAME = "__x_in__"     # pseudo-aggregator: copies h_in into the output row (posttrans([h || agg]) becomes one GEMM)
W_ABSNORM, W_BALANCED, W_SOFTMAX = range(3)
SCALE_IDENTITY, SCALE_AMPLIFICATION, SCALE_ATTENUATION = range(3)

_PLAIN = {"mean": AGG_MEAN, "sum": AGG_SUM, "max": AGG_MAX, "min": AGG_MIN, "std": AGG_STD, "var": AGG_VAR}
_DIR_RE = re.compile(r"^dir([1-3])-(av|smooth|dx|dx-no-abs|dx-balanced|0\.1|neg-0\.1)$")
_SCALERS = {"identity": SCALE_IDENTITY, "amplification": SCALE_AMPLIFICATION, "attenuation": SCALE_ATTENUATION}

AGGREGATOR_NAMES = tuple(list(_PLAIN) + [f"dir{k}-{s}" for s in ("av", "0.1", "neg-0.1", "dx", "dx-no-abs", "dx-balanced")
                                         for k in (1, 2, 3)])
SCALER_NAMES = tuple(_SCALERS)

Channel = Tuple[int, int, float]  # (kind, eig column, alpha)


def parse_aggregator(name: str) -> Tuple[int, Channel | None]:
    """name -> (op, channel or None)."""
    if name in _PLAIN:
        return _PLAIN[name], None
    if name == X_IN_NAME:
        return AGG_X_IN, None
    m = _DIR_RE.match(name)
    if m is None:
        raise KeyError(name)
    k, kind = int(m.group(1)), m.group(2)
    if kind in ("av", "smooth"):
        return AGG_DIR_AV, (W_ABSNORM, k, 0.0)
    if kind == "dx":
        return AGG_DIR_DX, (W_ABSNORM, k, 0.0)
    if kind == "dx-no-abs":
        return AGG_DIR_DX_NO_ABS, (W_ABSNORM, k, 0.0)
    if kind == "dx-balanced":
        return AGG_DIR_DX, (W_BALANCED, k, 0.0)
    return AGG_DIR_WSUM, (W_SOFTMAX, k, 0.1 if kind == "0.1" else -0.1)


def parse_scaler(name: str) -> int:
    return _SCALERS[name]


@dataclass
class Launch:
    """One kernel launch: a slice of the aggregator list with at most DGN_MAX_CH channels."""
    agg_offset: int
    ops: List[int] = field(default_factory=list)
    chs: List[int] = field(default_factory=list)
    channels: List[Channel] = field(default_factory=list)
    ch_offset: int = 0  # first plane of this launch in the plan-wide weight array


@dataclass
class AggPlan:
    aggregators: Tuple[str, ...]
    scalers: Tuple[str, ...]
    launches: List[Launch]
    applied_scalers: Tuple[int, ...]
    n_channels: int  # total weight planes over all launches

    @property
    def n_agg(self) -> int:
        return len(self.aggregators)

    @property
    def n_scalers(self) -> int:
        return len(self.applied_scalers)

    @property
    def channels(self) -> Tuple[Channel, ...]:
        return tuple(c for l in self.launches for c in l.channels)

    def out_width(self, F: int) -> int:
        return self.n_agg * self.n_scalers * F

    def needs_x_in(self) -> bool:
        return any(op in (AGG_DIR_DX, AGG_DIR_DX_NO_ABS, AGG_X_IN) for l in self.launches for op in l.ops)


def make_plan(aggregators: Sequence[str], scalers: Sequence[str]) -> AggPlan:
    aggregators = tuple(aggregators)
    scalers = tuple(scalers)
    if not aggregators:
        raise ValueError("at least one aggregator is required")
    if not scalers:
        raise ValueError("at least one scaler is required")
    parsed = [parse_aggregator(a) for a in aggregators]
    kinds = [parse_scaler(s) for s in scalers]
    # dgn_layer.py:170: the scaler concat only happens ``if len(self.scalers) > 1``
    applied = tuple(kinds) if len(kinds) > 1 else (SCALE_IDENTITY,)
    if len(applied) > _lib.DGN_MAX_SCALERS:
        raise ValueError(f"at most {_lib.DGN_MAX_SCALERS} scalers are supported")
    launches: List[Launch] = []
    cur = Launch(agg_offset=0)
    for i, (op, ch) in enumerate(parsed):
        new_ch = ch is not None and ch not in cur.channels
        if len(cur.ops) == _lib.DGN_MAX_AGG or (new_ch and len(cur.channels) == _lib.DGN_MAX_CH):
            launches.append(cur)
            cur = Launch(agg_offset=i)
            new_ch = ch is not None
        if new_ch:
            cur.channels.append(ch)
        cur.ops.append(op)
        cur.chs.append(cur.channels.index(ch) if ch is not None else 0)
    launches.append(cur)
    off = 0
    for l in launches:
        l.ch_offset = off
        off += len(l.channels)
    return AggPlan(aggregators, scalers, launches, applied, off)
