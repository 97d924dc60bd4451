"""dgn_amd: MI355X-native implementation of the DGN directional-aggregation layer.

Host side in Python on PyTorch-ROCm (device memory, streams, torch.distributed);
the aggregation itself is hand-written HIP for gfx950 behind the C ABI of
``include/dgn_hip.h`` (``libdgn_hip.so``, loaded with ctypes).  GPU only: there is no
CPU fallback anywhere in this package.
"""
from .graph import DGNGraph, as_dgn_graph, compute_edge_weights
from .spec import AGGREGATOR_NAMES, SCALER_NAMES, make_plan
from .ops import directional_aggregate
from .layers import FCLayer, MLP, get_activation
from .dgn_layer import (AGGREGATORS, SCALERS, DGNLayer, DGNLayerComplex, DGNLayerSimple, DGNLayerTower, DGNTower, EdgeTypeFeatures, get_dropout_state, reset_dropout_state, set_dropout_state)
from .readout import VirtualNode, max_nodes, mean_nodes, readout, sum_nodes
from .eig import laplacian_eigvecs

__all__ = ["DGNGraph", "as_dgn_graph", "compute_edge_weights", "make_plan", "directional_aggregate", "FCLayer", "MLP",
           "get_activation", "AGGREGATORS", "SCALERS", "DGNLayer", "DGNLayerSimple", "DGNLayerComplex", "DGNLayerTower",
           "DGNTower", "EdgeTypeFeatures", "AGGREGATOR_NAMES", "SCALER_NAMES", "VirtualNode", "sum_nodes", "mean_nodes", "max_nodes", "readout",
           "laplacian_eigvecs"]
