"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/dgn_hip.h declares (no compute without a GPU), the ctypes structs match the header's layout,
and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dgn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from dgn_amd import _lib
    header = open(os.path.join(ROOT, "include", "dgn_hip.h")).read()
    declared = set(re.findall(r"\b(dgn_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.dgn_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header(lib):
    from dgn_amd import _lib
    # every ctypes mirror against the sizeof the LIBRARY was compiled with (dgn_sizeof), then a few sizes spelled out for LP64
    for name in ("DgnGraph", "DgnChannel", "DgnAggSpec", "DgnMsg", "DgnMsgGrad", "DgnBnGrad", "DgnTowersLayer", "DgnTowersGrads", "DgnDegreeClasses",
                 "DgnDcLayout", "DgnDenseLayer", "DgnDenseGrads", "DgnBlockTable", "DgnBlockLayer", "DgnBlockGrads"):
        assert C.sizeof(getattr(_lib, name)) == lib.dgn_sizeof(name.encode()) > 0, name
    assert lib.dgn_sizeof(b"NoSuchStruct") == 0
    assert C.sizeof(_lib.DgnChannel) == 16
    assert C.sizeof(_lib.DgnGraph) == 8 * 9 + 4 * 2 + 8 * 2 + 8 + 8 + 8 + 8 + 8 + 8 * 5       # (+ max_in_degree padded, n_src, row_base, blk_cut, blk_gap padded; gblk_desc, n_gblk, gblk_rows padded, csc_order, dst_csr)
    assert C.sizeof(_lib.DgnAggSpec) == 4 * (1 + 16 + 16 + 1 + 1 + 4 + 1 + 1 + 1 + 1 + 1) + 8
    assert C.sizeof(_lib.DgnMsg) == 8 * 9 + 8 + 8                               # (+ edge_type, n_edge_types padded)
    assert C.sizeof(_lib.DgnMsgGrad) == 8 * 8 + 8                                # (+ accumulate, padded)
    header = open(os.path.join(ROOT, "include", "dgn_hip.h")).read()
    assert f"#define DGN_MAX_AGG {_lib.DGN_MAX_AGG}" in header
    assert f"#define DGN_MAX_CH {_lib.DGN_MAX_CH}" in header
    assert f"#define DGN_MAX_SCALERS {_lib.DGN_MAX_SCALERS}" in header


def test_argument_validation_without_gpu(lib):
    """Validation errors are reported before anything touches the device."""
    from dgn_amd import _lib
    g = _lib.DgnGraph()
    g.n_nodes, g.n_edges = 4, 0
    spec = _lib.DgnAggSpec()
    spec.n_agg = 0
    msg = _lib.DgnMsg()
    rc = lib.dgn_agg_forward(C.byref(g), C.byref(spec), C.byref(msg), None, 0, None, None, 0, None, 0, None)
    assert rc == -1 and b"null CSR" in lib.dgn_last_error()
    assert lib.dgn_agg_workspace_bytes(C.byref(g), C.byref(spec), 8) == 0
    rc = lib.dgn_edge_weights(C.byref(g), None, None, None, 4, 1, None, None, 0, None, 0, None)
    assert rc == -1


def test_plan_and_registry():
    import dgn_amd
    from dgn_amd import spec
    assert len(dgn_amd.AGGREGATOR_NAMES) == 24 and len(dgn_amd.SCALER_NAMES) == 3
    plan = dgn_amd.make_plan("mean max dir1-dx dir1-av dir1-dx-balanced".split(), ["amplification"])
    assert plan.applied_scalers == (spec.SCALE_IDENTITY,)          # lone scaler is not applied
    assert plan.n_channels == 2 and plan.launches[0].chs == [0, 0, 0, 0, 1]
    many = dgn_amd.make_plan(list(dgn_amd.AGGREGATOR_NAMES), ["identity", "attenuation"])
    assert sum(len(l.ops) for l in many.launches) == 24
    assert all(len(l.channels) <= 4 and len(l.ops) <= 16 for l in many.launches)
    assert dgn_amd.AGGREGATORS["dir2-smooth"].name == "dir2-smooth"
    with pytest.raises(KeyError):
        dgn_amd.AGGREGATORS["dir4-dx"]
    with pytest.raises(KeyError):
        dgn_amd.SCALERS["linear"]


def test_state_dict_layout_matches_reference(golden):
    """Same keys and shapes as the reference layers' state_dict (fixtures hold the reference's)."""
    import dgn_amd
    g = golden("g4_layers")
    for name in g["cases"].tolist():
        meta = g[f"{name}/meta"].tolist()
        layer = dgn_amd.DGNLayer(in_dim=int(meta[1]), out_dim=int(meta[2]), dropout=0.0, graph_norm=True, batch_norm=True,
                                 aggregators=meta[3], scalers=meta[4], avg_d={"log": torch.tensor(float(meta[5]))},
                                 type_net=meta[0], residual=True, towers=int(meta[6]), divide_input=bool(int(meta[7])),
                                 edge_features=bool(int(meta[8])), edge_dim=int(meta[9]), pretrans_layers=int(meta[10]),
                                 posttrans_layers=int(meta[11])).model
        ref = {k[len(name) + 5:]: g[k].shape for k in g.files if k.startswith(f"{name}/sd::")}
        mine = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
        assert mine == {k: tuple(v) for k, v in ref.items()}, name


def test_fresh_init_matches_reference_under_seed(golden):
    """FCLayer init (xavier_uniform gain 1/in_size, zero bias; layers.py:94-99) under the same seed."""
    import dgn_amd
    g = golden("g4_layers")
    name = "simple_fresh_init"
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(10, 10, 0.0, True, True, "mean dir1-dx", "identity", {"log": torch.tensor(1.1)}, "simple", True).model
    for k, v in layer.state_dict().items():
        assert torch.equal(v, torch.from_numpy(g[f"{name}/sd::{k}"])), k


def test_no_cpu_fallback():
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    src = torch.tensor([0, 1])
    with pytest.raises((dgn_amd._lib.DgnError, RuntimeError, AssertionError)):
        g = dgn_amd.DGNGraph(src, src.flip(0), 2, eig=torch.randn(2, 2))
        directional_aggregate(g, dgn_amd.make_plan(["mean"], ["identity"]), 1.0, x_src=torch.randn(2, 4))


def test_argument_validation_runs_before_any_device_work(lib):
    """Maximum sizes and malformed arguments are rejected by the host-side checks with an error code and a message
    (no kernel is launched, so this runs without a GPU): int32 CSR range, widths over the kernels' limits, nulls."""
    from dgn_amd import _lib
    err = lambda: lib.dgn_last_error().decode()
    g = _lib.DgnGraph()
    g.n_nodes, g.n_edges = 2 ** 31, 10                      # beyond the int32 CSR range
    spec, msg = _lib.DgnAggSpec(), _lib.DgnMsg()
    spec.n_agg, spec.n_scalers, spec.n_towers, spec.eps = 1, 1, 1, 1e-8
    msg.F, msg.x_src, msg.ld_src = 4, 1, 4                  # (a non-null dummy pointer: never dereferenced)
    rc = lib.dgn_agg_forward(C.byref(g), C.byref(spec), C.byref(msg), None, 0, None, None, 0, None, 0, None)
    assert rc == -1 and "int32" in err()
    g.n_nodes = 10
    rc = lib.dgn_agg_forward(C.byref(g), C.byref(spec), C.byref(msg), None, 0, None, None, 0, None, 0, None)
    assert rc == -1 and "null CSR" in err()
    g.indptr = g.src = 1                                    # (dummies again)
    spec.n_agg = 99                                         # more aggregators than one launch takes
    rc = lib.dgn_agg_forward(C.byref(g), C.byref(spec), C.byref(msg), None, 0, None, None, 0, None, 0, None)
    assert rc == -1 and "n_agg" in err()
    assert lib.dgn_bn_tail_forward(5, 2000, None, 2000, None, None, None, None, 0.1, 1e-5, 1, 0, None, None, None, None, None, 0, None, None) == -1
    assert "F <= 1024" in err()
    assert lib.dgn_scale_combine_forward(5, 1, 3, 4, None, None, None, None, None, 0, None) == -1      # S = 3 without a scale table
    assert lib.dgn_scale_combine_backward(5, 600, 1, 8, None, 0, None, None, None, None, None, 0, None, None) == -1
    assert "4096" in err()
    assert lib.dgn_bn_tail_workspace_bytes(0, 8) == 0 and lib.dgn_bn_tail_workspace_bytes(1000, 8) > 0
    assert lib.dgn_scale_combine_backward_workspace_bytes(1000, 5, 14) > 0


def test_linear_entry_points_validate_before_any_device_work(lib):
    """dgn_linear_*: width limits, dense-row and alignment requirements, workspace size -- all host-side checks."""
    err = lambda: lib.dgn_last_error().decode()
    assert lib.dgn_linear_supported(70, 140, 0) == 1 and lib.dgn_linear_supported(70, 140, 1) == 1
    assert lib.dgn_linear_supported(71, 140, 0) == 0 and lib.dgn_linear_supported(70, 162, 0) == 0      # odd / too wide
    assert lib.dgn_linear_supported(160, 160, 0) == 1 and lib.dgn_linear_supported(160, 160, 1) == 0    # 100 tiles > 45
    a = 1 << 12                                                                                        # dummy aligned pointer, never dereferenced
    assert lib.dgn_linear_forward(10, 71, 140, 1, a, 71, 0, a, 71, 0, 0, None, 0, a, 140, 0, None) == -1 and "even" in err()
    assert lib.dgn_linear_forward(10, 70, 140, 1, a, 72, 0, a, 70, 0, 0, None, 0, a, 140, 0, None) == -1 and "dense rows" in err()
    assert lib.dgn_linear_forward(10, 70, 140, 1, a + 4, 70, 0, a, 70, 0, 0, None, 0, a, 140, 0, None) == -1
    assert lib.dgn_linear_forward(10, 70, 140, 1, None, 70, 0, a, 70, 0, 0, None, 0, a, 140, 0, None) == -1 and "null" in err()
    assert lib.dgn_linear_forward(0, 70, 140, 1, None, 70, 0, None, 70, 0, 0, None, 0, None, 140, 0, None) == 0      # no rows: nothing to do
    need = lib.dgn_linear_wgrad_workspace_bytes(1000, 70, 140, 1)
    assert need > 0 and lib.dgn_linear_wgrad_workspace_bytes(1000, 70, 141, 1) == 0
    assert lib.dgn_linear_wgrad(1000, 70, 140, 1, a, 140, 0, a, 70, 0, a, 70, 0, None, 0, a, need - 1, None) == -1 and "workspace" in err()
    assert lib.dgn_linear_wgrad(1000, 64, 140, 1, a, 140, 0, a, 64, 0, a, 64, 0, a, 0, a, need, None) == -1 and "bias gradient" in err()
    assert lib.dgn_linear_combine_forward(10, 84, 5, 4, 14, a, 0, a, 84, 0, a, None, None, a, 70, None) == -1 and "3 scalers" in err()
    assert lib.dgn_linear_combine_forward(10, 84, 5, 3, 14, a, 0, a, 84, 0, None, None, None, a, 70, None) == -1       # 3 scalers need the table
    assert lib.dgn_linear_combine_backward_input(10, 5, 3, 13, 84, a, 0, a, a, 84, 0, a, 0, None) == -1 and "even f_out" in err()
    assert lib.dgn_linear_combine_backward_weight(10, 5, 3, 14, 84, a, 0, a, a, 0, a, 84, 0, None, 0, None) == -1 and "workspace" in err()
