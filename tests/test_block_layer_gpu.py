"""The graph-block layer route (csrc/dgn_blk_layer.hip, ops.block_layer: batches at the reference's own batch size as five launches per
step) in front of the oracle and of the streaming routes, through the layers' DEFAULT dispatch.

* every layer type and every BASELINE / shipped-json shape at oracle sizes: output, d h, every parameter gradient, BatchNorm running
  statistics and ``num_batches_tracked`` vs the oracle (reference: nets/dgn_layer.py:103-132, :178-202, :254-325);
* the aggregate rows formed in LDS vs the streaming sweep (``ops.directional_aggregate``) on the same inputs;
* the same layer through the streaming whole-layer route: outputs and gradients agree to fp32 rounding;
* run-to-run bit reproducibility (no atomics anywhere on the route);
* the route is really taken (a counter on ``ops.block_layer``), and left for batches / shapes outside its domain.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_layer(type_net, F_, aggs, scalers, graph_norm, avg, towers=5, seed=1, o1_weights=True):
    import dgn_amd
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, graph_norm, True, aggs, scalers, {"log": torch.tensor(avg)}, type_net, True, towers=towers,
                             edge_features=False, edge_dim=0).model
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():          # O(1) weights: the reference's init (gain 1 / in_size) makes layer outputs ~1e-3
        for p in layer.parameters():
            if p.dim() == 2 and o1_weights:
                p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
            else:
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    return layer, gen


def _count_route(monkeypatch):
    import dgn_amd
    taken = []
    real = dgn_amd.ops.block_layer
    monkeypatch.setattr(dgn_amd.ops, "block_layer", lambda *a, **k: taken.append(1) or real(*a, **k))
    return taken


def _vs_oracle(monkeypatch, type_net, F_, aggs, scalers, graph_norm, b, towers=5, expect_route=True):
    import dgn_amd
    from oracle import dgn_oracle as orc
    from parity_util import check
    from test_shipped_configs_gpu import _check_grad
    dev = torch.device("cuda")
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    avg = float(torch.log(torch.bincount(dst, minlength=N).float() + 1).mean())
    layer, gen = _make_layer(type_net, F_, aggs, scalers, graph_norm, avg, towers, o1_weights=type_net != "towers")
    h, ct = torch.randn(N, F_, generator=gen), torch.randn(N, F_, generator=gen)

    def oracle(dtype):
        sd = {k: (v.detach().to(dtype).requires_grad_("running" not in k) if v.dtype.is_floating_point else v.clone())
              for k, v in layer.state_dict().items()}
        names = [k for k, v in sd.items() if v.dtype.is_floating_point and v.requires_grad]
        cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(avg, dtype=dtype), graph_norm=graph_norm, batch_norm=True,
                   residual=True, towers=towers if type_net == "towers" else 1, divide_input=True, edge_features=False)
        hh = h.to(dtype).requires_grad_(True)
        y, stats = orc.layer_forward(type_net, sd, cfg, src, dst, N, b["eig"].to(dtype), hh, None, b["snorm_n"].to(dtype), training=True)
        return y, torch.autograd.grad(y, [hh] + [sd[k] for k in names], ct.to(dtype)), names, stats

    y32, g32, names, stats = oracle(torch.float32)
    y64, g64, _, _ = oracle(torch.float64)
    taken = _count_route(monkeypatch)
    layer = layer.to(dev).train()
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=b["eig"].to(dev))
    hd = h.to(dev).requires_grad_(True)
    y = layer(graph, hd, None, b["snorm_n"].to(dev))
    assert bool(taken) == expect_route, "the graph-block route was " + ("not taken" if expect_route else "taken")
    params = dict(layer.named_parameters())
    gd = torch.autograd.grad(y, [hd] + [params[k] for k in names], ct.to(dev))
    strict = not any(a in ("std", "var") for a in aggs.split())      # (std / var lists: the reference's own fp32 evaluation is unstable)
    check(y, y32, y64, f"block {type_net} F={F_} y", rtol=2e-5, atol=2e-5, abs_scale=1.0, max_escape_fraction=0.0 if strict else None)
    for a, r32, r64, k in zip(gd, g32, g64, ["h"] + names):
        if strict:
            _check_grad(a, r32, r64, f"block {type_net} F={F_} {k}", strict=True)
        else:      # (std / var lists: the looser caps, named)
            check(a, r32, r64, f"block {type_net} F={F_} {k}", rtol=1e-4, atol=2e-5, max_escape_fraction=0.01, max_local_fraction=0.05)
    for k, v in (stats or {}).items():
        np.testing.assert_allclose(layer.state_dict()[k].cpu().numpy(), v.numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    for k, v in layer.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1, k


def test_c2_towers_batch_vs_oracle(monkeypatch):
    """BASELINE configs[1] at the reference's batch size: towers x 5, hidden 70, mean max min dir1-av dir1-dx x three scalers."""
    from dgn_amd import synth
    _vs_oracle(monkeypatch, "towers", 70, "mean max min dir1-av dir1-dx", "identity amplification attenuation", True,
               synth.molecule_batch(128, seed=41, extra_bonds=3.9, eig_dim=6))


@pytest.mark.parametrize("hidden,towers", [(45, 5), (32, 4)])
def test_towers_other_widths_vs_oracle(monkeypatch, hidden, towers):
    from dgn_amd import synth
    _vs_oracle(monkeypatch, "towers", hidden, "mean max min dir1-av dir1-dx", "identity amplification attenuation", True,
               synth.molecule_batch(60, seed=7, extra_bonds=3.9, eig_dim=6), towers=towers)


def test_c1_simple_hidden75_vs_oracle(monkeypatch):
    from dgn_amd import synth
    _vs_oracle(monkeypatch, "simple", 75, "mean dir1-dx-no-abs", "identity amplification attenuation", True,
               synth.molecule_batch(128, seed=41, extra_bonds=3.9, eig_dim=6))


@pytest.mark.parametrize("type_net", ["simple", "complex"])
def test_c3_knn_hidden65_vs_oracle(monkeypatch, type_net):
    """CIFAR10-like 8-NN graphs of 85 - 150 nodes at hidden 65: a block (rows, row pointers, g_yr, three gradient accumulators) exceeds the
    LDS plan -- the layer must leave the route for the streaming kernels, and still meet the oracle."""
    from dgn_amd import synth
    _vs_oracle(monkeypatch, type_net, 65, "mean dir1-dx dir2-dx", "identity", True, synth.knn_batch(8, seed=41), expect_route=False)


@pytest.mark.parametrize("type_net", ["simple", "complex"])
def test_small_knn_graphs_on_the_route_vs_oracle(monkeypatch, type_net):
    """8-NN graphs of 40 - 70 nodes (in-degrees 0 .. ~20, zero in-degree rows): several row chunks per block, multi-step edge loops,
    the weight-gradient partial accumulated across chunks."""
    from dgn_amd import synth
    _vs_oracle(monkeypatch, type_net, 36, "mean dir1-dx dir2-dx", "identity", True, synth.knn_batch(10, seed=41, n_lo=40, n_hi=70))


def test_zinc_json_complex_hidden45_vs_oracle(monkeypatch):
    from dgn_amd import synth
    _vs_oracle(monkeypatch, "complex", 45, "mean dir1-dx dir1-av", "identity amplification attenuation", True,
               synth.molecule_batch(128, seed=41, extra_bonds=3.9, eig_dim=6))


@pytest.mark.parametrize("scalers", ["identity", "identity amplification attenuation"])
def test_c4_molhiv_simple_vs_oracle(monkeypatch, scalers):
    from dgn_amd import synth
    _vs_oracle(monkeypatch, "simple", 70, "mean max min dir1-dx dir1-av", scalers, False,
               synth.molecule_batch(200, seed=41, n_lo=10, n_hi=41, extra_bonds=4.3, eig_dim=4))


@pytest.mark.parametrize("type_net,aggs", [("simple", "mean sum max min std var dir1-av dir2-dx-no-abs dir1-0.1"),
                                           ("complex", "std dir1-dx-balanced dir2-neg-0.1 dir3-av max"),
                                           ("complex", "sum var min dir2-dx")])
def test_other_aggregator_lists_vs_oracle(monkeypatch, type_net, aggs):
    """every aggregator family of nets/aggregators.py:74-93 on the route (softmax and balanced channels, var / std, three eig columns)"""
    from dgn_amd import synth
    _vs_oracle(monkeypatch, type_net, 24, aggs, "identity amplification attenuation", True, synth.molecule_batch(50, seed=3, extra_bonds=3.9, eig_dim=6))


def test_isolated_nodes_vs_oracle(monkeypatch):
    from dgn_amd import synth
    b = dict(synth.molecule_batch(100, seed=47, extra_bonds=3.9, eig_dim=6))
    N = int(b["num_nodes"])
    cut = torch.rand(N, generator=torch.Generator().manual_seed(48)) < 0.08
    keep = ~cut[b["dst"]]
    b["src"], b["dst"] = b["src"][keep], b["dst"][keep]
    _vs_oracle(monkeypatch, "complex", 30, "mean max min dir1-dx dir1-av", "identity amplification attenuation", True, b)


def test_aggregate_rows_equal_the_streaming_sweep():
    """the aggregate rows formed in LDS (debug tap of the forward) vs ops.directional_aggregate on the same h / eig"""
    import dgn_amd
    from dgn_amd import ops, synth
    from dgn_amd.spec import make_plan
    dev = torch.device("cuda")
    b = synth.molecule_batch(64, seed=5, extra_bonds=3.9, eig_dim=6)
    N = int(b["num_nodes"])
    aggs = "mean max min std dir1-av dir1-dx dir2-dx-no-abs"
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    layer, gen = _make_layer("simple", 40, aggs, "identity amplification attenuation", True, avg)
    layer = layer.to(dev).train()
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    h = torch.randn(N, 40, generator=gen).to(dev).requires_grad_(True)
    ops._BLK_DBG = {}
    try:
        layer(graph, h, None, b["snorm_n"].to(dev))
        agg = ops._BLK_DBG["agg"]
    finally:
        ops._BLK_DBG = None
    plan = make_plan(aggs.split(), ["identity"])
    ref = ops.directional_aggregate(graph, plan, avg, x_src=h.detach(), x_in=h.detach(), eig=b["eig"].to(dev))
    np.testing.assert_allclose(agg.cpu().numpy(), ref.cpu().numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("type_net,F_,aggs,scalers", [("towers", 70, "mean max min dir1-av dir1-dx", "identity amplification attenuation"),
                                                      ("complex", 45, "mean dir1-dx dir1-av", "identity amplification attenuation"),
                                                      ("simple", 65, "mean dir1-dx dir2-dx", "identity")])
def test_block_route_vs_streaming_route_and_reproducible(monkeypatch, type_net, F_, aggs, scalers):
    import dgn_amd
    from dgn_amd import synth
    dev = torch.device("cuda")
    b = synth.knn_batch(6, seed=3, n_lo=40, n_hi=70) if type_net == "simple" else synth.molecule_batch(96, seed=11, extra_bonds=3.9, eig_dim=6)
    N = int(b["num_nodes"])
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    layer, gen = _make_layer(type_net, F_, aggs, scalers, True, avg, o1_weights=type_net != "towers")
    layer = layer.to(dev).train()
    sd0 = {k: v.clone() for k, v in layer.state_dict().items()}
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    h = torch.randn(N, F_, generator=gen).to(dev)
    ct = torch.randn(N, F_, generator=gen).to(dev)
    snorm = b["snorm_n"].to(dev)

    def run(max_nodes):
        monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", max_nodes)
        layer.load_state_dict(sd0)
        hh = h.clone().requires_grad_(True)
        y = layer(graph, hh, None, snorm)
        names = [k for k, _ in layer.named_parameters()]
        g = torch.autograd.grad(y, [hh] + list(layer.parameters()), ct)
        return y.detach(), dict(zip(["h"] + names, g)), {k: v.clone() for k, v in layer.state_dict().items() if "running" in k}

    taken = _count_route(monkeypatch)
    y_b, g_b, st_b = run(32768)
    assert taken
    y_b2, g_b2, _ = run(32768)
    assert torch.equal(y_b, y_b2) and all(torch.equal(g_b[k], g_b2[k]) for k in g_b), "the block route is not run-to-run reproducible"
    n_taken = len(taken)
    y_s, g_s, st_s = run(0)
    assert len(taken) == n_taken
    np.testing.assert_allclose(y_b.cpu().numpy(), y_s.cpu().numpy(), rtol=1e-5, atol=1e-5)
    for k in g_b:
        scale = max(1.0, float(g_s[k].abs().max()))
        # (max / min / |.| routings may flip between two fp32 evaluations of the same tie: a handful of entries, bounded)
        bad = (g_b[k] - g_s[k]).abs() > 2e-5 * scale + 1e-4 * g_s[k].abs()
        assert int(bad.sum()) <= max(1, int(2e-3 * bad.numel())), f"{k}: {int(bad.sum())} of {bad.numel()} entries differ between the routes"
    for k in st_b:
        np.testing.assert_allclose(st_b[k].cpu().numpy(), st_s[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_route_left_outside_its_domain(monkeypatch):
    """edge features, eval mode, and batches over the node limit keep the streaming routes"""
    import dgn_amd
    from dgn_amd import synth
    dev = torch.device("cuda")
    b = synth.molecule_batch(40, seed=2, extra_bonds=3.9, eig_dim=6)
    N = int(b["num_nodes"])
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    snorm = b["snorm_n"].to(dev)
    taken = _count_route(monkeypatch)
    layer, gen = _make_layer("complex", 30, "mean dir1-dx", "identity amplification attenuation", True, avg)
    layer = layer.to(dev)
    h = torch.randn(N, 30, generator=gen).to(dev).requires_grad_(True)
    layer.eval()
    layer(graph, h, None, snorm)
    assert not taken
    layer.train()
    monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", N - 1)
    layer(graph, h, None, snorm)
    assert not taken
    monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", 32768)
    layer(graph, h, None, snorm)
    assert len(taken) == 1
    torch.manual_seed(0)
    le = dgn_amd.DGNLayer(30, 30, 0.0, True, True, "mean dir1-dx", "identity", {"log": torch.tensor(avg)}, "complex", True, edge_features=True,
                          edge_dim=6).model.to(dev).train()
    le(graph, h, torch.randn(graph.num_edges, 6, device=dev), snorm)
    assert len(taken) == 1
    # a large posttrans (the HIV json's simple layer: hidden 70, five aggregators x three scalers = 1050 x 70 weights per workgroup) is
    # left to the streaming kernels at the DEFAULT bound (measured: 0.236 ms on the route, 0.149 off it), taken when the bound is lifted
    lh, _ = _make_layer("simple", 70, "mean max min dir1-dx dir1-av", "identity amplification attenuation", False, avg)
    lh = lh.to(dev).train()
    h70 = torch.randn(N, 70, generator=gen).to(dev).requires_grad_(True)
    monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_POST", 40960)
    graph2 = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))      # (a fresh table: the answer is cached per table)
    lh(graph2, h70, None, snorm)
    assert len(taken) == 1
    monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_POST", 1 << 30)
    graph3 = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    lh(graph3, h70, None, snorm)
    assert len(taken) == 2


@pytest.mark.parametrize("type_net", ["towers", "complex"])
def test_padded_batch_through_the_static_block_table(monkeypatch, type_net):
    """A batch inside capacity-padded static buffers (hipgraph.PaddedBatch: what a captured step replays over) on the route: the table
    is written from the graph sizes alone (slots from the row pointers, unused entries, BatchNorm over n_valid rows); valid rows and
    every gradient equal the unpadded batch's, padding rows of the output and of d h are zeros, the running statistics agree."""
    import copy
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.hipgraph import PaddedBatch
    dev = torch.device("cuda")
    b = synth.molecule_batch(40, seed=17, extra_bonds=3.9, eig_dim=6)
    N, E, sizes = int(b["num_nodes"]), b["src"].numel(), [int(s) for s in b["sizes"]]
    F_ = 70 if type_net == "towers" else 45
    aggs = "mean max min dir1-av dir1-dx" if type_net == "towers" else "mean dir1-dx dir1-av"
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    layer, gen = _make_layer(type_net, F_, aggs, "identity amplification attenuation", True, avg, o1_weights=type_net != "towers")
    layer = layer.to(dev).train()
    layer_p = copy.deepcopy(layer)
    h, ct = torch.randn(N, F_, generator=gen), torch.randn(N, F_, generator=gen)
    taken = _count_route(monkeypatch)
    # unpadded
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    hd = h.to(dev).requires_grad_(True)
    y = layer(graph, hd, None, b["snorm_n"].to(dev))
    y.backward(ct.to(dev))
    assert len(taken) == 1
    # padded
    n_cap, e_cap, g_cap = N + 75, E + 120, 48
    pb = PaddedBatch(n_cap, e_cap, dev, eig_dim=6)
    pb.graph.set_block_capacity(g_cap, max(sizes), 4 * max(sizes))
    snorm = pb.add_node_tensor("snorm", 1)
    with pytest.raises(ValueError):              # (a static table and no sizes: refused, a stale table would be silent garbage)
        pb.load(b["src"].to(dev), b["dst"].to(dev), N, b["eig"].to(dev), node={"snorm": b["snorm_n"].to(dev)})
    pb.load(b["src"].to(dev), b["dst"].to(dev), N, b["eig"].to(dev), node={"snorm": b["snorm_n"].to(dev)}, graph_sizes=sizes)
    hp = torch.zeros(n_cap, F_, device=dev)
    hp[:N] = h.to(dev)
    hp.requires_grad_(True)
    ctp = torch.full((n_cap, F_), float("nan"), device=dev)     # (the padding rows' cotangent must never be read)
    ctp[:N] = ct.to(dev)
    yp = layer_p(pb.graph, hp, None, snorm)
    yp.backward(ctp)
    assert len(taken) == 2, "the padded batch left the route"
    pb.graph.check_deferred()
    assert torch.equal(yp.detach()[:N], y.detach()) and float(yp.detach()[N:].abs().max()) == 0.0
    assert torch.equal(hp.grad[:N], hd.grad) and float(hp.grad[N:].abs().max()) == 0.0
    for (k, a), (_, c) in zip(layer_p.named_parameters(), layer.named_parameters()):
        np.testing.assert_allclose(a.grad.cpu().numpy(), c.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * max(1.0, float(c.grad.abs().max())), err_msg=k)
    for (k, a), (_, c) in zip(layer_p.state_dict().items(), layer.state_dict().items()):
        if "running" in k or "num_batches" in k:
            assert torch.equal(a, c), k
    # the evaluation forward over the same padded buffers (eval() under no_grad: running statistics, zero padding rows)
    layer.eval(); layer_p.eval()
    with torch.no_grad():
        ye = layer(graph, h.to(dev), None, b["snorm_n"].to(dev))
        ype = layer_p(pb.graph, hp.detach(), None, snorm)
    assert len(taken) == 4
    assert torch.equal(ype[:N], ye) and float(ype[N:].abs().max()) == 0.0


def test_padded_batch_beyond_the_block_capacity_is_reported(monkeypatch):
    """A graph larger than the static table's capacity: too many rows are refused when the table is loaded (host side: the sizes are
    known), too many edges are skipped by the kernels and reported by check_deferred."""
    import dgn_amd
    from dgn_amd import _lib, synth
    from dgn_amd.hipgraph import PaddedBatch
    dev = torch.device("cuda")
    b = synth.molecule_batch(20, seed=3, extra_bonds=3.9, eig_dim=6)
    N, E, sizes = int(b["num_nodes"]), b["src"].numel(), [int(s) for s in b["sizes"]]
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    layer, gen = _make_layer("simple", 32, "mean dir1-dx", "identity", True, avg)
    layer = layer.to(dev).train()
    pb = PaddedBatch(N + 10, E + 10, dev, eig_dim=6)
    pb.graph.set_block_capacity(24, max(sizes) - 1, 4 * max(sizes))
    with pytest.raises(ValueError):
        pb.load(b["src"].to(dev), b["dst"].to(dev), N, b["eig"].to(dev), graph_sizes=sizes)
    pb.graph.set_block_capacity(24, max(sizes), 8)                 # (every molecule has more than 8 directed edges)
    pb.load(b["src"].to(dev), b["dst"].to(dev), N, b["eig"].to(dev), graph_sizes=sizes)
    snorm = torch.ones(N + 10, 1, device=dev)
    # poison the caching allocator's free blocks: whatever the step leaves unwritten would come back as NaN (the skipped
    # blocks' y0 / d h rows and -- block 0 skipped -- d gamma / d beta were uninitialised memory until check_deferred raised)
    junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 20, 1 << 18, 1 << 16, (N + 10) * 32, (N + 10) * 32, 4096, 512, 64)]
    del junk
    hp = torch.randn(N + 10, 32, device=dev, requires_grad=True)
    y = layer(pb.graph, hp, None, snorm)
    y.sum().backward()
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(hp.grad).all())
    for k, v in layer.named_parameters():
        assert v.grad is not None and bool(torch.isfinite(v.grad).all()), k
    for k, v in layer.state_dict().items():
        assert bool(torch.isfinite(v.float()).all()), k
    with pytest.raises(_lib.DgnError):
        pb.graph.check_deferred()


def test_block_table_from_batch_num_nodes_needs_no_read_back(monkeypatch):
    """With dgl.batch's graph sizes on the graph (``batch_num_nodes``) the table is rows only (slots = -1: the kernels read the row
    pointers; the LDS plan uses rows x largest in-degree): same blocks, bit-identical layer output and gradients."""
    import copy
    import dgn_amd
    from dgn_amd import synth
    dev = torch.device("cuda")
    b = synth.molecule_batch(50, seed=23, extra_bonds=3.9, eig_dim=6)
    N = int(b["num_nodes"])
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    layer, gen = _make_layer("towers", 70, "mean max min dir1-av dir1-dx", "identity amplification attenuation", True, avg, o1_weights=False)
    layer = layer.to(dev).train()
    layer2 = copy.deepcopy(layer)
    h, ct = torch.randn(N, 70, generator=gen).to(dev), torch.randn(N, 70, generator=gen).to(dev)
    snorm = b["snorm_n"].to(dev)
    taken = _count_route(monkeypatch)
    g1 = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    g2 = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    g2.batch_num_nodes = [int(s) for s in b["sizes"]]
    outs = []
    for g, l in ((g1, layer), (g2, layer2)):
        hd = h.clone().requires_grad_(True)
        y = l(g, hd, None, snorm)
        y.backward(ct)
        outs.append((y.detach(), hd.grad, [p.grad for p in l.parameters()]))
    assert len(taken) == 2
    t1, t2 = g1.block_table(), g2.block_table()
    assert torch.equal(t1["desc"][:, :2], t2["desc"][:, :2]) and int(t2["desc"][:, 2:].max()) == -1 and int(t1["desc"][:, 2:].min()) >= 0
    assert t2["max_edges"] >= t1["max_edges"]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, c in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, c)


@pytest.mark.parametrize("type_net,F_,aggs", [("towers", 70, "mean max min dir1-av dir1-dx"), ("complex", 45, "mean dir1-dx dir1-av"), ("simple", 75, "mean dir1-dx-no-abs")])
def test_evaluation_forward_on_the_route(monkeypatch, type_net, F_, aggs):
    """eval() under no_grad (the reference's evaluation loops, train/train_molecules_graph_regression.py:47-66): the route's forward with
    BatchNorm on its RUNNING statistics -- against the oracle in evaluation mode and the streaming kernels; statistics and counters untouched."""
    import copy
    import dgn_amd
    from dgn_amd import synth
    from oracle import dgn_oracle as orc
    dev = torch.device("cuda")
    b = synth.molecule_batch(60, seed=31, extra_bonds=3.9, eig_dim=6)
    N = int(b["num_nodes"])
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    scalers = "identity amplification attenuation"
    layer, gen = _make_layer(type_net, F_, aggs, scalers, True, avg, o1_weights=type_net != "towers")
    with torch.no_grad():      # running statistics that are not the initial (0, 1)
        for k, v in layer.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(0.3 * torch.randn(v.shape, generator=gen))
            elif k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=gen))
    h = torch.randn(N, F_, generator=gen)
    sd = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(avg), graph_norm=True, batch_norm=True, residual=True,
               towers=5 if type_net == "towers" else 1, divide_input=True, edge_features=False)
    yo, _ = orc.layer_forward(type_net, sd, cfg, b["src"], b["dst"], N, b["eig"], h, None, b["snorm_n"], training=False)
    layer = layer.to(dev).eval()
    before = {k: v.clone() for k, v in layer.state_dict().items()}
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    taken = _count_route(monkeypatch)
    with torch.no_grad():
        y = layer(graph, h.to(dev), None, b["snorm_n"].to(dev))
        assert len(taken) == 1, "the evaluation forward left the route"
        monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", 0)
        ys = layer(graph, h.to(dev), None, b["snorm_n"].to(dev))
    assert len(taken) == 1
    np.testing.assert_allclose(y.cpu().numpy(), yo.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(y.cpu().numpy(), ys.cpu().numpy(), rtol=2e-5, atol=2e-5)
    for k, v in layer.state_dict().items():
        assert torch.equal(v, before[k]), k


@pytest.mark.parametrize("kind,type_net,F_,scalers", [("knn", "simple", 65, "identity"), ("knn", "simple", 70, "identity amplification attenuation")])
def test_evaluation_forward_of_knn_and_sbm_batches_on_the_route(monkeypatch, kind, type_net, F_, scalers):
    """Round 6: CIFAR10-like 8-NN graphs (85-150 nodes, hidden 65) at the json's batch size -- above ops.BLOCK_LAYER_MAX_NODES, and their
    BACKWARD does not fit the LDS (training keeps the streaming kernels; the COMPLEX layer's P | Q rows (170 KB) and PATTERN's SBM graphs,
    ~6 000 edges each, do not fit the forward either) -- run their
    EVALUATION forward (eval() under no_grad: the reference's validation / test loops, train/train_superpixels_graph_classification.py)
    on the route: dgn_block_layer_supported with eval_mode needs the forward plan alone, and every (block, tower) workgroup has a CU
    of its own.  Against the oracle in evaluation mode and the streaming kernels."""
    import dgn_amd
    from dgn_amd import synth
    from oracle import dgn_oracle as orc
    dev = torch.device("cuda")
    b = synth.knn_batch(100, seed=41) if kind == "knn" else synth.sbm_batch(n_graphs=100, seed=41)
    N = int(b["num_nodes"])
    assert N > dgn_amd.ops.BLOCK_LAYER_MAX_NODES
    aggs = "mean dir1-dx dir2-dx"
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    layer, gen = _make_layer(type_net, F_, aggs, scalers, True, avg)
    with torch.no_grad():
        for k, v in layer.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(0.3 * torch.randn(v.shape, generator=gen))
            elif k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=gen))
    h = torch.randn(N, F_, generator=gen)
    sd = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(avg), graph_norm=True, batch_norm=True, residual=True, towers=1,
               divide_input=True, edge_features=False)
    yo, _ = orc.layer_forward(type_net, sd, cfg, b["src"], b["dst"], N, b["eig"], h, None, b["snorm_n"], training=False)
    layer = layer.to(dev).eval()
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    taken = _count_route(monkeypatch)
    with torch.no_grad():
        y = layer(graph, h.to(dev), None, b["snorm_n"].to(dev))
        assert len(taken) == 1, "the evaluation forward of the k-NN / SBM batch did not take the route"
        monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", 0)
        ys = layer(graph, h.to(dev), None, b["snorm_n"].to(dev))
    assert len(taken) == 1
    scale = max(1.0, float(yo.abs().max()))
    np.testing.assert_allclose(y.cpu().numpy(), yo.numpy(), rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(y.cpu().numpy(), ys.cpu().numpy(), rtol=2e-5, atol=2e-5 * scale)
    # training on the same batch keeps the streaming kernels (the backward plan does not fit / the node limit)
    layer.train()
    monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", 8192)
    layer(graph, h.to(dev).requires_grad_(True), None, b["snorm_n"].to(dev)).sum().backward()
    assert len(taken) == 1


@pytest.mark.parametrize("type_net,F_,aggs", [("towers", 70, "mean max min dir1-av dir1-dx"), ("complex", 45, "mean dir1-dx dir1-av"), ("simple", 75, "mean dir1-dx-no-abs")])
def test_direct_parameter_gradients_are_the_autograd_ones(monkeypatch, type_net, F_, aggs):
    """``ops.DIRECT_PARAM_GRADS`` (opt-in): the block route's backward assigns the parameters' ``.grad`` itself instead of returning 33
    gradients through autograd.  Same kernels, so output, d h, every parameter gradient and the running statistics are BITWISE those of the
    default path; a second backward accumulates as autograd does."""
    import dgn_amd
    from dgn_amd import ops, synth
    dev = torch.device("cuda")
    b = synth.molecule_batch(64, seed=11)
    N = int(b["num_nodes"])
    avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
    res = {}
    for direct in (False, True):
        monkeypatch.setattr(ops, "DIRECT_PARAM_GRADS", direct)
        taken = _count_route(monkeypatch)
        layer, gen = _make_layer(type_net, F_, aggs, "identity amplification attenuation", True, avg)
        layer = layer.to(dev).train()
        graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
        graph.batch_num_nodes = b["sizes"].tolist()
        h = torch.randn(N, F_, generator=gen).to(dev).requires_grad_(True)
        ct = torch.randn(N, F_, generator=gen).to(dev)
        snorm = b["snorm_n"].to(dev)
        y = layer(graph, h, None, snorm)
        assert taken, "the graph-block route was not taken"
        y.backward(ct)
        first = [p.grad.clone() for p in layer.parameters()]
        y2 = layer(graph, h, None, snorm)
        y2.backward(ct)                                                        # accumulates into .grad (h.grad too)
        res[direct] = dict(y=y.detach().clone(), first=first, second=[p.grad.clone() for p in layer.parameters()], gh=h.grad.clone(),
                           stats=[v.clone() for k, v in layer.state_dict().items() if "running" in k])
    for key in ("first", "second", "stats"):
        for a, r in zip(res[True][key], res[False][key]):
            assert torch.equal(a, r), key
    assert torch.equal(res[True]["y"], res[False]["y"]) and torch.equal(res[True]["gh"], res[False]["gh"])
    assert all(g is not None and float(g.abs().max()) > 0 for g in res[True]["first"])
