"""BASELINE configs that round 1 benched but never put in front of the oracle:

* C5's own aggregator / scaler list (mean max min sum std dir1-dx dir2-dx dir3-dx x identity amplification attenuation,
  scalers applied IN the sweep, three weight channels, F = 128, K = 4) against the oracle at oracle-feasible size, row path
  and hub-slice path, values and gradient;
* C3's coordinate-eig mode (eig = [0, x, y], data/superpixels.py:423-428: column 0 is constant, so dir0 would be all-zero
  deltas; the config uses dir1 / dir2) on directed 8-NN graphs with zero-in-degree nodes, simple and complex layers;
* the FULL C5 graph (10 M nodes / 200 M edges, 3.07e10 output elements, 64-bit offsets) with C5's own list: sampled rows --
  random ones beyond the 2^31-element mark, the first and the last row, hub rows -- against the oracle run on the extracted
  sub-problem, plus the size-independent properties (linearity, sum = deg * mean, hub-slice path == row path on a row range).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C5_AGGS = "mean max min sum std dir1-dx dir2-dx dir3-dx".split()
C5_SCALERS = "identity amplification attenuation".split()


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def _as_good(ours, ref32, ref64, rtol, atol, msg=""):
    """within tolerance of the fp32 oracle, or as close to the fp64 oracle as the fp32 oracle is (x4): std / max / |.| are
    ill-conditioned where variances vanish or two messages tie, in the reference as much as here.  Counted, reported and bounded
    (at most 0.5 % of the entries may need the fp64 clause) by parity_util.check."""
    from parity_util import check
    check(ours, ref32, ref64, msg or "tensor", rtol=rtol, atol=atol)


@pytest.mark.parametrize("hub", [False, True], ids=["row-path", "hub-slices"])
def test_c5_aggregator_list_vs_oracle(hub):
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    from oracle import dgn_oracle as orc
    dev = _dev()
    rng = np.random.default_rng(50 + hub)
    N, E, F_, K = 61, 900, 128, 4
    dst = rng.integers(0, N - 1, E)                       # node N-1: zero in-degree
    long_rows = rng.random(E) < 0.35
    dst[long_rows] = rng.integers(0, 3, long_rows.sum())  # three rows of ~100 slots (two slot batches; hub path when sliced)
    src, dst = torch.from_numpy(rng.integers(0, N, E)), torch.from_numpy(dst)
    gen = torch.Generator().manual_seed(5)
    X, eig = torch.randn(N, F_, generator=gen), torch.randn(N, K, generator=gen)
    kw = dict(hub_threshold=64, hub_chunk=24) if hub else {}
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev), **kw)
    assert (graph.n_hub > 0) == hub
    plan = dgn_amd.make_plan(C5_AGGS, C5_SCALERS)
    assert plan.n_channels == 3 and plan.n_scalers == 3 and len(plan.launches) == 1
    avg = float(torch.log(torch.bincount(dst, minlength=N).float() + 1).mean())
    xd = X.to(dev).requires_grad_(True)
    y = directional_aggregate(graph, plan, avg, x_src=xd, x_in=xd)
    assert tuple(y.shape) == (N, 8 * 3 * F_)
    res = {}
    for dt in (torch.float32, torch.float64):
        xo = X.to(dt).requires_grad_(True)
        yo = orc.aggregate_graph(src, dst, N, xo[src], eig.to(dt), xo, C5_AGGS, C5_SCALERS, torch.tensor(avg, dtype=dt))
        ct = torch.randn(yo.shape, generator=torch.Generator().manual_seed(6))
        res[dt] = (yo, torch.autograd.grad(yo, xo, ct.to(dt))[0], ct)
    _as_good(y, res[torch.float32][0], res[torch.float64][0], 1e-5, 1e-5, "C5 list y")
    (gx,) = torch.autograd.grad(y, xd, res[torch.float32][2].to(dev))
    _as_good(gx, res[torch.float32][1], res[torch.float64][1], 1e-4, 2e-5, "C5 list grad")
    assert float(y[N - 1].detach().abs().max()) == 0.0


@pytest.mark.parametrize("type_net", ["simple", "complex"])
def test_c3_coordinate_eig_layer_vs_oracle(type_net):
    """CIFAR10 config (configs/superpixels_graph_classification_DGN_CIFAR10.json: hidden 65, mean dir1-dx dir2-dx, identity)
    on directed 8-NN superpixel-like graphs with eig = [0, x, y]; output, input gradient, every parameter gradient."""
    import dgn_amd
    from dgn_amd import synth
    from oracle import dgn_oracle as orc
    dev = _dev()
    b = synth.knn_batch(n_graphs=3, seed=7, n_lo=40, n_hi=60)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    eig = b["eig"]
    assert float(eig[:, 0].abs().max()) == 0.0 and eig.shape[1] == 3            # [0, x, y]
    indeg = torch.bincount(dst, minlength=N)
    assert int(indeg.min()) == 0 or int(indeg.max()) > 8                        # in-degree varies (k-NN is not symmetric)
    F_ = 65
    aggs, scalers = "mean dir1-dx dir2-dx", "identity"
    avg = float(torch.log(indeg.float() + 1).mean())
    torch.manual_seed(1)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, aggs, scalers, {"log": torch.tensor(avg)}, type_net, True,
                             edge_features=False, edge_dim=0).model
    gen = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(3.0).add_(0.05 * torch.randn(p.shape, generator=gen))
    h, ct = torch.randn(N, F_, generator=gen), torch.randn(N, F_, generator=gen)

    def oracle(dt):
        sd = {k: (v.detach().to(dt).requires_grad_("running" not in k) if v.dtype.is_floating_point else v.clone())
              for k, v in layer.state_dict().items()}
        names = [k for k, v in sd.items() if v.dtype.is_floating_point and v.requires_grad]
        cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(avg, dtype=dt), graph_norm=True, batch_norm=True,
                   residual=True, towers=1, divide_input=True, edge_features=False)
        hh = h.to(dt).requires_grad_(True)
        y, _ = orc.layer_forward(type_net, sd, cfg, src, dst, N, eig.to(dt), hh, None, b["snorm_n"].to(dt), training=True)
        return y, torch.autograd.grad(y, [hh] + [sd[k] for k in names], ct.to(dt)), names

    y32, g32, names = oracle(torch.float32)
    y64, g64, _ = oracle(torch.float64)
    layer = layer.to(dev).train()
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    hd = h.to(dev).requires_grad_(True)
    y = layer(graph, hd, None, b["snorm_n"].to(dev))
    _as_good(y, y32, y64, 2e-5, 2e-5, "y")
    params = dict(layer.named_parameters())
    gd = torch.autograd.grad(y, [hd] + [params[k] for k in names], ct.to(dev))
    for a, r32, r64, k in zip(gd, g32, g64, ["h"] + names):
        _as_good(a, r32, r64, 1e-4, 2e-5, k)


# ---- the full C5 graph ---------------------------------------------------------------------------------------------

def _extract_rows(indptr, src, rows):
    """(sub_src, sub_dst, node ids) of the sub-problem holding ``rows`` and all their in-edges: nodes = the rows and their
    sources, relabelled; only the chosen rows have in-edges.  All on the CPU."""
    rows = rows.long()
    beg, end = indptr[rows], indptr[rows + 1]
    cnt = end - beg
    pos = torch.repeat_interleave(beg, cnt) + (torch.arange(int(cnt.sum())) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    e_src = src[pos].long()
    e_dst = torch.repeat_interleave(rows, cnt)
    nodes = torch.unique(torch.cat([rows, e_src]))
    relabel = {int(v): i for i, v in enumerate(nodes.tolist())}
    f = lambda t: torch.tensor([relabel[int(v)] for v in t.tolist()], dtype=torch.long)
    return f(e_src), f(e_dst), nodes, f(rows)


@pytest.mark.timeout(1800)
def test_c5_full_graph_sampled_rows_and_properties():
    import dgn_amd
    from dgn_amd import dist as ddist, synth
    from dgn_amd.ops import launch_forward
    from oracle import dgn_oracle as orc
    dev = _dev()
    free, total = torch.cuda.mem_get_info(dev)
    if total < 200 * 2 ** 30:
        pytest.skip("needs the 288 GB of an MI355X (the C5 output alone is 123 GB)")
    N, E_target, F_ = 10_000_000, 200_000_000, 128
    indptr, src, eig = synth.powerlaw_csr(N, E_target, dev, seed=0)
    graph = dgn_amd.DGNGraph.from_csr(indptr, src, eig=eig)
    E = graph.num_edges
    assert graph.n_hub > 0
    plan = dgn_amd.make_plan(C5_AGGS, C5_SCALERS)
    W = plan.out_width(F_)
    assert N * W > 2 ** 31                                                      # 64-bit output offsets are exercised
    avg = float(graph.log_deg.mean().item())
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(N, F_, device=dev, generator=gen)
    Y = torch.randn(N, F_, device=dev, generator=gen)
    w = graph.edge_weights(plan)
    deg = graph.in_degree
    # rows to look at: random rows, the first / last rows, rows past the 2^31-element mark, the smallest and the largest
    # hub row (sliced path), and a contiguous range for the hub-vs-row-path comparison
    rs = torch.Generator().manual_seed(1)
    first_big = (2 ** 31) // W + 1
    hub_rows = torch.nonzero(deg > graph.hub_threshold).flatten()
    hub_sorted = hub_rows[torch.argsort(deg[hub_rows])]
    small_hubs = hub_sorted[:3].cpu()
    sample = torch.unique(torch.cat([torch.randint(0, N, (300,), generator=rs), torch.tensor([0, 1, N - 2, N - 1, first_big, first_big + 1]),
                                     torch.randint(first_big, N, (100,), generator=rs), small_hubs]))
    sample_d = sample.to(dev)
    out = torch.empty(N, W, device=dev)

    def fwd(Z):
        launch_forward(graph, plan, 1, avg, w, Z, None, None, Z, out)
        return out

    oX = fwd(X)
    got = oX[sample_d].cpu()                                                    # [n_sample, W]
    blk = lambda t, s, a: t[:, (s * 8 + a) * F_:(s * 8 + a + 1) * F_]
    # sum = deg * mean on EVERY row (scaler block 0 = identity), streamed in row chunks
    worst = 0.0
    for r0 in range(0, N, 1_000_000):
        r1 = min(N, r0 + 1_000_000)
        mean, s = oX[r0:r1, 0:F_], oX[r0:r1, 3 * F_:4 * F_]
        d = deg[r0:r1].float().unsqueeze(1)
        worst = max(worst, float(((s - mean * d).abs() / (1 + s.abs())).max()))
        assert bool((oX[r0:r1, F_:2 * F_] >= oX[r0:r1, 2 * F_:3 * F_]).all())   # max >= min
    assert worst < 1e-4, worst
    # hub-slice path == row path: a contiguous destination range around the largest hub, run as its own shard with slicing off
    big = int(hub_sorted[-1])
    r0, r1 = max(0, big - 2000), min(N, big + 2000)
    ref_rows = oX[r0:r1].clone()
    shard = ddist.shard_rows(indptr, src, r0, r1, hub_threshold=2 ** 30)
    assert shard.n_hub == 0 and int(shard.in_degree.max()) == int(deg[big])
    w_sh = shard.edge_weights(plan, eig)
    out_sh = torch.empty(r1 - r0, W, device=dev)
    launch_forward(shard, plan, 1, avg, w_sh, X, None, None, X[r0:r1], out_sh)
    # slices merge in slot order: only the association of the fp32 sums differs.  One wave adding the d ~ 1e6 messages of
    # the largest hub one after the other carries ~ sqrt(d) u relative rounding noise (the sliced path is the MORE accurate
    # one), so that row gets a bound that grows with sqrt(d); every other row of the range must agree to 2e-5.
    d_loc = deg[r0:r1].float()
    per_row = (out_sh - ref_rows).abs().amax(1) / ref_rows.abs().amax(1).clamp_min(1.0)
    lim = torch.clamp(2.0 ** -23 * torch.sqrt(d_loc) * 4.0, min=2e-5)
    assert bool((per_row <= lim).all()), (float((per_row / lim).max()), int(torch.argmax(per_row / lim)) + r0)
    del out_sh, ref_rows, shard, w_sh
    # linearity of the linear aggregators (mean, sum) under every scaler, on the sampled rows
    gotY = fwd(Y)[sample_d].cpu()
    gotL = fwd(2.0 * X - 0.5 * Y)[sample_d].cpu()
    for s in range(3):
        for a in (0, 3):
            ref = 2.0 * blk(got, s, a) - 0.5 * blk(gotY, s, a)
            err = float((blk(gotL, s, a) - ref).abs().max() / ref.abs().max().clamp_min(1.0))
            assert err < 2e-5, (s, a, err)
    del out, oX
    torch.cuda.empty_cache()
    # the sampled rows against the oracle on the extracted sub-problem (rows + their sources, relabelled)
    sub_src, sub_dst, nodes, sub_rows = _extract_rows(indptr.cpu(), src.cpu(), sample)
    Xs, eigs = X[nodes.to(dev)].cpu(), eig[nodes.to(dev)].cpu()
    n_sub = nodes.numel()
    y32 = orc.aggregate_graph(sub_src, sub_dst, n_sub, Xs[sub_src], eigs, Xs, C5_AGGS, C5_SCALERS, torch.tensor(avg))[sub_rows]
    y64 = orc.aggregate_graph(sub_src, sub_dst, n_sub, Xs.double()[sub_src], eigs.double(), Xs.double(), C5_AGGS, C5_SCALERS,
                              torch.tensor(avg, dtype=torch.float64))[sub_rows]
    assert int((deg[sample_d] > graph.hub_threshold).sum()) >= 3               # hub rows are among the checked rows
    _as_good(got, y32, y64, 2e-5, 2e-5, "sampled C5 rows (incl. row N-1 at element offset %d)" % ((N - 1) * W))


@pytest.mark.timeout(1800)
def test_c5_full_graph_layer_forward_sampled_rows_vs_oracle():
    """The FULL C5 graph through a whole ``DGNLayerSimple.forward`` (nets/dgn_layer.py:178-202, evaluation mode, C5's list, hidden 128):
    sweep with the scalers folded -> one product per in-degree class on the rows below in-degree 32 + the folded product on the 6 % hub
    rows (``ops.dc_posttrans_split``) -> BatchNorm on the running statistics -> ReLU -> residual.  Sampled rows -- low in-degrees, rows
    past in-degree 31 (the folded product), rows past the sweep's hub threshold (sliced path), the first and last rows -- against the
    oracle's layer on the extracted sub-problem."""
    import dgn_amd
    from dgn_amd import ops, synth
    from oracle import dgn_oracle as orc
    dev = _dev()
    free, total = torch.cuda.mem_get_info(dev)
    if total < 200 * 2 ** 30:
        pytest.skip("needs the memory of an MI355X (the aggregate block alone is 41 GB)")
    N, E_target, F_ = 10_000_000, 200_000_000, 128
    indptr, src, eig = synth.powerlaw_csr(N, E_target, dev, seed=0)
    graph = dgn_amd.DGNGraph.from_csr(indptr, src, eig=eig)
    deg = graph.in_degree
    avg = float(graph.log_deg.mean().item())
    aggs, scalers = " ".join(C5_AGGS), " ".join(C5_SCALERS)
    torch.manual_seed(3)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, False, True, aggs, scalers, {"log": torch.tensor(avg)}, "simple", True, towers=1, edge_features=False,
                             edge_dim=0).model
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        layer.batchnorm_h.running_mean.copy_(torch.randn(F_, generator=gen) * 0.1)
        layer.batchnorm_h.running_var.copy_(torch.rand(F_, generator=gen) + 0.5)
        for p in layer.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).eval()
    X = torch.randn(N, F_, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    calls = []
    real = ops.dc_posttrans_split
    ops.dc_posttrans_split = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            out = layer(graph, X, None, None)
    finally:
        ops.dc_posttrans_split = real
    assert len(calls) == 1, "the split degree-class route was not taken"
    assert out.shape == (N, F_) and bool(torch.isfinite(out).all())
    rs = torch.Generator().manual_seed(6)
    sweep_hubs = torch.nonzero(deg > graph.hub_threshold).flatten()
    small_hubs = sweep_hubs[torch.argsort(deg[sweep_hubs])][:3].cpu()
    mid = torch.nonzero((deg >= 32) & (deg < 300)).flatten()
    mid = mid[torch.randint(0, mid.numel(), (40,), generator=rs).to(dev)].cpu()
    sample = torch.unique(torch.cat([torch.randint(0, N, (300,), generator=rs), torch.tensor([0, 1, N - 2, N - 1]), small_hubs, mid]))
    sample_d = sample.to(dev)
    d_s = deg[sample_d]
    assert int((d_s < 32).sum()) >= 200 and int((d_s >= 32).sum()) >= 40 and int((d_s > graph.hub_threshold).sum()) >= 3
    got = out[sample_d].cpu()
    del out
    torch.cuda.empty_cache()
    sub_src, sub_dst, nodes, sub_rows = _extract_rows(indptr.cpu(), src.cpu(), sample)
    Xs, eigs = X[nodes.to(dev)].cpu(), eig[nodes.to(dev)].cpu()
    n_sub = nodes.numel()
    cfg = dict(aggregators=aggs, scalers=scalers, graph_norm=False, batch_norm=True, residual=True, towers=1, divide_input=False, edge_features=False)
    res = {}
    for dt in (torch.float32, torch.float64):
        sdt = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
        res[dt] = orc.layer_forward("simple", sdt, dict(cfg, avg_log=torch.tensor(avg, dtype=dt)), sub_src, sub_dst, n_sub, eigs.to(dt), Xs.to(dt), None,
                                    None, training=False)[0][sub_rows]
    _as_good(got, res[torch.float32], res[torch.float64], 2e-5, 2e-5, "sampled rows of the C5 layer forward (split degree-class route)")
