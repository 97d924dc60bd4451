"""The staging-free ("pull") backward of the sweep, dgn_agg_backward_csc: lists without max / min / std / var on graphs with more than
three in-edges per row -- CIFAR10 / MNIST superpixel k-NN graphs and the SBM graphs of PATTERN / CLUSTER in the reference's configs
(configs/superpixels_graph_classification_DGN_CIFAR10.json:23, configs/SBMs_node_clustering_DGN_PATTERN.json:26).  It must give the
gradients of the staged two-phase scatter bit for bit, and both must match the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _graph(seed, N, E):
    """Multigraph with duplicates, a node without in-edges (N - 1), a node without out-edges (N - 2) and a few longer rows."""
    rng = np.random.default_rng(seed)
    dst = rng.integers(0, N - 1, E)
    long_rows = rng.random(E) < 0.3
    dst[long_rows] = rng.integers(0, 3, long_rows.sum())
    src = rng.integers(0, N - 2, E)
    src[src == N - 2] = 0
    return torch.from_numpy(src), torch.from_numpy(dst)


LISTS = [["mean", "dir1-dx", "dir2-dx"],                              # CIFAR10 / PATTERN json
         ["mean", "sum", "dir1-av", "dir2-dx-no-abs"],               # dir-av: the |w| coefficient vectors
         ["sum"],                                                      # no weight channel at all
         ["mean", "dir1-dx-balanced", "dir2-0.1", "dir3-neg-0.1"]]   # balanced / softmax channels


def _sinks_equal(a, b):
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("F_", [6, 66, 128, 300])
@pytest.mark.parametrize("aggs", LISTS)
@pytest.mark.parametrize("msg_form", ["src", "pair"])
def test_pull_backward_is_bitwise_the_staged_backward(monkeypatch, F_, aggs, msg_form):
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import launch_backward
    N, E = 47, 900
    src, dst = _graph(F_ + len(aggs), N, E)
    gen = torch.Generator().manual_seed(F_)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=torch.randn(N, 4, generator=gen).to(dev))
    assert graph.pull_capable
    plan = dgn_amd.make_plan(aggs, ["identity"])
    xs, xd, xin = (torch.randn(N, F_, generator=gen).to(dev) for _ in range(3))
    w = graph.edge_weights(plan)
    g_out = torch.randn(N, plan.out_width(F_), generator=gen).to(dev)

    def run():
        if msg_form == "src":            # simple layer: message = h[src], h_in = h, one gradient buffer for both
            g = torch.full((N, F_), float("nan"), device=dev)
            launch_backward(graph, plan, 1, 0.9, w, xs, None, None, xs, g_out, g, None, None, g, accumulate=False)
            return [g]
        sinks = [torch.full((N, F_), float("nan"), device=dev) for _ in range(3)]
        launch_backward(graph, plan, 1, 0.9, w, xs, xd, None, xin, g_out, sinks[0], sinks[1], None, sinks[2], accumulate=False)
        return sinks

    pulled = run()
    monkeypatch.setenv("DGN_NO_PULL", "1")
    staged = run()
    monkeypatch.delenv("DGN_NO_PULL")
    for t in pulled:
        assert torch.isfinite(t).all()
    _sinks_equal(pulled, staged)
    _sinks_equal(run(), pulled)                                       # and it is reproducible run to run


def test_pull_path_is_the_one_that_runs(monkeypatch):
    """With the csc view's destination rows absent the library takes the staged path; with them it needs no staging buffer: a workspace
    that only fits the coefficient rows is accepted."""
    dev = _dev()
    import ctypes as C
    import dgn_amd
    from dgn_amd import _lib, ops
    lib = _lib.load()
    N, E, F_ = 64, 64 * 12, 66
    src, dst = _graph(3, N, E)
    gen = torch.Generator().manual_seed(0)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=torch.randn(N, 3, generator=gen).to(dev))
    plan = dgn_amd.make_plan(["mean", "dir1-dx", "dir2-dx"], ["identity"])
    graph.ensure_csc_dst()
    spec = ops._spec_structs(plan, 1, 1.0, 0)[0]
    g = graph.c_graph
    coef_bytes = N * 3 * F_ * 4
    stage_bytes = E * F_ * 4
    need = lib.dgn_agg_backward_workspace_bytes(C.byref(g), C.byref(spec), F_, 1)
    assert need >= stage_bytes > coef_bytes
    xs = torch.randn(N, F_, generator=gen).to(dev)
    w = graph.edge_weights(plan)
    wc = graph.weights_csc(w)
    assert torch.equal(wc, w[:, graph._csc_order.long()])
    g_out = torch.randn(N, plan.out_width(F_), generator=gen).to(dev)
    ref = torch.empty(N, F_, device=dev)
    ops.launch_backward(graph, plan, 1, 1.0, w, xs, None, None, xs, g_out, ref, None, None, ref, accumulate=False)
    # the same call by hand with a workspace of the coefficient rows only
    small = torch.empty(((coef_bytes + 255) // 256) * 256, dtype=torch.uint8, device=dev)
    out = torch.full((N, F_), float("nan"), device=dev)
    grads = _lib.DgnMsgGrad()
    grads.g_src, grads.ld_src, grads.g_in, grads.ld_in, grads.accumulate = out.data_ptr(), F_, out.data_ptr(), F_, 0
    msg = ops._msg_struct(F_, xs, None, None, xs, None)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.dgn_agg_backward_csc(C.byref(g), C.byref(spec), C.byref(msg), w.data_ptr(), w.stride(0), wc.data_ptr(), wc.stride(0),
                                  graph.log_deg.data_ptr(), g_out.data_ptr(), g_out.stride(0), C.byref(grads), small.data_ptr(), small.numel(), st)
    _lib.check(rc, "dgn_agg_backward_csc")
    assert torch.equal(out, ref)


@pytest.mark.parametrize("kind,aggs", [("knn", "mean dir1-dx dir2-dx"), ("sbm", "mean dir1-dx dir2-dx"), ("knn", "mean sum dir1-av dir2-dx-no-abs")])
def test_pull_backward_vs_oracle_on_knn_and_sbm_batches(kind, aggs):
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.ops import directional_aggregate
    from oracle import dgn_oracle as orc
    b = synth.knn_batch(n_graphs=3, seed=5) if kind == "knn" else synth.sbm_batch(n_graphs=2, seed=5, n_lo=30, n_hi=50)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    eig = b["eig"].float()
    F_ = 10
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(N, F_, generator=gen)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    assert graph.pull_capable
    names = aggs.split()
    plan = dgn_amd.make_plan(names, ["identity"])
    hd = h.to(dev).requires_grad_(True)
    y = directional_aggregate(graph, plan, 1.0, x_src=hd, x_in=hd)
    ho = h.clone().requires_grad_(True)
    yo = orc.aggregate_graph(src, dst, N, ho[src], eig, ho, names, ["identity"], torch.tensor(1.0))
    assert torch.allclose(y.cpu(), yo, rtol=1e-5, atol=1e-5)
    ct = torch.randn(yo.shape, generator=gen)
    gd = torch.autograd.grad(y, hd, ct.to(dev))[0].cpu()
    go = torch.autograd.grad(yo, ho, ct)[0]
    scale = float(go.abs().max())
    assert float((gd - go).abs().max()) <= 1e-4 * max(scale, 1.0)


@pytest.mark.parametrize("type_net,F_", [("simple", 65), ("complex", 47)])
def test_dense_layer_with_pull_backward_equals_staged(monkeypatch, type_net, F_):
    """The whole simple / complex layer call (dgn_dense_layer_backward) on a k-NN batch: every gradient identical with and without the
    pull path."""
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    b = synth.knn_batch(n_graphs=40, seed=2)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].float().to(dev))
    assert graph.pull_capable
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, "mean dir1-dx dir2-dx", "identity amplification attenuation", {"log": 2.1}, type_net, True).model.to(dev)
    h = torch.randn(N, F_, device=dev, requires_grad=True)
    snorm = b["snorm_n"].float().to(dev)
    ct = torch.randn(N, F_, device=dev)

    def run():
        layer.zero_grad()
        out = layer(graph, h, None, snorm)
        gh, = torch.autograd.grad(out, h, ct, retain_graph=True)
        out.backward(ct)
        return [out.detach().clone(), gh] + [p.grad.clone() for p in layer.parameters()]

    a = run()
    monkeypatch.setenv("DGN_NO_PULL", "1")
    c = run()
    for x, y in zip(a, c):
        assert torch.equal(x, y)
