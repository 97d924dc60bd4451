"""GPU parity: the HIP path (through the C ABI) against the golden fixtures produced by the
imported reference and against the CPU oracle on seeded random graphs.  fp32; tolerances are
stated per comparison (north star: 1e-5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = torch.from_numpy
RTOL, ATOL = 1e-5, 1e-5


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _close(a, b, rtol=RTOL, atol=ATOL, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def _as_good(ours, ref32, ref64, rtol=RTOL, atol=ATOL, msg=""):
    """Parity criterion for ill-conditioned outputs (std gradients through near-zero variances amplify
    fp32 rounding by 1/(2*sqrt(1e-8))): pass if within tolerance of the reference's fp32 result, OR if
    our max error against an fp64 evaluation is no worse than a small multiple of the reference's own."""
    ours = ours.detach().cpu().double().numpy() if torch.is_tensor(ours) else np.asarray(ours, dtype=np.float64)
    ref32 = ref32.detach().cpu().double().numpy() if torch.is_tensor(ref32) else np.asarray(ref32, dtype=np.float64)
    ref64 = ref64.detach().cpu().double().numpy() if torch.is_tensor(ref64) else np.asarray(ref64, dtype=np.float64)
    if np.allclose(ours, ref32, rtol=rtol, atol=atol):
        return
    scale = max(1.0, float(np.abs(ref64).max()))
    err_ours = float(np.abs(ours - ref64).max())
    err_ref = float(np.abs(ref32 - ref64).max())
    assert err_ours <= atol * scale + 4.0 * err_ref, f"{msg}: max err vs fp64 {err_ours:.3e}, reference's own {err_ref:.3e}"


def _mailbox_graph(n, D, dev):
    import dgn_amd
    indptr = torch.arange(0, (n + 1) * D, D, dtype=torch.int64, device=dev)
    src = torch.zeros(n * D, dtype=torch.int64, device=dev)
    return dgn_amd.DGNGraph.from_csr(indptr, src)


def _run_mailbox(name, h, es, ed, x, ct, dev):
    """AGGREGATORS[name](h[n,D,F], eig_s, eig_d, h_in) through the kernels: slot-level eig input."""
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    n, D, F_ = h.shape
    g = _mailbox_graph(n, D, dev)
    plan = dgn_amd.make_plan([name], ["identity"])
    w = None
    if plan.n_channels:
        w = dgn_amd.compute_edge_weights(g, plan.channels, eig_s_edge=es.reshape(n * D, -1).contiguous().to(dev),
                                         eig_d_edge=ed.reshape(n * D, -1).contiguous().to(dev))
    m = h.reshape(n * D, F_).contiguous().to(dev).requires_grad_(True)
    xx = x.to(dev).clone().requires_grad_(True)
    y = directional_aggregate(g, plan, 1.0, m_edge=m, x_in=xx, weights=w)
    gm, gx = torch.autograd.grad(y, [m, xx], ct.to(dev), allow_unused=True)
    return y, gm.reshape(n, D, F_), (gx if gx is not None else torch.zeros_like(xx))


@pytest.mark.parametrize("fixture", ["g1_aggregators", "g5_edge_cases"])
def test_mailbox_aggregators_vs_reference(golden, fixture):
    dev = _dev()
    g = golden(fixture)
    names = g["names"].tolist()
    cases = [f"c{i}" for i in range(int(g["n_cases"]))] if fixture == "g1_aggregators" else g["cases"].tolist()
    for c in cases:
        h, es, ed = T(g[f"{c}/h"]), T(g[f"{c}/eig_s"]), T(g[f"{c}/eig_d"])
        x, ct = T(g[f"{c}/h_in"]), T(g[f"{c}/cot"])
        # the reference's own fp32 noise on these cases: sums of squares of 1000-scaled features
        loose = c == "const_msgs"
        for name in names:
            y, gh, gx = _run_mailbox(name, h, es, ed, x, ct, dev)
            if loose and name in ("std", "var"):
                # catastrophic cancellation (mean(m^2) - mean(m)^2 of five IDENTICAL messages at |m| ~ 1e3): the fp32
                # result is rounding noise in the reference too, so the check is anchored on an fp64 evaluation
                # (aggregators.py:20-28 in double) with the standard forward-error bound of the expression,
                # |fl(var) - var| <= (d + 3) u (mean(m^2) + mean(m)^2), u = 2^-24, element by element; std = sqrt(var + eps)
                # inherits sqrt(bound).  Gradients: as close to fp64 as the reference's own fp32 result (x4).
                h64 = h.double().requires_grad_(True)
                d = h.shape[1]
                m1, m2 = h64.mean(1), (h64 * h64).mean(1)
                var64 = torch.relu(m2 - m1 * m1)
                y64 = var64 if name == "var" else torch.sqrt(var64 + 1e-8)
                bound = (d + 3) * 2.0 ** -24 * (m2 + m1 * m1).detach()
                err = (y.detach().cpu().double() - y64.detach()).abs()
                lim = bound if name == "var" else torch.sqrt(bound + 1e-8)
                assert bool((err <= lim + 1e-6).all()), f"{c} {name}: {float((err - lim).max()):.3e} over the fp32 error bound"
                # Gradients: exactly 0 in exact arithmetic (identical messages).  In fp32 the relu gate [rawvar > 0] opens on
                # rounding noise, and then  d var / d m_j = ct (2/d) (m_j - fl(mean)),  |m_j - fl(mean)| <= (d + 1) u |m|;
                # for std that is divided by 2 std with std >= sqrt(u) |m| whenever the gate is open (one quantum of m^2).
                (gh64,) = torch.autograd.grad(y64, h64, ct.double())
                assert float(gh64.abs().max()) <= 1e-9
                u, cta, ma = 2.0 ** -24, ct.double().abs().unsqueeze(1), h.double().abs()
                lim_var = cta * (2.0 / d) * (d + 1) * u * ma + 1e-6
                lim_g = lim_var if name == "var" else cta * 2.0 * (d + 1) / d * 2.0 ** -12 + 1e-6
                errg = gh.detach().cpu().double().abs()
                assert bool((errg <= lim_g).all()), f"{c} {name} gh: {float((errg - lim_g).max()):.3e} over the bound"
                assert float(gx.abs().max()) == 0.0
                continue
            _close(y, g[f"{c}/{name}/y"], msg=f"{c} {name} y")
            _close(gh, g[f"{c}/{name}/gh"], msg=f"{c} {name} gh")
            _close(gx, g[f"{c}/{name}/gx"], msg=f"{c} {name} gx")


def test_registry_function_api(golden):
    """AGGREGATORS[name](h, eig_s, eig_d, h_in) / SCALERS[name](h, D, avg_d): the reference's function API
    (nets/aggregators.py:74-93, nets/scalers.py:21) served by the kernels."""
    dev = _dev()
    import dgn_amd
    g = golden("g1_aggregators")
    c = "c4"
    h, es, ed, x = (T(g[f"{c}/{k}"]).to(dev) for k in ("h", "eig_s", "eig_d", "h_in"))
    for name in g["names"].tolist():
        _close(dgn_amd.AGGREGATORS[name](h, es, ed, x), g[f"{c}/{name}/y"], msg=name)
    g2 = golden("g2_scalers")
    hh = T(g2["h"]).to(dev)
    for name in dgn_amd.SCALER_NAMES:
        got = dgn_amd.SCALERS[name](hh, D=7, avg_d={"log": torch.tensor(1.3, device=dev)})
        _close(got, g2[f"a1/D7/{name}"], 1e-6, 1e-7, msg=name)


def test_graph_level_concat_order(golden):
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    g = golden("g3_reduce")
    graph = dgn_amd.DGNGraph(T(g["src"]).to(dev), T(g["dst"]).to(dev), int(g["N"]), eig=T(g["eig"]).to(dev))
    aggs = str(g["aggregators"]).split()
    for tag in ("id", "amp_only", "three", "att_amp"):
        scalers = str(g[f"{tag}/scalers"]).split()
        plan = dgn_amd.make_plan(aggs, scalers)
        h = T(g["h"]).to(dev).requires_grad_(True)
        y = directional_aggregate(graph, plan, 0.8, x_src=h, x_in=h)
        assert tuple(y.shape) == g[f"{tag}/y"].shape
        _close(y, g[f"{tag}/y"], msg=tag)
        gh = torch.autograd.grad(y, h, T(g[f"{tag}/cot"]).to(dev))[0]
        _close(gh, g[f"{tag}/gh"], msg=tag + " gh")


def _build_layer_from_fixture(g, name, dev):
    import dgn_amd
    meta = g[f"{name}/meta"].tolist()
    type_net, din, dout, aggs, scalers, avg = meta[0], int(meta[1]), int(meta[2]), meta[3], meta[4], float(meta[5])
    layer = dgn_amd.DGNLayer(in_dim=din, out_dim=dout, dropout=0.0, graph_norm=True, batch_norm=True, aggregators=aggs,
                             scalers=scalers, avg_d={"log": torch.tensor(avg)}, type_net=type_net, residual=True,
                             towers=int(meta[6]), divide_input=bool(int(meta[7])), edge_features=bool(int(meta[8])),
                             edge_dim=int(meta[9]), pretrans_layers=int(meta[10]), posttrans_layers=int(meta[11])).model
    sd = {k[len(name) + 5:]: T(g[k]) for k in g.files if k.startswith(f"{name}/sd::")}
    missing, unexpected = layer.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return layer.to(dev), bool(int(meta[12]))


@pytest.mark.parametrize("linear_min_rows", [None, 0], ids=["default-routes", "streaming-linear"])
def test_layers_vs_reference(golden, monkeypatch, linear_min_rows):
    """(second run: every dense product whose widths allow it goes through the streaming dgn_linear_* kernels, which the
    small fixture graphs would not reach on their own)"""
    dev = _dev()
    import dgn_amd
    from oracle import dgn_oracle as orc
    if linear_min_rows is not None:
        monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", linear_min_rows)
    g = golden("g4_layers")
    src, dst, N = T(g["src"]).to(dev), T(g["dst"]).to(dev), int(g["N"])
    snorm = T(g["snorm_n"]).to(dev)
    for name in g["cases"].tolist():
        # fp64 evaluation of the same layer by the oracle (for ill-conditioned gradients)
        meta = g[f"{name}/meta"].tolist()
        cfg = dict(aggregators=meta[3], scalers=meta[4], avg_log=torch.tensor(float(meta[5]), dtype=torch.float64),
                   towers=int(meta[6]), divide_input=bool(int(meta[7])), edge_features=bool(int(meta[8])),
                   graph_norm=True, batch_norm=True, residual=True)
        sd64 = {k[len(name) + 5:]: T(g[k]).double() for k in g.files if k.startswith(f"{name}/sd::")}
        pn = [k[len(name) + 5:] for k in g.files if k.startswith(f"{name}/gp::")]
        for k in pn:
            sd64[k].requires_grad_(True)
        h64 = T(g[f"{name}/h"]).double().requires_grad_(True)
        e64 = T(g[f"{name}/e"]).double().requires_grad_(True)
        y64, _ = orc.layer_forward(meta[0], sd64, cfg, src.cpu(), dst.cpu(), N, T(g[f"{name}/eig"]).double(), h64, e64,
                                   snorm.cpu().double(), training=bool(int(meta[12])))
        g64 = torch.autograd.grad(y64, [h64, e64] + [sd64[k] for k in pn], T(g[f"{name}/cot"]).double(), allow_unused=True)
        layer, train = _build_layer_from_fixture(g, name, dev)
        layer.train(train)
        graph = dgn_amd.DGNGraph(src, dst, N, eig=T(g[f"{name}/eig"]).to(dev))
        h = T(g[f"{name}/h"]).to(dev).requires_grad_(True)
        e = T(g[f"{name}/e"]).to(dev).requires_grad_(True)
        y = layer(graph, h, e, snorm)
        _close(y, g[f"{name}/y"], 2e-5, 2e-5, msg=f"{name} y")
        params = dict(layer.named_parameters())
        grads = torch.autograd.grad(y, [h, e] + [params[k] for k in pn], T(g[f"{name}/cot"]).to(dev), allow_unused=True)
        _as_good(grads[0], g[f"{name}/gh"], g64[0], 1e-4, 2e-5, msg=f"{name} gh")
        if grads[1] is not None:
            _as_good(grads[1], g[f"{name}/ge"], g64[1], 1e-4, 2e-5, msg=f"{name} ge")
        for k, gr, gr64 in zip(pn, grads[2:], g64[2:]):
            ref = g[f"{name}/gp::{k}"]
            if gr is None:
                gr = np.zeros_like(ref)
            _as_good(gr, ref, gr64 if gr64 is not None else np.zeros_like(ref), 1e-4, 5e-5, msg=f"{name} grad {k}")
        for k, v in layer.state_dict().items():
            if "running" in k:
                _close(v, g[f"{name}/after::{k}"], 1e-5, 1e-6, msg=f"{name} {k}")


@pytest.mark.parametrize("case", [(7, 7, "mean max dir1-dx std", "identity amplification attenuation", 1),
                                  (7, 5, "mean dir1-dx dir2-dx", "identity", 1),
                                  (9, 9, "sum min dir1-av", "identity attenuation", 2)])
def test_simple_layer_odd_width_vs_oracle(golden, case):
    """odd hidden sizes (ZINC simple 75, CIFAR10 65) run the sweep on a zero-padded even width: values and every
    gradient must equal the unpadded reference computation"""
    dev = _dev()
    import dgn_amd
    from oracle import dgn_oracle as orc
    din, dout, aggs, scalers, post = case
    g = golden("g4_layers")
    src, dst, N = T(g["src"]), T(g["dst"]), int(g["N"])
    gen = torch.Generator().manual_seed(din * 10 + dout)
    h, eig, snorm = torch.randn(N, din, generator=gen), torch.randn(N, 4, generator=gen), torch.rand(N, 1, generator=gen) + 0.5
    torch.manual_seed(3)
    layer = dgn_amd.DGNLayer(din, dout, 0.0, True, True, aggs, scalers, {"log": torch.tensor(1.1)}, "simple", True,
                             posttrans_layers=post).model
    with torch.no_grad():
        for q in layer.parameters():
            q.mul_(4.0).add_(0.05 * torch.randn(q.shape, generator=gen))
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k)
          for k, v in layer.state_dict().items()}
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(1.1), graph_norm=True, batch_norm=True, residual=True,
               towers=1, divide_input=True, edge_features=False)
    ho = h.clone().requires_grad_(True)
    yo, _ = orc.layer_forward("simple", sd, cfg, src, dst, N, eig, ho, None, snorm, training=True)
    ct = torch.randn(yo.shape, generator=gen)
    names = [k for k, v in sd.items() if v.requires_grad]
    go = torch.autograd.grad(yo, [ho] + [sd[k] for k in names], ct)
    layer = layer.to(dev).train(True)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    hd = h.to(dev).requires_grad_(True)
    y = layer(graph, hd, None, snorm.to(dev))
    _close(y, yo, 2e-5, 2e-5)
    params = dict(layer.named_parameters())
    gd = torch.autograd.grad(y, [hd] + [params[k] for k in names], ct.to(dev))
    for a, r, k in zip(gd, go, ["h"] + names):
        scale = max(1.0, float(r.abs().max()))
        _close(a, r, 1e-4, 2e-5 * scale, msg=k)


@pytest.mark.parametrize("F_", [300, 131, 6])
def test_short_row_kernels_wide_and_odd_features_vs_oracle(F_):
    """molecule batch (4-rows-per-wave forward, single-batch backward) with several feature tiles (F = 300: two
    16-byte-lane tiles; F = 131: three 4-byte-lane tiles) against the oracle, values and gradients"""
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.ops import directional_aggregate
    from oracle import dgn_oracle as orc
    b = synth.molecule_batch(24, seed=F_, laplacian_eig=False)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    gen = torch.Generator().manual_seed(F_)
    h, x, eig = torch.randn(N, F_, generator=gen), torch.randn(N, F_, generator=gen), torch.randn(N, 3, generator=gen)
    aggs, scalers = ["mean", "max", "std", "dir1-dx", "dir2-av"], ["identity", "amplification", "attenuation"]
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    assert graph.num_edges <= 3 * N                      # the host picks the short-row kernels for this graph
    plan = dgn_amd.make_plan(aggs, scalers)
    hd, xd = h.to(dev).requires_grad_(True), x.to(dev).requires_grad_(True)
    y = directional_aggregate(graph, plan, 1.3, x_src=hd, x_in=xd)
    ho, xo = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yo = orc.aggregate_graph(src, dst, N, ho[src], eig, xo, aggs, scalers, torch.tensor(1.3))
    ho64, xo64 = h.double().requires_grad_(True), x.double().requires_grad_(True)
    y64 = orc.aggregate_graph(src, dst, N, ho64[src], eig.double(), xo64, aggs, scalers, torch.tensor(1.3, dtype=torch.float64))
    _as_good(y, yo, y64, 2e-5, 2e-5)      # (std of two nearly equal messages: sqrt(1e-8 + fp32 cancellation noise), also in the reference)
    ct = torch.randn(yo.shape, generator=gen)
    gd = torch.autograd.grad(y, [hd, xd], ct.to(dev))
    go = torch.autograd.grad(yo, [ho, xo], ct)
    g64 = torch.autograd.grad(y64, [ho64, xo64], ct.double())
    for a, r, r64 in zip(gd, go, g64):
        _as_good(a, r, r64, 1e-4, 2e-5)


def _random_graph(seed, N, E, zero_in=True):
    rng = np.random.default_rng(seed)
    dst = rng.integers(0, N - (1 if zero_in else 0), E)      # last node never a destination
    hub = rng.random(E) < 0.3                                   # a few long rows
    dst[hub] = rng.integers(0, 3, hub.sum())
    src = rng.integers(0, N, E)
    return torch.from_numpy(src), torch.from_numpy(dst)


ALL_AGGS = ["mean", "sum", "max", "min", "std", "var", "dir1-av", "dir2-dx", "dir3-dx-no-abs", "dir1-dx-balanced",
            "dir2-0.1", "dir3-neg-0.1", "dir1-dx", "dir2-av"]


@pytest.mark.parametrize("F_,hub", [(5, False), (70, False), (75, True), (128, True), (200, True), (260, False)])
def test_random_graph_vs_oracle(F_, hub):
    """Seeded random multigraph (duplicates, zero in-degree node, long rows) against the oracle;
    ``hub`` forces the sliced hub-row path with tiny thresholds."""
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    from oracle import dgn_oracle as orc
    N, E, K = 41, 600, 4
    src, dst = _random_graph(F_, N, E)
    gen = torch.Generator().manual_seed(F_)
    h = torch.randn(N, F_, generator=gen)
    eig = torch.randn(N, K, generator=gen)
    scalers = ["identity", "amplification", "attenuation"]
    kw = dict(hub_threshold=16, hub_chunk=7) if hub else {}
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev), **kw)
    assert (graph.n_hub > 0) == hub
    plan = dgn_amd.make_plan(ALL_AGGS, scalers)
    assert len(plan.launches) > 1        # > 4 weight channels: split in several launches
    hd = h.to(dev).requires_grad_(True)
    y = directional_aggregate(graph, plan, 1.2, x_src=hd, x_in=hd)
    ho = h.clone().requires_grad_(True)
    yo = orc.aggregate_graph(src, dst, N, ho[src], eig, ho, ALL_AGGS, scalers, torch.tensor(1.2))
    _close(y, yo, 2e-5, 2e-5)
    ct = torch.randn(yo.shape, generator=gen)
    _close(torch.autograd.grad(y, hd, ct.to(dev))[0], torch.autograd.grad(yo, ho, ct)[0], 1e-4, 1e-4)
    assert float(y[N - 1].detach().abs().max()) == 0.0      # zero in-degree row -> zeros


@pytest.mark.parametrize("F_", [6, 7])      # even: two-phase scatter, odd: atomic scatter
@pytest.mark.parametrize("hub", [False, True])
def test_backward_define_vs_accumulate_mode(F_, hub):
    """DgnMsgGrad.accumulate: 0 defines the sinks (garbage in, gradient out), 1 adds to what is there."""
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import launch_backward
    N, E = 41, 600
    src, dst = _random_graph(F_ + 100, N, E)
    gen = torch.Generator().manual_seed(F_)
    kw = dict(hub_threshold=16, hub_chunk=7) if hub else {}
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=torch.randn(N, 3, generator=gen).to(dev), **kw)
    plan = dgn_amd.make_plan(["mean", "max", "std", "dir1-dx", "dir2-av"], ["identity", "amplification"])
    xs, xd, xin = (torch.randn(N, F_, generator=gen).to(dev) for _ in range(3))
    w = graph.edge_weights(plan)
    g_out = torch.randn(N, plan.out_width(F_), generator=gen).to(dev)

    def run(accumulate, fill):
        sinks = [torch.full((N, F_), fill, device=dev) for _ in range(3)]
        launch_backward(graph, plan, 1, 0.9, w, xs, xd, None, xin, g_out, sinks[0], sinks[1], None, sinks[2], accumulate=accumulate)
        return sinks

    fresh = run(False, float("nan"))                 # uninitialised sinks are fine in define mode
    added = run(True, 1.5)
    for a, b in zip(fresh, added):
        assert torch.isfinite(a).all()
        _close(b - 1.5, a, 1e-5, 1e-5)
    again = run(False, 7.0)                          # and define mode does not depend on the previous content
    for a, b in zip(fresh, again):                   # (odd F: atomic scatter, the add order varies run to run)
        _close(b, a, 1e-5, 1e-5)


def test_three_term_message_vs_oracle():
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    from oracle import dgn_oracle as orc
    N, E, K, F_ = 30, 200, 4, 12
    src, dst = _random_graph(3, N, E)
    gen = torch.Generator().manual_seed(9)
    P, Q, x = (torch.randn(N, F_, generator=gen) for _ in range(3))
    R = torch.randn(E, F_, generator=gen)
    eig = torch.randn(N, K, generator=gen)
    aggs, scalers = ["mean", "max", "min", "std", "dir1-av", "dir1-dx"], ["identity", "attenuation"]
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev), hub_threshold=20, hub_chunk=8)
    plan = dgn_amd.make_plan(aggs, scalers)
    leaves = [t.to(dev).requires_grad_(True) for t in (P, Q, R, x)]
    y = directional_aggregate(graph, plan, 0.9, x_src=leaves[0], x_dst=leaves[1], m_edge=graph.to_slot_order(leaves[2]),
                              x_in=leaves[3])
    lo = [t.clone().requires_grad_(True) for t in (P, Q, R, x)]
    yo = orc.aggregate_graph(src, dst, N, lo[0][src] + lo[1][dst] + lo[2], eig, lo[3], aggs, scalers, torch.tensor(0.9))
    _close(y, yo, 2e-5, 2e-5)
    ct = torch.randn(yo.shape, generator=gen)
    for a, b in zip(torch.autograd.grad(y, leaves, ct.to(dev)), torch.autograd.grad(yo, lo, ct)):
        _close(a, b, 1e-4, 1e-4)


@pytest.mark.parametrize("case", ["molecules", "molecules_towers", "long_rows", "hub", "split_plan"])
def test_edge_type_table_equals_gathered_rows_and_oracle(case):
    """m_edge as a [K, F] table + per-slot types (DgnMsg.edge_type) against (1) the same call with the gathered [E, F] rows --
    forward and the node gradients bit-equal, the table's gradient = the rows' gradient summed by type -- and (2) the oracle."""
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.ops import directional_aggregate
    from oracle import dgn_oracle as orc
    gen = torch.Generator().manual_seed(11)
    K, n_towers, kw = 5, 1, {}
    aggs, scalers = ["mean", "max", "std", "dir1-dx", "dir1-av"], ["identity", "amplification"]
    if case.startswith("molecules"):
        b = synth.molecule_batch(60, seed=3, laplacian_eig=False)
        src, dst, N, F_ = b["src"], b["dst"], int(b["num_nodes"]), 70
        eig = b["eig"]
        if case == "molecules_towers":
            n_towers, scalers = 5, ["identity"]
    else:
        N, E, F_ = 41, 600, 12
        src, dst = _random_graph(5, N, E)
        eig = torch.randn(N, 4, generator=gen)
        if case == "hub":
            kw = dict(hub_threshold=16, hub_chunk=7)
        if case == "split_plan":
            aggs = ALL_AGGS
    E = src.numel()
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev), **kw)
    assert (graph.n_hub > 0) == (case == "hub")
    plan = dgn_amd.make_plan(aggs, scalers)
    P, Q, x = (torch.randn(N, F_, generator=gen) for _ in range(3))
    table = torch.randn(K, F_, generator=gen)
    types = torch.randint(0, K, (E,), generator=gen)
    types_slot = graph.to_slot_order(types.to(dev)).to(torch.int32).contiguous()
    tower_major = n_towers > 1

    def run(table_mode):
        leaves = [t.to(dev).requires_grad_(True) for t in (P, Q, table, x)]
        pq = torch.cat([leaves[0], leaves[1]], dim=1)
        if table_mode:
            y = directional_aggregate(graph, plan, 0.9, x_pair=pq, m_edge=leaves[2], x_in=leaves[3], edge_type=types_slot,
                                      n_towers=n_towers, tower_major=tower_major)
        else:
            rows = leaves[2].index_select(0, types_slot.long())                       # [E, F] in slot order
            y = directional_aggregate(graph, plan, 0.9, x_pair=pq, m_edge=rows, x_in=leaves[3], n_towers=n_towers, tower_major=tower_major)
        ct = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dev)
        return y.detach(), torch.autograd.grad(y, leaves, ct), ct

    y_t, g_t, ct = run(True)
    y_d, g_d, _ = run(False)
    assert torch.equal(y_t, y_d)
    for i in (0, 1, 3):
        if case == "hub":                                   # (hub slices add their row gradients with atomics: order varies run to run)
            _close(g_t[i], g_d[i], 1e-5, 1e-5 * float(g_d[i].abs().max()))
        else:
            assert torch.equal(g_t[i], g_d[i])
    _close(g_t[2], g_d[2], 1e-4, 1e-4 * float(g_d[2].abs().max()))
    if n_towers == 1:
        refs = []
        for dt in (torch.float32, torch.float64):
            lo = [t.clone().to(dt).requires_grad_(True) for t in (P, Q, table, x)]
            yo = orc.aggregate_graph(src, dst, N, lo[0][src] + lo[1][dst] + lo[2][types], eig.to(dt), lo[3], aggs, scalers, torch.tensor(0.9, dtype=dt))
            refs.append((yo.detach(), torch.autograd.grad(yo, lo, ct.cpu().to(dt))))
        _as_good(y_t, refs[0][0], refs[1][0], 1e-4, 1e-4)                     # (std of near-constant rows: see _as_good)
        for a, b32, b64 in zip(g_t, refs[0][1], refs[1][1]):
            _as_good(a, b32, b64, 2e-4, 2e-4)


def test_edge_type_features_layer_equals_embedding_lookup():
    """DGNLayer (towers and complex) fed EdgeTypeFeatures(table, types) vs the same layer fed the gathered embedding rows
    (what dgn_net.py:75 hands the reference layer): same output, same parameter / input gradients, and a gradient for the table."""
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    b = synth.molecule_batch(50, seed=4, laplacian_eig=False)
    N, E = int(b["num_nodes"]), b["src"].numel()
    gen = torch.Generator().manual_seed(2)
    types = torch.randint(0, 4, (E,), generator=gen).to(dev)
    for type_net, towers in (("towers", 5), ("complex", 1)):
        torch.manual_seed(0)
        layer = dgn_amd.DGNLayer(70, 70, 0.0, True, True, "mean max std dir1-dx dir1-av", "identity amplification", {"log": torch.tensor(1.2)},
                                 type_net, True, towers=towers, divide_input=True, edge_features=True, edge_dim=10).model.to(dev).train()
        graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
        h0 = torch.randn(N, 70, generator=gen).to(dev)
        emb0 = torch.randn(4, 10, generator=gen).to(dev)
        snorm = b["snorm_n"].to(dev)
        ct = torch.randn(N, 70, generator=gen).to(dev)
        outs = []
        for mode in ("table", "rows"):
            h, emb = h0.clone().requires_grad_(True), emb0.clone().requires_grad_(True)
            e = dgn_amd.EdgeTypeFeatures(emb, types) if mode == "table" else emb.index_select(0, types)
            for p_ in layer.parameters():
                p_.grad = None
            for m in layer.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
            y = layer(graph, h, e, snorm)
            y.backward(ct)
            outs.append((y.detach(), h.grad, emb.grad, [p_.grad.clone() for p_ in layer.parameters() if p_.grad is not None]))
        (y1, gh1, ge1, gp1), (y2, gh2, ge2, gp2) = outs
        _close(y1, y2, 1e-5, 1e-5)
        _close(gh1, gh2, 1e-4, 1e-5)
        _close(ge1, ge2, 1e-4, 1e-4 * float(ge2.abs().max()))
        assert len(gp1) == len(gp2)
        for a, b_ in zip(gp1, gp2):
            _close(a, b_, 1e-4, 1e-4 * max(1e-3, float(b_.abs().max())))


def test_empty_and_edgeless_graphs():
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    plan = dgn_amd.make_plan(["mean", "dir1-dx"], ["identity", "amplification"])
    e0 = torch.zeros(0, dtype=torch.long, device=dev)
    g = dgn_amd.DGNGraph(e0, e0, 5, eig=torch.randn(5, 3, device=dev))
    h = torch.randn(5, 6, device=dev, requires_grad=True)
    y = directional_aggregate(g, plan, 1.0, x_src=h, x_in=h)
    assert y.shape == (5, 24) and float(y.abs().max()) == 0.0
    (gh,) = torch.autograd.grad(y, h, torch.ones_like(y))
    assert float(gh.abs().max()) == 0.0


def test_errors_are_loud():
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import directional_aggregate
    with pytest.raises(KeyError):
        dgn_amd.make_plan(["dir4-dx"], ["identity"])
    with pytest.raises(KeyError):
        dgn_amd.DGNLayer(4, 4, 0.0, True, True, "mean", "linear", {"log": torch.tensor(1.0)}, "simple", True)
    src = torch.tensor([0, 1], device=dev)
    g = dgn_amd.DGNGraph(src, src.flip(0), 2, eig=torch.randn(2, 2, device=dev))
    plan = dgn_amd.make_plan(["dir3-dx"], ["identity"])
    h = torch.randn(2, 4, device=dev)
    with pytest.raises(IndexError):                      # eig has 2 columns, dir3 needs column 3
        directional_aggregate(g, plan, 1.0, x_src=h, x_in=h)
    with pytest.raises(dgn_amd._lib.DgnError):          # CPU tensors are refused, no fallback
        directional_aggregate(g, dgn_amd.make_plan(["mean"], ["identity"]), 1.0, x_src=h.cpu())


def test_scale_combine_and_x_in_block_vs_torch():
    """dgn_scale_combine_* and the h_in pass-through block against their plain-torch restatements (fp32)."""
    dev = _dev()
    import dgn_amd
    from dgn_amd.ops import directional_aggregate, scale_combine
    from dgn_amd.spec import X_IN_NAME
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import oracle_scale_combine
    gen = torch.Generator().manual_seed(4)
    T_, N, S, fo = 3, 257, 3, 7
    z = torch.randn(T_, N, S * fo, generator=gen)
    sc, b, rs = torch.rand(N, S, generator=gen) + 0.5, torch.randn(T_ * fo, generator=gen), torch.rand(N, 1, generator=gen)
    ct = torch.randn(N, T_ * fo, generator=gen)
    for use_sc, use_b, use_rs in ((True, True, True), (False, True, False), (True, False, True)):
        zz = (z if use_sc else z[:, :, :fo].contiguous())
        zd = zz.to(dev).requires_grad_(True)
        bd = b.to(dev).requires_grad_(True) if use_b else None
        y = scale_combine(zd, sc.to(dev) if use_sc else None, bd, rs.to(dev) if use_rs else None)
        zo = zz.clone().requires_grad_(True)
        bo = b.clone().requires_grad_(True) if use_b else None
        yo = oracle_scale_combine(zo, sc if use_sc else None, bo, rs if use_rs else None)
        _close(y, yo, 1e-6, 1e-6)
        y.backward(ct.to(dev)); yo.backward(ct)
        _close(zd.grad, zo.grad, 1e-6, 1e-6)
        if use_b:
            _close(bd.grad, bo.grad, 1e-5, 1e-5)
    # pass-through block: [agg | x_in], also on the zero in-degree node, and its gradient
    src, dst = _random_graph(5, 40, 300)
    g = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), 40, eig=torch.randn(40, 3, generator=gen).to(dev))
    plan = dgn_amd.make_plan(["mean", "dir1-dx", X_IN_NAME], ["identity"])
    plain = dgn_amd.make_plan(["mean", "dir1-dx"], ["identity"])
    P = torch.randn(40, 6, generator=gen).to(dev).requires_grad_(True)
    H = torch.randn(40, 6, generator=gen).to(dev).requires_grad_(True)
    ya = directional_aggregate(g, plan, 1.0, x_src=P, x_in=H)
    yb = torch.cat([directional_aggregate(g, plain, 1.0, x_src=P, x_in=H), H], dim=1)
    _close(ya, yb, 0, 0)
    c2 = torch.randn(40, 18, generator=gen).to(dev)
    for a, b_ in zip(torch.autograd.grad(ya, [P, H], c2), torch.autograd.grad(yb, [P, H], c2)):
        _close(a, b_, 1e-5, 1e-6)


@pytest.mark.parametrize("act", ["none", "relu", "leaky_relu"])
@pytest.mark.parametrize("shape", [(1000, 70), (37, 300), (5, 1), (0, 8)])
def test_bias_act_vs_torch(act, shape):
    """bias + activation (+ residual) tail of an FCLayer: values and gradients (incl. the fused bias gradient)"""
    dev = _dev()
    from dgn_amd.ops import bias_act
    N, F_ = shape
    gen = torch.Generator().manual_seed(N + F_)
    x, b, res, ct = (torch.randn(N, F_, generator=gen), torch.randn(F_, generator=gen), torch.randn(N, F_, generator=gen),
                     torch.randn(N, F_, generator=gen))
    fn = {"none": lambda v: v, "relu": torch.relu, "leaky_relu": lambda v: torch.nn.functional.leaky_relu(v, 0.01)}[act]
    for use_res in (False, True):
        ld = [t.clone().to(dev).requires_grad_(True) for t in (x, b, res)]
        lc = [t.clone().double().requires_grad_(True) for t in (x, b, res)]
        y = bias_act(ld[0], ld[1], act, 0.01, ld[2] if use_res else None)
        yc = fn(lc[0] + lc[1]) + (lc[2] if use_res else 0)
        _close(y, yc.float(), 1e-6, 1e-6)
        gd = torch.autograd.grad(y, ld[:3] if use_res else ld[:2], ct.to(dev), allow_unused=True)
        gc = torch.autograd.grad(yc, lc[:3] if use_res else lc[:2], ct.double(), allow_unused=True)
        for a, r in zip(gd, gc):
            _close(a, r.float(), 1e-5, 1e-5 * max(1.0, float(r.abs().max()) if r.numel() else 1.0))


def test_bn_tail_vs_torch_batchnorm():
    """Fused layer tail ([relu](BatchNorm1d(x)) [+ residual]) against nn.BatchNorm1d: outputs, all gradients,
    running statistics and num_batches_tracked, one module and five per-tower modules, train and eval."""
    dev = _dev()
    from dgn_amd.ops import bn_tail
    gen = torch.Generator().manual_seed(6)
    N, W = 1000, 70
    x0 = torch.randn(N, W, generator=gen) * 3 + 1.5
    res0, ct = torch.randn(N, W, generator=gen), torch.randn(N, W, generator=gen)
    for n_bn, relu, use_res in ((1, True, True), (5, False, False), (1, False, True)):
        def make():
            bns = [torch.nn.BatchNorm1d(W // n_bn) for _ in range(n_bn)]
            g2 = torch.Generator().manual_seed(7)
            for b in bns:
                with torch.no_grad():
                    b.weight.copy_(torch.rand(b.weight.shape, generator=g2) + 0.5)
                    b.bias.copy_(torch.randn(b.bias.shape, generator=g2))
            return bns
        ref, mine = make(), [b.to(dev) for b in make()]
        xr = x0.clone().requires_grad_(True)
        rr = res0.clone().requires_grad_(True)
        w = W // n_bn
        yr = torch.cat([b(xr[:, i * w:(i + 1) * w]) for i, b in enumerate(ref)], dim=1)
        yr = torch.relu(yr) if relu else yr
        yr = yr + rr if use_res else yr
        yr.backward(ct)
        xm = x0.to(dev).requires_grad_(True)
        rm = res0.to(dev).requires_grad_(True)
        ym = bn_tail(xm, mine if n_bn > 1 else mine[0], True, relu=relu, residual=rm if use_res else None)
        ym.backward(ct.to(dev))
        _close(ym, yr, 1e-5, 1e-5)
        _close(xm.grad, xr.grad, 1e-4, 1e-5)
        if use_res:
            _close(rm.grad, rr.grad, 0, 0)
        for a, b in zip(mine, ref):
            _close(a.weight.grad, b.weight.grad, 1e-4, 1e-4)
            _close(a.bias.grad, b.bias.grad, 1e-4, 1e-4)
            _close(a.running_mean, b.running_mean, 1e-5, 1e-6)
            _close(a.running_var, b.running_var, 1e-5, 1e-6)
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
        for b in ref + mine:
            b.eval()
        with torch.no_grad():
            ye = torch.cat([b(x0[:, i * w:(i + 1) * w]) for i, b in enumerate(ref)], dim=1)
            ye = torch.relu(ye) if relu else ye
            _close(bn_tail(x0.to(dev), mine if n_bn > 1 else mine[0], False, relu=relu), ye, 1e-5, 1e-5)


@pytest.mark.parametrize("N,W", [(1003, 75), (997, 65), (5, 75), (1, 65), (3, 70), (4099, 70), (2, 3), (77, 255), (1030, 130)])
@pytest.mark.parametrize("relu,use_res", [(True, True), (False, False)])
def test_bn_tail_on_widths_that_are_no_multiple_of_four(N, W, relu, use_res):
    """Round 6 (csrc/dgn_bn_tail.hip column_partials_flat4 / bn_apply_flat4): dense rows of a width that is no multiple of four are read as
    one flat array of 16-byte chunks (a thread's four columns repeat with the period of lcm(W, 4) floats).  Against nn.BatchNorm1d in
    fp64: output, input / affine gradients, running statistics -- row counts that are no multiple of the period, fewer rows than a
    period, a ragged last chunk."""
    dev = _dev()
    from dgn_amd.ops import bn_tail
    gen = torch.Generator().manual_seed(N * 1000 + W)
    x0 = torch.randn(N, W, generator=gen) * 2 + 0.7
    res0, ct = torch.randn(N, W, generator=gen), torch.randn(N, W, generator=gen)
    ref, mine = torch.nn.BatchNorm1d(W).double(), torch.nn.BatchNorm1d(W).to(dev)
    with torch.no_grad():
        ga, be = torch.rand(W, generator=gen) + 0.5, torch.randn(W, generator=gen)
        for b in (ref, mine):
            b.weight.copy_(ga)
            b.bias.copy_(be)
    if N == 1:
        ref.eval(), mine.eval()                      # (torch refuses one row in training mode)
    xr = x0.double().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    yr = yr + res0.double() if use_res else yr
    yr.backward(ct.double())
    xm = x0.to(dev).requires_grad_(True)
    rm = res0.to(dev)
    ym = bn_tail(xm, mine, mine.training, relu=relu, residual=rm if use_res else None)
    ym.backward(ct.to(dev))
    tol = 2e-4 if N < 8 else 2e-5                    # (a handful of rows: the variance is small against eps-free cancellation)
    _close(ym, yr.float(), tol, tol)
    _close(xm.grad, xr.grad.float(), 10 * tol, tol)
    _close(mine.weight.grad, ref.weight.grad.float(), 10 * tol, 10 * tol)
    _close(mine.bias.grad, ref.bias.grad.float(), 10 * tol, 10 * tol)
    _close(mine.running_mean, ref.running_mean.float(), 1e-5, 1e-6)
    _close(mine.running_var, ref.running_var.float(), 1e-5, 1e-6)


@pytest.mark.parametrize("N,W", [(1003, 75), (997, 65), (5, 75), (3, 70), (4099, 70), (2, 3), (1030, 130)])
@pytest.mark.parametrize("relu,with_rows", [(True, True), (False, False)])
def test_one_tower_combine_bn_tail_on_widths_that_are_no_multiple_of_four(N, W, relu, with_rows):
    """Round 6 (csrc/dgn_combine.hip combine_bwd_flat4): the backward of bias + graph norm + BatchNorm (+ ReLU) of a one-tower, scaler-free
    posttrans output -- the simple / complex layers' tail -- on dense rows of a width that is no multiple of four runs on flat 16-byte
    chunks.  Against torch in fp64: output, d z, d bias, the affine gradients; ragged row counts, fewer rows than a period."""
    dev = _dev()
    from dgn_amd import ops
    gen = torch.Generator().manual_seed(N * 100 + W)
    z0 = torch.randn(1, N, W, generator=gen) * 1.5 + 0.3
    b0, rs0, ct = torch.randn(W, generator=gen), torch.rand(N, generator=gen) + 0.5, torch.randn(N, W, generator=gen)
    ga0, be0 = torch.rand(W, generator=gen) + 0.5, torch.randn(W, generator=gen)
    zr, br, gr, ber = (t.double().requires_grad_(True) for t in (z0, b0, ga0, be0))
    yr = zr[0] + br
    if with_rows:
        yr = yr * rs0.double()[:, None]
    mu, var = yr.mean(0), yr.var(0, unbiased=False)
    yr = (yr - mu) / torch.sqrt(var + 1e-5) * gr + ber
    yr = torch.relu(yr) if relu else yr
    yr.backward(ct.double())
    zm, bm, gm, bem = (t.to(dev).requires_grad_(True) for t in (z0, b0, ga0, be0))
    rm, rv = torch.zeros(W, device=dev), torch.ones(W, device=dev)
    ym = ops.combine_bn_tail(zm, None, bm, rs0.to(dev) if with_rows else None, gm, bem, rm, rv, None, 0.1, 1e-5, relu=relu)
    ym.backward(ct.to(dev))
    tol = 5e-4 if N < 8 else 3e-5
    _close(ym, yr.float(), tol, tol)
    _close(zm.grad, zr.grad.float(), 10 * tol, tol)
    _close(bm.grad, br.grad.float(), 10 * tol, 10 * tol)
    _close(gm.grad, gr.grad.float(), 10 * tol, 10 * tol)
    _close(bem.grad, ber.grad.float(), 10 * tol, 10 * tol)


@pytest.mark.parametrize("rows_per_wave", ["4"])
@pytest.mark.parametrize("case", ["towers", "pair_std", "simple", "edge_table"])
def test_grouped_row_backward_equals_row_per_wave(monkeypatch, rows_per_wave, case):
    """agg_bwd_short (2 / 4 rows of a molecule batch per wave, every load of the group in flight at once) against agg_bwd_rows
    (DGN_BWD_ROWS_PER_WAVE=1): bit-identical gradients, on static aggregator lists (the fast case) and with a ragged tail of rows;
    isolated atoms (zero in-degree) and a long row in the batch exercise the per-row fallback inside the grouped kernel."""
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.ops import directional_aggregate
    b = synth.molecule_batch(90, seed=17, laplacian_eig=False)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    # two isolated nodes and one node with 9 in-edges appended
    hub = N + 2
    extra_src = torch.arange(0, 9)
    src = torch.cat([src, extra_src])
    dst = torch.cat([dst, torch.full((9,), hub)])
    N = N + 3
    gen = torch.Generator().manual_seed(8)
    eig = torch.randn(N, 4, generator=gen)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    if case == "towers":          # the headline list: static operators, tower-major, h_in pass-through block
        F_, T, aggs, scalers = 70, 5, ["mean", "max", "min", "dir1-av", "dir1-dx"], ["identity"]
        from dgn_amd.dgn_layer import X_IN_NAME
        plan = dgn_amd.make_plan(aggs + [X_IN_NAME], scalers)
    elif case == "pair_std":
        F_, T, aggs, scalers = 12, 1, ["mean", "max", "std", "dir1-dx"], ["identity"]
        plan = dgn_amd.make_plan(aggs, scalers)
    elif case == "edge_table":     # P|Q + a 5-type edge table (DgnMsg.edge_type): the EDGE instantiation
        F_, T, aggs, scalers = 70, 5, ["mean", "max", "min", "dir1-av", "dir1-dx"], ["identity"]     # (a static list: the grouped path)
        from dgn_amd.dgn_layer import X_IN_NAME
        plan = dgn_amd.make_plan(aggs + [X_IN_NAME], scalers)
    else:
        F_, T, aggs, scalers = 76, 1, ["mean", "dir1-dx-no-abs"], ["identity", "amplification", "attenuation"]
        plan = dgn_amd.make_plan(aggs, scalers)
    table = torch.randn(5, F_, generator=gen)
    types_slot = graph.to_slot_order(torch.randint(0, 5, (src.numel(),), generator=gen).to(dev)).to(torch.int32).contiguous()
    X = torch.randn(N, F_, generator=gen)
    PQ = torch.randn(N, 2 * F_, generator=gen)

    def run(rb):
        monkeypatch.setattr(dgn_amd._lib.options, "bwd_rows_per_wave", int(rb))
        x = X.to(dev).requires_grad_(True)
        if case == "simple":
            y = directional_aggregate(graph, plan, 1.1, x_src=x, x_in=x)
            leaves = [x]
        else:
            pq = PQ.to(dev).requires_grad_(True)
            leaves = [pq, x]
            kw_e = {}
            if case == "edge_table":
                tb = table.to(dev).requires_grad_(True)
                leaves.append(tb)
                kw_e = dict(m_edge=tb, edge_type=types_slot)
            y = directional_aggregate(graph, plan, 1.1, x_pair=pq, x_in=x, n_towers=T, tower_major=T > 1, **kw_e)
        ct = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        return torch.autograd.grad(y, leaves, ct)

    ref = run("1")
    got = run(rows_per_wave)
    for a, r in zip(got, ref):
        assert torch.isfinite(a).all()
        assert torch.equal(a, r)


@pytest.mark.parametrize("residual,graph_norm,scalers", [(True, True, "identity amplification attenuation"), (False, False, "identity attenuation"),
                                                         (True, True, "identity")])
def test_whole_layer_call_equals_per_kernel_route(monkeypatch, residual, graph_norm, scalers):
    """dgn_towers_layer_forward / _backward (one C call per direction) against the same layer run kernel by kernel through
    the per-op autograd nodes: output, every gradient, BatchNorm running statistics."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)                 # both routes on the streaming Linear kernels
    b = synth.molecule_batch(200, seed=21, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    F_ = 70
    torch.manual_seed(2)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, graph_norm, True, "mean max min dir1-av dir1-dx", scalers, {"log": torch.tensor(1.2)}, "towers",
                             residual, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=gen))
    h0 = torch.randn(N, F_, device=dev, generator=gen)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    res = {}
    for whole in (True, False):
        monkeypatch.setattr(dgn_amd.ops, "WHOLE_LAYER", whole)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        used = []
        if whole:
            orig = dgn_amd.dgn_layer.towers_layer
            monkeypatch.setattr(dgn_amd.dgn_layer, "towers_layer", lambda *a, **k: (used.append(1), orig(*a, **k))[1])
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        if whole:
            assert used, "the whole-layer entry point was not taken"
            monkeypatch.setattr(dgn_amd.dgn_layer, "towers_layer", orig)
        res[whole] = (y.detach(), h.grad, {k: v.grad for k, v in lay.named_parameters()}, {k: v.clone() for k, v in lay.state_dict().items() if "running" in k or "num_batches" in k})
    (ya, ga, pa, sa), (yb, gb, pb, sb) = res[True], res[False]
    _close(ya, yb, 1e-6, 1e-6)
    _close(ga, gb, 1e-5, 1e-5)
    assert set(pa) == set(pb)
    for k in pa:
        assert pa[k] is not None and pb[k] is not None, k
        atol = 1e-5 * max(1.0, float(pb[k].abs().max()))
        if "posttrans" in k and k.endswith("bias"):
            # a bias in front of BatchNorm: its true gradient is ZERO; what either route returns is the rounding error of BatchNorm's
            # backward column sums (x gamma invstd) -- the whole-layer call derives them from the mixing weight gradient's fp32 partials
            # (round 6, csrc/dgn_towers.hip mix_bn_finalize), the per-kernel route from fp64 partials: both noise, N-row sums of O(1) terms
            atol = 2e-7 * N
        _close(pa[k], pb[k], 1e-5, atol, msg=k)
    for k in sa:
        _close(sa[k], sb[k], 1e-6, 1e-6, msg=k)


@pytest.mark.parametrize("residual,graph_norm,scalers,n_graphs", [(True, True, "identity amplification attenuation", 200), (False, False, "attenuation identity", 37),
                                                                   (True, True, "identity", 90)])
def test_towers_backward_with_the_fused_mixing_backward_is_bitwise_the_separate_passes(monkeypatch, residual, graph_norm, scalers, n_graphs):
    """Round 6 (csrc/dgn_towers.hip, option mix_bwd_fused): the mixing weight gradient straight from g_out and the activation mask
    (dgn_linear_wgrad_bn_act_mask), BatchNorm's backward + graph norm in the epilogue of the mixing network's input-gradient product,
    tower-major (dgn_linear_forward_act_mask_bnb), posttrans' bias gradient from the identity scaler's column sums of its weight-gradient
    pass -- against the round-5 sequence (masked gradient written, combine-backward pass): the same arithmetic in the same order, so every
    gradient has the same bits; only d b_post (a numerically zero tensor without graph norm) is summed in another order.  Partial last
    strips (row counts that are no multiple of 16) included."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import _lib, synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    monkeypatch.setattr(dgn_amd.ops, "WHOLE_LAYER_MIN_ROWS", 0)
    b = synth.molecule_batch(n_graphs, seed=29, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    F_ = 70
    torch.manual_seed(5)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, graph_norm, True, "mean max min dir1-av dir1-dx", scalers, {"log": torch.tensor(1.1)}, "towers",
                             residual, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=gen))
    h0 = torch.randn(N, F_, device=dev, generator=gen)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    res = {}
    used = []
    orig = dgn_amd.dgn_layer.towers_layer
    monkeypatch.setattr(dgn_amd.dgn_layer, "towers_layer", lambda *a, **k: (used.append(1), orig(*a, **k))[1])
    for fused in (1, 0):
        monkeypatch.setattr(_lib.options, "mix_bwd_fused", fused)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        res[fused] = (y.detach(), h.grad, {k: v.grad for k, v in lay.named_parameters()})
    assert len(used) == 2, "the whole-layer entry point was not taken"
    (ya, ga, pa), (yb, gb, pb) = res[1], res[0]
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    for k in pa:
        if "posttrans" in k and k.endswith("bias"):
            _close(pa[k], pb[k], 1e-5, (1e-5 if graph_norm else 2e-7 * N) * max(1.0, float(pb[k].abs().max())), msg=k)
        else:
            assert torch.equal(pa[k], pb[k]), k


@pytest.mark.parametrize("residual,n_graphs", [(True, 200), (False, 37), (True, 1)])
def test_towers_backward_with_the_one_pass_pretrans_backward_is_bitwise_the_two_kernels(monkeypatch, residual, n_graphs):
    """Round 6 (csrc/dgn_linear_bd.hip bd_backward_both, option bd_bwd_fused): the block-diagonal pretrans product's input gradient and its
    weight / bias gradient in ONE pass over d(P|Q) against bd_backward_input + bd_wgrad -- the input gradient has the same bits (same MFMAs in
    the same order per output); the weight gradient's partial sums meet in another order (fp32 rounding), reproducibly (partial last strips
    and a single-graph batch included)."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import _lib, synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    monkeypatch.setattr(dgn_amd.ops, "WHOLE_LAYER_MIN_ROWS", 0)
    b = synth.molecule_batch(n_graphs, seed=31, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    F_ = 70
    torch.manual_seed(6)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, "mean max min dir1-av dir1-dx", "identity amplification attenuation", {"log": torch.tensor(1.1)},
                             "towers", residual, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(8)
    h0 = torch.randn(N, F_, device=dev, generator=gen)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    res = {}
    used = []
    orig = dgn_amd.dgn_layer.towers_layer
    monkeypatch.setattr(dgn_amd.dgn_layer, "towers_layer", lambda *a, **k: (used.append(1), orig(*a, **k))[1])
    for fused in (1, 0):
        monkeypatch.setattr(_lib.options, "bd_bwd_fused", fused)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        res[fused] = (y.detach(), h.grad, {k: v.grad for k, v in lay.named_parameters()})
    assert len(used) == 2, "the whole-layer entry point was not taken"
    (ya, ga, pa), (yb, gb, pb) = res[1], res[0]
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    for k in pa:
        if "pretrans" in k:
            # the one-pass kernel's workgroups own other strips than bd_wgrad's: the partial sums meet in another order (fp32 rounding only)
            scale = float(pb[k].abs().max()) + 1e-30
            assert float((pa[k] - pb[k]).abs().max()) <= 2e-6 * scale + 1e-6, k
        else:
            assert torch.equal(pa[k], pb[k]), k
    # and it is reproducible run to run (fixed strip ownership, fixed summation order)
    monkeypatch.setattr(_lib.options, "bd_bwd_fused", 1)
    lay = copy.deepcopy(layer).train()
    h = h0.clone().requires_grad_(True)
    lay(graph, h, None, snorm).backward(ct)
    assert torch.equal(h.grad, ga)
    for k, v in lay.named_parameters():
        assert torch.equal(v.grad, pa[k]), k


@pytest.mark.parametrize("n_graphs,graph_norm", [(200, True), (37, False), (1, True)])
def test_towers_forward_with_batchnorm_statistics_from_the_posttrans_epilogue(monkeypatch, n_graphs, graph_norm):
    """Round 6 (csrc/dgn_linear_kernels.hpp ts_linear<.., kCombine>: LinParams.bn_part, option bn_stats_fused): BatchNorm's column sums of
    the posttrans output ride in that product's epilogue (fp64 cells in LDS, per-workgroup partials, bn_finalize) instead of bn_stats' pass
    over y0.  Same per-element values, another fp64 summation order: mean / invstd agree to fp32 rounding of an fp64 sum, the layer's
    output, every gradient, the running statistics and num_batches_tracked with them; reproducible run to run."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import _lib, synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    monkeypatch.setattr(dgn_amd.ops, "WHOLE_LAYER_MIN_ROWS", 0)
    b = synth.molecule_batch(n_graphs, seed=17, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    F_ = 70
    torch.manual_seed(9)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, graph_norm, True, "mean max min dir1-av dir1-dx", "identity amplification attenuation", {"log": torch.tensor(1.1)},
                             "towers", True, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(4)
    h0 = torch.randn(N, F_, device=dev, generator=gen) * 2.0 + 0.5
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    used = []
    orig = dgn_amd.dgn_layer.towers_layer
    monkeypatch.setattr(dgn_amd.dgn_layer, "towers_layer", lambda *a, **k: (used.append(1), orig(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(_lib.options, "bn_stats_fused", fused)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        bufs = {k: v.clone() for k, v in lay.named_buffers()}
        return y.detach(), h.grad, {k: v.grad for k, v in lay.named_parameters()}, bufs

    ya, ga, pa, ba = run(1)
    yb, gb, pb, bb = run(0)
    assert len(used) == 2, "the whole-layer entry point was not taken"

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 4e-6 * scale + 1e-7, what
    close(ya, yb, "y")
    close(ga, gb, "d h")
    for k in pa:
        close(pa[k], pb[k], k)
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]) == 1, k
        else:
            close(ba[k].float(), bb[k].float(), k)
    yc, gc, pc, bc = run(1)
    assert torch.equal(ya, yc) and torch.equal(ga, gc)
    for k in pa:
        assert torch.equal(pa[k], pc[k]), k
    for k in ba:
        assert torch.equal(ba[k], bc[k]), k


@pytest.mark.parametrize("type_net,F_,scalers,n_graphs", [("simple", 75, "identity amplification attenuation", 150), ("simple", 70, "identity amplification attenuation", 37),
                                                         ("simple", 65, "identity", 150), ("complex", 46, "identity amplification attenuation", 90),
                                                         ("complex", 45, "identity", 3)])
def test_dense_layer_forward_with_batchnorm_statistics_from_the_posttrans_pass(monkeypatch, type_net, F_, scalers, n_graphs):
    """Round 6 (option bn_stats_fused on the simple / complex whole-layer call): BatchNorm's column sums of the posttrans output ride in the
    degree-class product's epilogue (csrc/dgn_dc_kernels.hpp dc_tile: DcGemmParams.bn_part -- three scalers) or in the scale-combine pass
    (csrc/dgn_combine.hip combine_fwd: bn_part -- the folded route / one scaler) instead of bn_stats' pass over y.  Same values, another
    summation order: output, every gradient and the running statistics agree to fp32 rounding, num_batches_tracked counts once, and the
    result is reproducible run to run."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import _lib, synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    monkeypatch.setattr(dgn_amd.ops, "WHOLE_LAYER_MIN_ROWS", 0)
    b = synth.molecule_batch(n_graphs, seed=23, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    torch.manual_seed(12)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, "mean max min dir1-av dir1-dx", scalers, {"log": torch.tensor(1.1)},
                             type_net, True, towers=1, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    h0 = torch.randn(N, F_, device=dev, generator=gen) * 1.5 + 0.4
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    used = []
    orig = dgn_amd.ops.dense_layer
    monkeypatch.setattr(dgn_amd.ops, "dense_layer", lambda *a, **k: (used.append(1), orig(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(_lib.options, "bn_stats_fused", fused)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        return y.detach(), h.grad, {k: v.grad for k, v in lay.named_parameters()}, {k: v.clone() for k, v in lay.named_buffers()}

    ya, ga, pa, ba = run(1)
    yb, gb, pb, bb = run(0)
    assert len(used) == 2, "the whole-layer entry point was not taken"

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 5e-6 * scale + 1e-7, what
    close(ya, yb, "y")
    close(ga, gb, "d h")
    for k in pa:
        if pa[k] is not None:
            close(pa[k], pb[k], k)
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]) == 1, k
        else:
            close(ba[k].float(), bb[k].float(), k)
    yc, gc, pc, bc = run(1)
    assert torch.equal(ya, yc) and torch.equal(ga, gc)
    for k in ba:
        assert torch.equal(ba[k], bc[k]), k


@pytest.mark.parametrize("aggs,T", [("mean max min dir1-av dir1-dx", 5), ("mean max min dir1-dx dir1-av", 1), ("mean max min dir1-av dir1-dx", 1)])
@pytest.mark.parametrize("ties", [False, True])
def test_backward_from_the_aux_table_is_bitwise_the_recomputing_backward(monkeypatch, aggs, T, ties):
    """dgn_agg_forward_aux / dgn_agg_backward_aux: the forward records the slots of every row's first maximum / minimum and the sign of
    the dx residual, the backward works from that byte table instead of forming the messages again.  Gradients must be the bits of the
    recomputing backward -- also with ties everywhere (integer features: first occurrence wins) and with dx residuals that are exactly
    zero (constant eig column on part of the batch: sign(0) = 0)."""
    dev = _dev()
    import dgn_amd
    from dgn_amd import ops, synth
    b = synth.molecule_batch(60, seed=13, laplacian_eig=False)
    N = int(b["num_nodes"])
    eig = b["eig"].clone()
    eig[: N // 3] = 0.25                                           # no direction on these graphs: all weights zero
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=eig.to(dev))
    F_ = 70
    names = aggs.split()
    plan = dgn_amd.make_plan(names + ["__x_in__"], ["identity"]) if T > 1 else dgn_amd.make_plan(names, ["identity"])
    gen = torch.Generator().manual_seed(3)
    mk = (lambda: torch.randint(-2, 3, (N, F_), generator=gen).float().to(dev)) if ties else (lambda: torch.randn(N, F_, generator=gen).to(dev))
    xs, xd, xin = mk(), mk(), mk()
    nb = ops.agg_aux_bytes(graph, plan, T, F_, xs, xd, None, xin)
    assert nb >= N * F_, "this launch should have an aux table"
    w = graph.edge_weights(plan)
    A = plan.n_agg
    K = A * (F_ // T)
    out = torch.empty(T, N, K, device=dev) if T > 1 else torch.empty(N, A * F_, device=dev)
    out2 = torch.empty_like(out)
    aux = torch.full((nb,), 0xEE, dtype=torch.uint8, device=dev)
    ops.launch_forward(graph, plan, T, 1.0, w, xs, xd, None, xin, out, aux=aux)
    ops.launch_forward(graph, plan, T, 1.0, w, xs, xd, None, xin, out2)
    assert torch.equal(out, out2)                                    # the tracking forward computes the same values
    g_out = torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).to(dev)

    def run(a):
        sinks = [torch.full((N, F_), float("nan"), device=dev) for _ in range(3)]
        ops.launch_backward(graph, plan, T, 1.0, w, xs, xd, None, xin, g_out, sinks[0], sinks[1], None, sinks[2], accumulate=False, aux=a)
        return sinks

    with_aux, recomputed = run(aux), run(None)
    for x, y in zip(with_aux, recomputed):
        assert torch.isfinite(x).all()
        assert torch.equal(x, y)
    # poisoned operands prove that the aux backward of the grouped rows does not read the messages' inputs at all: only rows in groups
    # the per-row routine handles (a row with more than four or no in-edges in the group, the last partial group) may differ
    deg = (graph.indptr[1:] - graph.indptr[:-1]).long()
    grp_ok = ((deg >= 1) & (deg <= 4)).view(-1)[: (N // 4) * 4].view(-1, 4).all(1).repeat_interleave(4)
    xs_bad = torch.full_like(xs, float("nan"))
    sinks = [torch.full((N, F_), float("nan"), device=dev) for _ in range(3)]
    ops.launch_backward(graph, plan, T, 1.0, w, xs_bad, torch.full_like(xd, float("nan")), None, torch.full_like(xin, float("nan")), g_out,
                        sinks[0], sinks[1], None, sinks[2], accumulate=False, aux=aux)
    rows = torch.nonzero(grp_ok).view(-1)
    assert rows.numel() > N // 2
    assert torch.equal(sinks[1][rows], recomputed[1][rows])          # d x_dst of the grouped rows
    assert torch.equal(sinks[2][rows], recomputed[2][rows])          # d x_in


@pytest.mark.parametrize("kind", ["knn", "sbm"])
@pytest.mark.parametrize("form", ["simple", "complex"])
def test_backward_from_the_sign_table_on_longer_rows_is_bitwise_the_recomputing_backward(monkeypatch, kind, form):
    """Row-per-wave kernels (k-NN / SBM batches: CIFAR10, PATTERN json lists `mean dir1-dx dir2-dx`): the aux table holds the dx signs
    only, the backward then gathers nothing.  Same bits as the recomputing backward; rows with more than 64 in-edges (SBM) included.
    (The staged backward on both sides: the graph backward, which takes these batches with a sign table, writes d x_dst in closed
    form -- fp32 rounding apart, tests/test_block_backward_gpu.py.)"""
    dev = _dev()
    import dgn_amd
    from dgn_amd import ops, synth
    monkeypatch.setattr(dgn_amd.graph.DGNGraph, "_ensure_graph_blocks", lambda self, enabled=True: False)
    b = synth.knn_batch(n_graphs=6, seed=3) if kind == "knn" else synth.sbm_batch(n_graphs=3, seed=3)
    N = int(b["num_nodes"])
    eig = b["eig"].float().clone()
    eig[: N // 4] = 0.5                                              # residuals that are exactly zero on part of the batch
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=eig.to(dev))
    deg = (graph.indptr[1:] - graph.indptr[:-1])
    if kind == "sbm":
        assert int(deg.max()) > 64
    F_ = 66 if form == "simple" else 48
    names = ["mean", "dir1-dx", "dir2-dx"] + (["__x_in__"] if form == "complex" else [])
    plan = dgn_amd.make_plan(names, ["identity"])
    gen = torch.Generator().manual_seed(5)
    xs = torch.randn(N, F_, generator=gen).to(dev)
    xd = torch.randn(N, F_, generator=gen).to(dev) if form == "complex" else None
    xin = torch.randn(N, F_, generator=gen).to(dev) if form == "complex" else xs
    nb = ops.agg_aux_bytes(graph, plan, 1, F_, xs, xd, None, xin)
    assert nb >= N * F_, "this launch should have a sign table"
    w = graph.edge_weights(plan)
    out, out2 = torch.empty(N, plan.out_width(F_), device=dev), torch.empty(N, plan.out_width(F_), device=dev)
    aux = torch.full((nb,), 0xEE, dtype=torch.uint8, device=dev)
    ops.launch_forward(graph, plan, 1, 1.0, w, xs, xd, None, xin, out, aux=aux)
    ops.launch_forward(graph, plan, 1, 1.0, w, xs, xd, None, xin, out2)
    assert torch.equal(out, out2)
    g_out = torch.randn(out.shape, generator=gen).to(dev)

    def run(a, xs_, xd_, xin_):
        if form == "simple":
            g = torch.full((N, F_), float("nan"), device=dev)
            ops.launch_backward(graph, plan, 1, 1.0, w, xs_, None, None, xs_, g_out, g, None, None, g, accumulate=False, aux=a)
            return [g]
        sinks = [torch.full((N, F_), float("nan"), device=dev) for _ in range(3)]
        ops.launch_backward(graph, plan, 1, 1.0, w, xs_, xd_, None, xin_, g_out, sinks[0], sinks[1], None, sinks[2], accumulate=False, aux=a)
        return sinks

    ref = run(None, xs, xd, xin)
    got = run(aux, xs, xd, xin)
    for x, y in zip(got, ref):
        assert torch.isfinite(x).all()
        assert torch.equal(x, y)
    # with the table the backward reads none of the message operands
    nan = lambda t: None if t is None else torch.full_like(t, float("nan"))
    xs_bad = nan(xs)
    poisoned = run(aux, xs_bad, nan(xd), xs_bad if form == "simple" else nan(xin))
    for x, y in zip(poisoned, ref):
        assert torch.equal(x, y)


def test_towers_layer_with_the_aux_table_is_bitwise_the_recomputing_layer(monkeypatch):
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    b = synth.molecule_batch(150, seed=8, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    torch.manual_seed(1)
    layer = dgn_amd.DGNLayer(70, 70, 0.0, True, True, "mean max min dir1-av dir1-dx", "identity amplification attenuation", {"log": torch.tensor(1.2)},
                             "towers", True, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(2)
    h0 = torch.randn(N, 70, device=dev, generator=gen)
    ct = torch.randn(N, 70, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    res = []
    for aux in (True, False):
        monkeypatch.setattr(dgn_amd.ops, "AGG_AUX", aux)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        res.append([y.detach(), h.grad] + [p.grad for p in lay.parameters()])
    for a, c in zip(*res):
        assert torch.equal(a, c)


@pytest.mark.parametrize("n_graphs", [200, 3])
@pytest.mark.parametrize("residual", [True, False])
def test_towers_layer_with_the_activation_mask_is_bitwise_the_layer_with_z(monkeypatch, n_graphs, residual):
    """DgnTowersLayer.zmask: the mixing network's pre-activation kept as a sign mask (1/8 of the bytes).  Output and every gradient must be
    the bits of the run that stores z (DGN_NO_ZMASK=1); 3 graphs: only a partial last strip."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import _lib, synth
    assert _lib.load().dgn_towers_layer_zmask_supported(5, 14) == 1
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    b = synth.molecule_batch(n_graphs, seed=5, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    torch.manual_seed(4)
    layer = dgn_amd.DGNLayer(70, 70, 0.0, True, True, "mean max min dir1-av dir1-dx", "identity amplification attenuation", {"log": torch.tensor(1.2)},
                             "towers", residual, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(6)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=gen))
    h0 = torch.randn(N, 70, device=dev, generator=gen)
    ct = torch.randn(N, 70, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    res = []
    for mask in (True, False):
        if not mask:
            monkeypatch.setattr(dgn_amd._lib.options, "no_zmask", 1)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        res.append([y.detach(), h.grad] + [p.grad for p in lay.parameters()])
    names = ["y", "h"] + [k for k, _ in layer.named_parameters()]
    for k, a, c in zip(names, *res):
        assert torch.isfinite(a).all()
        if "posttrans" in k and k.endswith("bias"):
            # (round 6: with the mask the backward takes d b_post from the posttrans weight-gradient pass' column sums, the run with z
            #  from the combine-backward pass' partials: another summation order of the same terms)
            _close(a, c, 1e-5, 1e-5 * max(1.0, float(c.abs().max())), msg=k)
        else:
            assert torch.equal(a, c), k


@pytest.mark.parametrize("n_graphs,scalers", [(300, "identity amplification attenuation"), (7, "identity attenuation"), (40, "identity")])
def test_fused_forward_equals_separate_kernels(monkeypatch, n_graphs, scalers):
    """layer_fwd_fused (the posttrans product inside the sweep: aggregate rows in LDS -> MFMA -> scale-combine) against the
    sweep + streaming Linear + combine kernels it replaces on the no-grad path, and against the oracle's layer."""
    dev = _dev()
    import dgn_amd
    from dgn_amd import synth
    from oracle import dgn_oracle as orc
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    b = synth.molecule_batch(n_graphs, seed=31, laplacian_eig=False)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=b["eig"].to(dev))
    F_ = 70
    aggs = "mean max min dir1-av dir1-dx"
    torch.manual_seed(4)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, aggs, scalers, {"log": torch.tensor(1.15)}, "towers", True, towers=5,
                             edge_features=False, edge_dim=0).model
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen))
        for t in layer.towers:
            t.batchnorm_h.running_mean.normal_(generator=gen)
            t.batchnorm_h.running_var.uniform_(0.5, 1.5, generator=gen)
    h = torch.randn(N, F_, generator=gen)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(1.15), graph_norm=True, batch_norm=True, residual=True, towers=5,
               divide_input=True, edge_features=False)
    with torch.no_grad():
        yo, _ = orc.layer_forward("towers", sd, cfg, src, dst, N, b["eig"], h, None, b["snorm_n"], training=False)
    layer = layer.to(dev).eval()
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(dgn_amd.ops, "FUSED_FORWARD", fused)
        taken = []
        orig = dgn_amd.ops.fused_sweep_posttrans_forward
        monkeypatch.setattr(dgn_amd.ops, "fused_sweep_posttrans_forward", lambda *a, **k: (taken.append(1), orig(*a, **k))[1])
        with torch.no_grad():
            outs[fused] = layer(graph, h.to(dev), None, b["snorm_n"].to(dev))
        monkeypatch.setattr(dgn_amd.ops, "fused_sweep_posttrans_forward", orig)
        assert bool(taken) == fused
    assert torch.equal(outs[True], outs[False])                     # same arithmetic in the same order: bit-identical
    _close(outs[True], yo, 2e-5, 2e-5)


@pytest.mark.parametrize("type_net,F_,aggs,scalers,graph_norm,residual", [
    ("simple", 75, "mean dir1-dx-no-abs", "identity amplification attenuation", True, True),          # c1 (odd width: padded inside the call)
    ("simple", 70, "mean max min dir1-dx dir1-av", "identity", False, True),                            # HIV json: one scaler, no graph norm
    ("simple", 65, "mean dir1-dx dir2-dx", "identity", True, False),                                    # CIFAR10 json widths
    ("complex", 45, "mean dir1-dx dir1-av", "identity amplification attenuation", True, True),          # ZINC json
    ("complex", 20, "mean max std dir1-dx", "attenuation identity", False, True),                       # identity not first
    ("complex", 8, "mean max", "identity", True, False),                                                # one scaler
])
def test_whole_dense_layer_call_equals_per_kernel_route(monkeypatch, type_net, F_, aggs, scalers, graph_norm, residual):
    """dgn_dense_layer_forward / _backward (the simple / complex layers as one C call per direction: padding, weight folds and their
    adjoints inside) against the same layer run kernel by kernel through the per-op autograd nodes: output, d h, every parameter
    gradient (in the reference's layout), BatchNorm running statistics."""
    dev = _dev()
    import copy
    import dgn_amd
    from dgn_amd import synth
    monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", 0)
    monkeypatch.setattr(dgn_amd.ops, "WIDE_MIN_ROWS", 0)
    b = synth.molecule_batch(150, seed=11, laplacian_eig=False)
    N = int(b["num_nodes"])
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    torch.manual_seed(2)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, graph_norm, True, aggs, scalers, {"log": torch.tensor(1.2)}, type_net, residual,
                             edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, device=dev, generator=gen) / p.shape[1] ** 0.5)
            else:
                p.add_(0.1 * torch.randn(p.shape, device=dev, generator=gen))
    h0 = torch.randn(N, F_, device=dev, generator=gen)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    res = {}
    for whole in (True, False):
        monkeypatch.setattr(dgn_amd.ops, "WHOLE_LAYER", whole)
        lay = copy.deepcopy(layer).train()
        h = h0.clone().requires_grad_(True)
        used = []
        orig = dgn_amd.ops.dense_layer
        monkeypatch.setattr(dgn_amd.ops, "dense_layer", lambda *a, **k: (used.append(1), orig(*a, **k))[1])
        y = lay(graph, h, None, snorm)
        y.backward(ct)
        monkeypatch.setattr(dgn_amd.ops, "dense_layer", orig)
        assert bool(used) == whole, "whole-layer entry point " + ("not taken" if whole else "taken")
        res[whole] = (y.detach(), h.grad, {k: v.grad for k, v in lay.named_parameters()}, {k: v.clone() for k, v in lay.state_dict().items() if "running" in k or "num_batches" in k})
    (ya, ga, pa, sa), (yb, gb, pb, sb) = res[True], res[False]
    _close(ya, yb, 2e-6, 2e-6 * float(yb.abs().max()))
    _close(ga, gb, 1e-5, 1e-5 * float(gb.abs().max()))
    assert set(pa) == set(pb)
    for k in pa:
        assert pa[k] is not None and pb[k] is not None, k
        # (the posttrans bias feeds a BatchNorm: its exact gradient is zero, both routes return the rounding noise of a column sum)
        scale = float(ct.abs().sum(0).max()) if k.endswith("posttrans.fully_connected.0.linear.bias") else max(1.0, float(pb[k].abs().max()))
        _close(pa[k], pb[k], 1e-5, 2e-5 * scale, msg=k)
    for k in sa:
        _close(sa[k], sb[k], 1e-6, 1e-6, msg=k)

