"""Host logic of the layers without a GPU: the layer algebra around the kernels (P|Q decomposition, scaler folding, h_in
pass-through block, zero-column padding of odd hidden sizes, weight re-arrangements) with the aggregation served by the CPU oracle
(tests/oracle_backend.py), against the oracle's reference-structured ``layer_forward``.  The kernels themselves are covered by the
``-m gpu`` tests; this file pins the algebra so that a wrong permutation or padding is caught here, on CPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture
def oracle_backend(monkeypatch):
    import dgn_amd.dgn_layer as dl
    import oracle_backend as ob
    monkeypatch.setattr(dl, "directional_aggregate", ob.oracle_directional_aggregate)
    monkeypatch.setattr(dl, "scale_combine", ob.oracle_scale_combine)
    monkeypatch.setattr(dl, "bn_tail", ob.oracle_bn_tail)
    monkeypatch.setattr(dl, "bn_tail_fused", ob.oracle_bn_tail_fused)
    monkeypatch.setattr(dl, "combine_bn_tail", ob.oracle_combine_bn_tail)
    return ob


CASES = [
    # type_net, hidden, aggregators, scalers, graph_norm, towers, edge_dim
    ("complex", 45, "mean dir1-dx dir1-av", "identity amplification attenuation", True, 1, 0),      # ZINC json (odd: padded path)
    ("complex", 47, "mean dir1-dx dir2-dx", "identity amplification attenuation", True, 1, 0),      # PATTERN json
    ("complex", 9, "mean max dir1-dx std", "identity attenuation", True, 1, 3),                     # odd + edge features
    ("complex", 8, "mean max dir1-dx", "identity amplification attenuation", False, 1, 0),          # even (unpadded)
    ("simple", 7, "mean max min dir1-dx dir1-av", "identity amplification attenuation", False, 1, 0),   # HIV list, odd width
    ("simple", 70, "mean max min dir1-dx dir1-av", "identity", False, 1, 0),                        # HIV json as shipped
    ("towers", 10, "mean max min dir1-av dir1-dx", "identity amplification attenuation", True, 5, 0),
    ("towers", 15, "mean dir1-dx", "identity amplification attenuation", True, 5, 0),               # odd per-tower width
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-h{c[1]}-s{len(c[3].split())}-e{c[6]}" for c in CASES])
def test_layer_algebra_vs_oracle_layer(oracle_backend, case):
    import dgn_amd
    from dgn_amd import synth
    from oracle import dgn_oracle as orc
    type_net, F_, aggs, scalers, graph_norm, towers, edge_dim = case
    b = synth.molecule_batch(6, seed=3, laplacian_eig=False)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    E = src.numel()
    avg = 1.1
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, graph_norm, True, aggs, scalers, {"log": torch.tensor(avg)}, type_net, True, towers=towers,
                             edge_features=edge_dim > 0, edge_dim=edge_dim).model.double()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen, dtype=torch.float64) / p.shape[1] ** 0.5)
            else:
                p.add_(0.1 * torch.randn(p.shape, generator=gen, dtype=torch.float64))
    h = torch.randn(N, F_, generator=gen, dtype=torch.float64)
    ct = torch.randn(N, F_, generator=gen, dtype=torch.float64)
    ef = torch.randn(E, edge_dim, generator=gen, dtype=torch.float64) if edge_dim else None
    eig, snorm = b["eig"].double(), b["snorm_n"].double()

    sd = {k: (v.detach().clone().requires_grad_("running" not in k) if v.dtype.is_floating_point else v.clone())
          for k, v in layer.state_dict().items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and v.requires_grad]
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(avg, dtype=torch.float64), graph_norm=graph_norm, batch_norm=True,
               residual=True, towers=towers, divide_input=True, edge_features=edge_dim > 0)
    ho = h.clone().requires_grad_(True)
    yo, stats = orc.layer_forward(type_net, sd, cfg, src, dst, N, eig, ho, ef, snorm, training=True)
    go = torch.autograd.grad(yo, [ho] + [sd[k] for k in names], ct)

    layer.train()
    graph = dgn_amd.DGNGraph(src, dst, N, eig=eig)
    hd = h.clone().requires_grad_(True)
    y = layer(graph, hd, ef, snorm)
    params = dict(layer.named_parameters())
    gd = torch.autograd.grad(y, [hd] + [params[k] for k in names], ct)
    np.testing.assert_allclose(y.detach().numpy(), yo.detach().numpy(), rtol=2e-6, atol=2e-6)        # (the degree-scaler table is fp32 by the reference's own semantics)
    for a, r, k in zip(gd, go, ["h"] + names):
        np.testing.assert_allclose(a.numpy(), r.numpy(), rtol=1e-5, atol=1e-5 * max(1.0, float(r.abs().max())), err_msg=k)
    for k, v in stats.items():
        np.testing.assert_allclose(layer.state_dict()[k].numpy(), v.numpy(), rtol=1e-6, atol=1e-8, err_msg=k)


def test_degree_class_algebra_and_virtual_row_space():
    """Degree-class posttrans on CPU (the kernels: tests/test_dc_hip.py): (1) DGNGraph.degree_classes() -- every node once, units hold one
    class, classes ascend, stable inside a class; (2) the identity the route rests on, with the oracle's scalers
    (oracle/dgn_oracle.py, nets/scalers.py:7-18): posttrans(cat(agg x scalers)) == agg (sum_s scale_s(deg) W_s)^T per in-degree."""
    import dgn_amd
    from dgn_amd.dgn_layer import _scale_table
    from dgn_amd.spec import make_plan
    g = torch.Generator().manual_seed(0)
    N = 700
    deg = torch.randint(0, 7, (N,), generator=g)
    dst = torch.repeat_interleave(torch.arange(N), deg)
    src = torch.randint(0, N, (dst.numel(),), generator=g)
    graph = dgn_amd.DGNGraph(src, dst, N)
    dc = graph.degree_classes()
    vperm, uc = dc["vperm"], dc["unit_class"]
    assert vperm.numel() == 64 * dc["n_units"]
    live = vperm[vperm >= 0]
    assert sorted(live.tolist()) == list(range(N))
    for u in range(dc["n_units"]):
        rows = vperm[64 * u: 64 * u + 64]
        rows = rows[rows >= 0].long()
        assert rows.numel() and bool((deg[rows] == uc[u]).all())
    assert bool((uc[1:] >= uc[:-1]).all())
    assert dc["present"].tolist() == torch.bincount(deg, minlength=32).tolist()
    for c in range(7):
        nodes = live[deg[live.long()] == c]
        assert bool((nodes[1:] > nodes[:-1]).all())
    assert bool((deg[dc["rep"]][dc["present"] > 0] == torch.arange(32)[dc["present"] > 0]).all())
    # the algebra, fp64: scalers applied to the aggregates then one Linear == per-class folded weights
    plan = make_plan(["mean", "max"], ["identity", "amplification", "attenuation"])
    sc = _scale_table(graph, plan.applied_scalers, 1.3).double()                 # [N, S] per node
    S, A, F, fo = sc.shape[1], 2, 6, 5
    agg = torch.randn(N, A * F, generator=g, dtype=torch.float64)
    W = torch.randn(fo, S * A * F, generator=g, dtype=torch.float64)             # reference layout: scaler-major input blocks
    ref = torch.cat([agg * sc[:, s:s + 1] for s in range(S)], dim=1) @ W.t()
    cls = sc[dc["rep"]]                                                          # [32, S]: the class rows of the table
    Wc = torch.einsum("cs,osk->cok", cls, W.view(fo, S, A * F))
    out = torch.einsum("nk,nok->no", agg, Wc[deg])
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-12, atol=1e-12)
    # a graph with an in-degree of 32 or more keeps the folded route
    dst2 = torch.cat([torch.zeros(40, dtype=torch.long), torch.arange(1, 50)])
    assert dgn_amd.DGNGraph(torch.randint(0, 50, (dst2.numel(),), generator=g), dst2, 50).degree_classes() is None


def test_convert_sync_batchnorm_shares_parameters_and_is_plain_batchnorm_without_a_process_group():
    """dist.convert_sync_batchnorm (optional SyncBN): same parameter / buffer OBJECTS and state_dict keys; without an
    initialised process group (or in eval mode) the module is nn.BatchNorm1d."""
    import torch
    from dgn_amd import dist as ddist
    from dgn_amd.layers import MLP
    torch.manual_seed(0)
    net = torch.nn.Sequential(MLP(6, 8, 4, layers=2, mid_b_norm=True, last_b_norm=True), torch.nn.BatchNorm1d(4))
    ref = torch.nn.Sequential(MLP(6, 8, 4, layers=2, mid_b_norm=True, last_b_norm=True), torch.nn.BatchNorm1d(4))
    ref.load_state_dict(net.state_dict())
    before = {n: p for n, p in net.named_parameters()}
    keys = list(net.state_dict().keys())
    ddist.convert_sync_batchnorm(net)
    assert list(net.state_dict().keys()) == keys
    assert all(p is before[n] for n, p in net.named_parameters())
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    assert len(bns) == 3 and all(isinstance(m, ddist.SyncBatchNorm1d) for m in bns)
    x = torch.randn(10, 6)
    for mode in (True, False):
        net.train(mode), ref.train(mode)
        torch.testing.assert_close(net(x), ref(x))
    for (n, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        torch.testing.assert_close(a, b, msg=n)
