"""The graph-regression net around the layer (dgn_amd.nets.DGNNet) against fixture G10: the reference's own net run on the same
batch with the same weights -- scores, L1 loss, every parameter gradient, BatchNorm running statistics.  The `towers_edge` case feeds
the bond-type embedding to the layers as an edge-type table (EdgeTypeFeatures) where the reference gathers [E, edge_dim] rows."""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


def _net_params(cfg, edge_feat):
    type_net, _, mode = cfg
    return dict(num_atom_type=6, num_bond_type=4, hidden_dim=20, out_dim=20, in_feat_dropout=0.0, dropout=0.0, L=3, type_net=type_net,
                pos_enc_dim=0, readout=mode, graph_norm=True, batch_norm=True, aggregators="mean max dir1-av dir1-dx",
                scalers="identity amplification", avg_d={"log": torch.tensor(1.1)}, residual=True, edge_feat=edge_feat,
                edge_dim=6 if edge_feat else 0, pretrans_layers=1, posttrans_layers=1, device="cuda")


@gpu
@pytest.mark.parametrize("case", ["towers_edge", "towers", "simple"])
def test_net_vs_reference_fixture(golden, case):
    import dgn_amd
    from dgn_amd.nets import DGNNet
    g = golden("g10_net")
    dev = torch.device("cuda")
    cfg = [str(x) for x in g[f"{case}/cfg"]]
    edge_feat = cfg[1] == "1"
    net = DGNNet(_net_params(cfg, edge_feat))
    sd = {k.split("sd::", 1)[1]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(f"{case}/sd::")}
    assert set(sd) == set(net.state_dict()), set(sd) ^ set(net.state_dict())
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train(True)
    N = int(g["N"])
    graph = dgn_amd.DGNGraph(torch.from_numpy(g["src"]).to(dev), torch.from_numpy(g["dst"]).to(dev), N, eig=torch.from_numpy(g["eig"]).to(dev))
    graph.batch_num_nodes = [int(s) for s in g["sizes"]]
    atoms, bonds = torch.from_numpy(g["atoms"]).to(dev), torch.from_numpy(g["bonds"]).to(dev)
    snorm, targets = torch.from_numpy(g["snorm"]).to(dev), torch.from_numpy(g["targets"]).to(dev)
    scores = net(graph, atoms, bonds if edge_feat else None, snorm, None)
    loss = net.loss(scores, targets)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), g[f"{case}/scores"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(loss.item(), float(g[f"{case}/loss"]), rtol=1e-5)
    loss.backward()
    n_checked = 0
    for k, q in net.named_parameters():
        key = f"{case}/gp::{k}"
        if key in g:
            ref = g[key]
            np.testing.assert_allclose(q.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1e-2, float(np.abs(ref).max())), err_msg=k)
            n_checked += 1
    assert n_checked >= 15
    for k, v in net.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g[f"{case}/after::{k}"], rtol=1e-4, atol=1e-5, err_msg=k)


def test_mlp_readout_keys_and_shapes():
    from dgn_amd.nets import MLPReadout
    m = MLPReadout(40, 1)
    assert list(m.state_dict()) == ["FC_layers.0.weight", "FC_layers.0.bias", "FC_layers.1.weight", "FC_layers.1.bias", "FC_layers.2.weight", "FC_layers.2.bias"]
    assert [tuple(fc.weight.shape) for fc in m.FC_layers] == [(20, 40), (10, 20), (1, 10)]
    assert m(torch.zeros(3, 40)).shape == (3, 1)


@gpu
@pytest.mark.parametrize("block_route", [False, True])
def test_captured_net_step_equals_eager_training(monkeypatch, block_route):
    """hipgraph.CapturedNetStep (whole training step of the net -- forward, masked L1 loss, backward, Adam -- as ONE HIP graph over
    capacity-padded static buffers) against eager training on the unpadded batches: per-step losses and the parameters after
    several batches of different sizes.  ``block_route``: both sides on the graph-block layer route (the captured side through the
    padded graph's STATIC block table: unused entries, slots from the row pointers, BatchNorm over ``n_valid`` rows, zero padding rows)."""
    import copy
    import dgn_amd
    taken = []
    if block_route:
        monkeypatch.setattr(dgn_amd.ops, "BLOCK_LAYER_MAX_NODES", 32768)
        real = dgn_amd.ops.block_layer
        monkeypatch.setattr(dgn_amd.ops, "block_layer", lambda *a, **k: taken.append(a[0]) or real(*a, **k))
    from dgn_amd import synth
    from dgn_amd.hipgraph import CapturedNetStep
    from dgn_amd.nets import DGNNet
    dev = torch.device("cuda")
    torch.manual_seed(3)
    params = dict(num_atom_type=9, num_bond_type=4, hidden_dim=20, out_dim=20, in_feat_dropout=0.0, dropout=0.0, L=3, type_net="towers",
                  pos_enc_dim=0, readout="mean", graph_norm=True, batch_norm=True, aggregators="mean max min dir1-av dir1-dx",
                  scalers="identity amplification attenuation", avg_d={"log": torch.tensor(1.1)}, residual=True, edge_feat=False, edge_dim=0,
                  pretrans_layers=1, posttrans_layers=1, device="cuda")
    net_e = DGNNet(params).to(dev).train()
    net_c = copy.deepcopy(net_e)
    # an eager autograd pass of the SAME net on the default stream before the capture (fatal without rewrap_parameters)
    b0 = synth.molecule_batch(5, seed=1, laplacian_eig=False)
    g0 = dgn_amd.DGNGraph(b0["src"].to(dev), b0["dst"].to(dev), int(b0["num_nodes"]), eig=b0["eig"].to(dev))
    g0.batch_num_nodes = [int(s) for s in b0["sizes"]]
    rm = {k: v.clone() for k, v in net_c.state_dict().items()}
    net_c(g0, torch.zeros(int(b0["num_nodes"]), dtype=torch.long, device=dev), None, b0["snorm_n"].to(dev), None).sum().backward()
    net_c.load_state_dict(rm)                                   # (undo the BatchNorm running-statistics update of that pass)
    for p_ in net_c.parameters():
        p_.grad = None
    del g0
    gen = torch.Generator().manual_seed(5)
    batches = []
    for i, n_graphs in enumerate((24, 31, 17, 28)):
        b = synth.molecule_batch(n_graphs, seed=60 + i, laplacian_eig=False)
        N = int(b["num_nodes"])
        batches.append(dict(src=b["src"].to(dev), dst=b["dst"].to(dev), N=N, eig=b["eig"].to(dev), sizes=[int(s) for s in b["sizes"]],
                            atoms=torch.randint(0, 9, (N,), generator=gen).to(dev), snorm=b["snorm_n"].to(dev),
                            y=torch.randn(n_graphs, 1, generator=gen).to(dev)))
    order = [0, 0, 1, 2, 3, 1]                                  # (the first two = the capture's warm-up steps on batch 0)
    # eager reference: plain graphs, the net's own loss
    # (plain SGD on both sides: Adam's normalised update turns rounding-level gradient differences on near-zero-gradient elements
    #  into O(lr) parameter differences, which says nothing about the step being the same)
    opt = torch.optim.SGD(net_e.parameters(), lr=1e-2)
    losses_e = []
    for i in order:
        b = batches[i]
        g = dgn_amd.DGNGraph(b["src"], b["dst"], b["N"], eig=b["eig"])
        g.batch_num_nodes = b["sizes"]
        opt.zero_grad(set_to_none=True)
        loss = net_e.loss(net_e(g, b["atoms"], None, b["snorm"], None), b["y"])
        loss.backward()
        opt.step()
        losses_e.append(float(loss))
    # captured: capacity for the largest batch
    n_cap = max(b["N"] for b in batches) + 40
    e_cap = max(b["src"].numel() for b in batches) + 64
    from dgn_amd.hipgraph import rewrap_parameters
    rewrap_parameters(net_c)
    caps = dict(max_graph_nodes=max(max(b["sizes"]) for b in batches), max_graph_edges=4 * max(max(b["sizes"]) for b in batches)) if block_route else {}
    cs = CapturedNetStep(net_c, n_cap, e_cap, g_cap=40, eig_dim=batches[0]["eig"].shape[1], optimizer=torch.optim.SGD(net_c.parameters(), lr=1e-2), **caps)
    load = lambda b: cs.load(b["src"], b["dst"], b["N"], b["eig"], b["atoms"], b["snorm"], b["sizes"], b["y"])
    load(batches[0])
    cs.capture(warmup=2)                                          # two real steps on batch 0, then the capture (not executed)
    losses_c = []
    for i in order[2:]:
        load(batches[i])
        losses_c.append(float(cs.step()))
    np.testing.assert_allclose(losses_c, losses_e[2:], rtol=2e-4, atol=1e-5)
    if block_route:
        assert any(getattr(g, "_pad", None) is not None for g in taken), "the captured step did not take the graph-block route"
        assert any(getattr(g, "_pad", None) is None for g in taken), "the eager steps did not take the graph-block route"
        cs.pb.graph.check_deferred()                              # (no block beyond the capacity)
    for (k, a), (_, b) in zip(net_c.named_parameters(), net_e.named_parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-4, atol=2e-5, err_msg=k)
    for (k, a), (_, b) in zip(net_c.state_dict().items(), net_e.state_dict().items()):
        if "running" in k:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-3, atol=1e-5, err_msg=k)


@gpu
def test_captured_net_step_default_optimizer_trains():
    """The default construction (parameters re-wrapped, fused capturable Adam): the loss of a fixed batch goes down over replays."""
    from dgn_amd import synth
    from dgn_amd.hipgraph import CapturedNetStep
    from dgn_amd.nets import DGNNet
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = DGNNet(dict(num_atom_type=9, num_bond_type=4, hidden_dim=20, out_dim=20, in_feat_dropout=0.0, dropout=0.0, L=2, type_net="towers", pos_enc_dim=0,
                      readout="mean", graph_norm=True, batch_norm=True, aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation",
                      avg_d={"log": torch.tensor(1.1)}, residual=True, edge_feat=False, edge_dim=0, pretrans_layers=1, posttrans_layers=1,
                      device="cuda")).to(dev).train()
    b = synth.molecule_batch(30, seed=9, laplacian_eig=False)
    N = int(b["num_nodes"])
    gen = torch.Generator().manual_seed(1)
    cs = CapturedNetStep(net, N + 50, b["src"].numel() + 50, g_cap=33, eig_dim=b["eig"].shape[1], lr=5e-3)
    cs.load(b["src"].to(dev), b["dst"].to(dev), N, b["eig"].to(dev), torch.randint(0, 9, (N,), generator=gen).to(dev), b["snorm_n"].to(dev),
            [int(s) for s in b["sizes"]], torch.randn(30, 1, generator=gen).to(dev))
    cs.capture(warmup=2)
    first = float(cs.step())
    for _ in range(60):
        last = float(cs.step())
    assert np.isfinite(last) and last < 0.7 * first, (first, last)
