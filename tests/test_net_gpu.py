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
