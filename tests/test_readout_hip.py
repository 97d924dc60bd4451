"""Readouts / VirtualNode on the HIP sweep (bipartite graph->nodes CSR) vs the reference fixtures (G8) and the oracle."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(sizes):
    return types.SimpleNamespace(batch_num_nodes=list(sizes), ndata={})


def _close(a, ref, rtol=1e-5, atol=1e-5, msg=""):
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol * scale, err_msg=msg)


def test_readout_modes_vs_reference(golden):
    import dgn_amd
    g8 = golden("g8_readouts")
    sizes, eig = g8["sizes"].tolist(), torch.from_numpy(g8["eig"]).cuda()
    for mode in ("sum", "max", "mean", "directional", "directional_abs"):
        h = torch.from_numpy(g8[f"readout/{mode}/h"]).cuda().requires_grad_(True)
        b = _batch(sizes)
        b.ndata["eig"] = eig
        hg = dgn_amd.readout(b, h, mode)
        _close(hg.detach().cpu().numpy(), g8[f"readout/{mode}/hg"], msg=mode)
        gh, = torch.autograd.grad(hg, [h], torch.from_numpy(g8[f"readout/{mode}/cot"]).cuda())
        _close(gh.cpu().numpy(), g8[f"readout/{mode}/gh"], msg=mode)


def test_virtual_node_vs_reference(golden):
    import dgn_amd
    g8 = golden("g8_readouts")
    sizes = g8["sizes"].tolist()
    for c in range(int(g8["vn/n_cases"])):
        pre = f"vn/c{c}"
        vn_type, b_norm, residual = g8[f"{pre}/cfg"].tolist()
        vn = dgn_amd.VirtualNode(dim=8, dropout=0.0, batch_norm=bool(int(b_norm)), bias=True, residual=bool(int(residual)),
                                 vn_type=vn_type).cuda()
        sd = {k.split("sd::")[1]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith(pre + "/sd::")}
        assert set(sd) == set(vn.state_dict())                       # same keys as the reference module
        vn.load_state_dict(sd)
        vn.train(True)
        h = torch.from_numpy(g8[f"{pre}/h"]).cuda().requires_grad_(True)
        vh = torch.from_numpy(g8[f"{pre}/vn_h"]).cuda().requires_grad_(True)
        vn_out, h_out = vn(_batch(sizes), h, vh)
        _close(vn_out.detach().cpu().numpy(), g8[f"{pre}/vn_out"])
        _close(h_out.detach().cpu().numpy(), g8[f"{pre}/h_out"])
        names = [n for n, _ in vn.named_parameters()]
        grads = torch.autograd.grad([vn_out, h_out], [h, vh] + list(vn.parameters()),
                                    [torch.from_numpy(g8[f"{pre}/cot_v"]).cuda(), torch.from_numpy(g8[f"{pre}/cot_h"]).cuda()])
        _close(grads[0].cpu().numpy(), g8[f"{pre}/gh"], rtol=1e-4)
        _close(grads[1].cpu().numpy(), g8[f"{pre}/gvn"], rtol=1e-4)
        for n, gr in zip(names, grads[2:]):
            _close(gr.cpu().numpy(), g8[f"{pre}/gp::{n}"], rtol=1e-4, msg=n)
        for k in g8.files:
            if k.startswith(pre + "/after::"):
                _close(vn.state_dict()[k.split("after::")[1]].cpu().numpy(), g8[k])


@pytest.mark.parametrize("F_", [6, 7, 70])
def test_readouts_random_batch_vs_oracle(F_):
    """ragged batch incl. an empty graph, single-node graphs and a 5000-node graph (hub-slice path), odd and even F"""
    import dgn_amd
    from oracle import readout_oracle as ro
    sizes = [3, 0, 1, 5000, 17, 1, 64, 2049]
    N = sum(sizes)
    gen = torch.Generator().manual_seed(F_)
    h = torch.randn(N, F_, generator=gen)
    eig = torch.randn(N, 3, generator=gen)
    b = _batch(sizes)
    b.ndata["eig"] = eig.cuda()
    for mode in ("sum", "max", "mean", "directional", "directional_abs"):
        hc, hd = h.clone().requires_grad_(True), h.clone().cuda().requires_grad_(True)
        ref = ro.readout(hc, sizes, mode, eig)
        out = dgn_amd.readout(b, hd, mode)
        assert out.shape == ref.shape
        _close(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-5, msg=mode)
        ct = torch.randn(ref.shape, generator=gen)
        g_ref, = torch.autograd.grad(ref, [hc], ct)
        g_out, = torch.autograd.grad(out, [hd], ct.cuda())
        _close(g_out.cpu().numpy(), g_ref.numpy(), rtol=2e-5, atol=2e-5, msg=mode)


class _DGLShaped:
    """What the reference's nets hand to the layer: a batched DGLGraph 0.4.  Only the surface dgn_amd touches:
    all_edges(order='eid'), number_of_nodes(), ndata (eig on the CPU, like the reference keeps it), batch_num_nodes."""

    def __init__(self, src, dst, n, eig, sizes):
        self._src, self._dst, self._n = src, dst, n
        self.ndata = {"eig": eig}
        self.edata = {}
        self.batch_num_nodes = list(sizes)

    def all_edges(self, order="eid"):
        assert order == "eid"
        return self._src, self._dst

    def number_of_nodes(self):
        return self._n


def test_drop_in_flow_with_a_dgl_shaped_graph(golden):
    """INTEGRATION.md route A end to end: the nets' call sequence (layer, g.ndata['h'] = h, readout) with a DGL-shaped
    batch object and CPU-resident eig/edge lists, against the same layer fed a DGNGraph and against the fixtures."""
    import dgn_amd
    from oracle import readout_oracle as ro
    g4, g8 = golden("g4_layers"), golden("g8_readouts")
    name = "towers_c2"
    dev = torch.device("cuda")
    src, dst, N = torch.from_numpy(g4["src"]), torch.from_numpy(g4["dst"]), int(g4["N"])
    eig = torch.from_numpy(g4[f"{name}/eig"])                              # stays on the CPU
    sizes = [17, 23, N - 40]
    batch = _DGLShaped(src, dst, N, eig, sizes)
    from test_hip_parity import _build_layer_from_fixture
    layer, train = _build_layer_from_fixture(g4, name, dev)
    layer.train(train)
    h = torch.from_numpy(g4[f"{name}/h"]).to(dev).requires_grad_(True)
    e = torch.from_numpy(g4[f"{name}/e"]).to(dev)
    snorm = torch.from_numpy(g4["snorm_n"]).to(dev)
    y = layer(batch, h, e, snorm)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g4[f"{name}/y"], rtol=2e-5, atol=2e-5)
    assert getattr(batch, "_dgn_graph", None) is not None                   # converted once, cached on the object
    batch.ndata["h"] = y
    hg = dgn_amd.readout(batch, "h", "directional")                        # ndata key, eig taken from the batch object
    ref = ro.readout(y.detach().cpu(), sizes, "directional", eig)
    np.testing.assert_allclose(hg.detach().cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    hg.sum().backward()
    assert h.grad is not None and torch.isfinite(h.grad).all()
    assert g8 is not None
