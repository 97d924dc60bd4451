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
