"""GPU parity of the dense API (dgn_amd.dense, models/pytorch semantics) against golden vectors produced by the
imported reference: every working aggregator (with and without self loops), all scalers, dense DGNLayer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_dense_aggregators_vs_reference(golden):
    dev = _dev()
    from dgn_amd import dense
    g = golden("g6_dense")
    avg_d = {"log": torch.tensor(float(g["avg_log"]), device=dev), "lin": torch.tensor(float(g["avg_lin"]), device=dev)}
    for c in g["cases"].tolist():
        X, adj, eig = T(g[f"{c}/X"]).to(dev), T(g[f"{c}/adj"]).to(dev), T(g[f"{c}/eig"]).to(dev)
        B, N, _, F_ = X.shape
        db = dense.DenseBatch(adj)
        for sl in (0, 1):
            for name in g["names"].tolist():
                XX = X.clone().requires_grad_(True)
                msg = XX[db.b, db.i, db.j]
                y = dense.aggregate(name, db, msg, eig, bool(sl), avg_d).reshape(B, N, -1)
                tag = f"{c}/sl{sl}/{name}"
                np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}/y"], rtol=2e-5, atol=2e-5, err_msg=tag)
                (gX,) = torch.autograd.grad(y, XX, T(g[f"{tag}/cot"]).to(dev))
                np.testing.assert_allclose(gX.cpu().numpy(), g[f"{tag}/gX"], rtol=1e-4, atol=2e-5, err_msg=tag + " gX")
        m = T(g[f"{c}/scaler_in"]).to(dev)
        for s in dense.SCALER_NAMES:
            got = dense._scale(s, m.reshape(B * N, -1), adj, avg_d).reshape(B, N, -1)
            np.testing.assert_allclose(got.cpu().numpy(), g[f"{c}/scaler/{s}"], rtol=1e-6, atol=1e-6)
        for bad in g["broken"].tolist():
            with pytest.raises(TypeError):
                dense.aggregate(bad, db, X[db.b, db.i, db.j], eig, False, avg_d)


def test_dense_layers_vs_reference(golden):
    dev = _dev()
    from dgn_amd import dense
    g = golden("g6_dense")
    avg_d = {"log": torch.tensor(float(g["avg_log"]), device=dev), "lin": torch.tensor(float(g["avg_lin"]), device=dev)}
    for name in g["layer_cases"].tolist():
        meta = g[f"{name}/meta"].tolist()
        layer = dense.DGNLayer(in_features=int(meta[3]), out_features=int(meta[4]), aggregators=meta[1].split(),
                               scalers=meta[2].split(), NN_eig=False, avg_d=avg_d, eigs=None, towers=int(meta[0]),
                               self_loop=False, divide_input=bool(int(meta[6])), device="cuda")
        sd = {k[len(name) + 5:]: T(g[k]) for k in g.files if k.startswith(f"{name}/sd::")}
        assert {k: tuple(v.shape) for k, v in layer.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
        layer.load_state_dict(sd)
        layer = layer.to(dev)
        c = meta[5]
        inp = T(g[f"{name}/input"]).to(dev).requires_grad_(True)
        y = layer(inp, T(g[f"{c}/adj"]).to(dev), T(g[f"{c}/eig"]).to(dev))
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{name}/y"], rtol=5e-5, atol=2e-5, err_msg=name)
        params = dict(layer.named_parameters())
        pn = [k[len(name) + 5:] for k in g.files if k.startswith(f"{name}/gp::")]
        grads = torch.autograd.grad(y, [inp] + [params[k] for k in pn], T(g[f"{name}/cot"]).to(dev))
        np.testing.assert_allclose(grads[0].cpu().numpy(), g[f"{name}/ginput"], rtol=2e-4, atol=5e-5)
        for k, gr in zip(pn, grads[1:]):
            np.testing.assert_allclose(gr.cpu().numpy(), g[f"{name}/gp::{k}"], rtol=2e-4, atol=1e-4, err_msg=f"{name} {k}")
    with pytest.raises(TypeError):      # the reference's constructor probe fails on these, so does ours
        dense.DGNLayer(4, 4, ["mean_attenuated"], ["identity"], False, avg_d, None)
    with pytest.raises(KeyError):
        dense.DGNLayer(4, 4, ["dir6-dx"], ["identity"], False, avg_d, None)
