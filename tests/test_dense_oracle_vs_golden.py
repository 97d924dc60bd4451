"""Pins oracle/dense_oracle.py (dense models/pytorch formulation) to golden vectors produced by importing
the reference (tests/golden/make_golden.py::g6_dense)."""
import numpy as np
import pytest
import torch

from oracle import dense_oracle as dorc

torch.set_num_threads(1)
T = torch.from_numpy


def _avg(g):
    return {"log": torch.tensor(float(g["avg_log"])), "lin": torch.tensor(float(g["avg_lin"]))}


def test_dense_registry(golden):
    g = golden("g6_dense")
    assert sorted(dorc.AGGREGATOR_NAMES) == sorted(g["names"].tolist())
    for b in g["broken"].tolist():
        with pytest.raises(TypeError):
            dorc.aggregate(b, torch.zeros(1, 2, 2, 1), torch.ones(1, 2, 2))


def test_dense_aggregators_and_scalers(golden):
    g = golden("g6_dense")
    avg_d = _avg(g)
    for c in g["cases"].tolist():
        X, adj, eig = T(g[f"{c}/X"]), T(g[f"{c}/adj"]), T(g[f"{c}/eig"])
        for sl in (0, 1):
            for name in g["names"].tolist():
                XX = X.clone().requires_grad_(True)
                y = dorc.aggregate(name, XX, adj, eig, bool(sl), avg_d)
                tag = f"{c}/sl{sl}/{name}"
                np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}/y"], rtol=1e-5, atol=1e-6, err_msg=tag)
                (gX,) = torch.autograd.grad(y, XX, T(g[f"{tag}/cot"]))
                np.testing.assert_allclose(gX.numpy(), g[f"{tag}/gX"], rtol=1e-4, atol=1e-5, err_msg=tag)
        m = T(g[f"{c}/scaler_in"])
        for s in ("identity", "amplification", "attenuation", "linear", "inverse_linear"):
            np.testing.assert_allclose(dorc.scale(s, m, adj, avg_d).numpy(), g[f"{c}/scaler/{s}"], rtol=1e-6, atol=1e-7)


def test_dense_layers(golden):
    g = golden("g6_dense")
    avg_d = _avg(g)
    for name in g["layer_cases"].tolist():
        meta = g[f"{name}/meta"].tolist()
        cfg = dict(towers=int(meta[0]), aggregators=meta[1].split(), scalers=meta[2].split(), avg_d=avg_d,
                   divide_input=bool(int(meta[6])), self_loop=False)
        c = meta[5]
        sd = {k[len(name) + 5:]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith(f"{name}/sd::")}
        inp = T(g[f"{name}/input"]).clone().requires_grad_(True)
        y = dorc.layer_forward(sd, cfg, inp, T(g[f"{c}/adj"]), T(g[f"{c}/eig"]))
        np.testing.assert_allclose(y.detach().numpy(), g[f"{name}/y"], rtol=2e-5, atol=2e-6, err_msg=name)
        pn = [k[len(name) + 5:] for k in g.files if k.startswith(f"{name}/gp::")]
        grads = torch.autograd.grad(y, [inp] + [sd[k] for k in pn], T(g[f"{name}/cot"]))
        np.testing.assert_allclose(grads[0].numpy(), g[f"{name}/ginput"], rtol=1e-4, atol=1e-5)
        for k, gr in zip(pn, grads[1:]):
            np.testing.assert_allclose(gr.numpy(), g[f"{name}/gp::{k}"], rtol=1e-4, atol=2e-5, err_msg=f"{name} {k}")
