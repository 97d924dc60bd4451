"""Pins oracle/dgn_oracle.py to the reference: every golden vector produced by
importing the reference (tests/golden/make_golden.py) must be reproduced."""
import numpy as np
import pytest
import torch

from oracle import dgn_oracle as orc

torch.set_num_threads(1)
T = torch.from_numpy


def _close(a, b, rtol=1e-6, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_registry_names(golden):
    g1 = golden("g1_aggregators")
    assert sorted(orc.AGGREGATOR_NAMES) == sorted(g1["names"].tolist())
    assert len(orc.AGGREGATOR_NAMES) == 24
    with pytest.raises(KeyError):
        orc.get_aggregator("dir4-dx")          # DGL-path registry stops at dir3 (aggregators.py:76-93)
    orc.get_aggregator("dir1-smooth")            # alias of dir1-av


@pytest.mark.parametrize("fixture", ["g1_aggregators", "g5_edge_cases"])
def test_mailbox_aggregators(golden, fixture):
    g = golden(fixture)
    names = g["names"].tolist()
    cases = [f"c{i}" for i in range(int(g["n_cases"]))] if fixture == "g1_aggregators" else g["cases"].tolist()
    for c in cases:
        h, es, ed = T(g[f"{c}/h"]), T(g[f"{c}/eig_s"]), T(g[f"{c}/eig_d"])
        x, ct = T(g[f"{c}/h_in"]), T(g[f"{c}/cot"])
        for name in names:
            hh = h.clone().requires_grad_(True)
            xx = x.clone().requires_grad_(True)
            y = orc.get_aggregator(name)(hh, es, ed, xx)
            gh, gx = torch.autograd.grad(y, [hh, xx], ct, allow_unused=True)
            _close(y.detach(), g[f"{c}/{name}/y"])
            _close(gh, g[f"{c}/{name}/gh"])
            _close(gx if gx is not None else torch.zeros_like(x), g[f"{c}/{name}/gx"])


def test_dir_smooth_alias(golden):
    g = golden("g1_aggregators")
    h, es, ed, x = (T(g[f"c3/{k}"]) for k in ("h", "eig_s", "eig_d", "h_in"))
    _close(orc.get_aggregator("dir2-smooth")(h, es, ed, x), g["c3/dir2-av/y"], 0, 0)


def test_scalers_bitwise(golden):
    g = golden("g2_scalers")
    h = T(g["h"])
    for ai, avg in enumerate(g["avg"].tolist()):
        for D in g["D"].tolist():
            for name in orc.SCALER_NAMES:
                got = orc.scale(name, h, int(D), torch.tensor(avg, dtype=torch.float32))
                assert np.array_equal(got.numpy(), g[f"a{ai}/D{D}/{name}"]), (name, D, avg)


def test_reduce_concat_order_and_single_scaler_rule(golden):
    g = golden("g3_reduce")
    src, dst, N = T(g["src"]), T(g["dst"]), int(g["N"])
    h, eig = T(g["h"]), T(g["eig"])
    aggs = str(g["aggregators"]).split()
    for tag in ("id", "amp_only", "three", "att_amp"):
        scalers = str(g[f"{tag}/scalers"]).split()
        hh = h.clone().requires_grad_(True)
        y = orc.aggregate_graph(src, dst, N, hh[src], eig, hh, aggs, scalers, torch.tensor(0.8))
        assert y.shape == g[f"{tag}/y"].shape
        _close(y.detach(), g[f"{tag}/y"])
        gh = torch.autograd.grad(y, hh, T(g[f"{tag}/cot"]))[0]
        _close(gh, g[f"{tag}/gh"], 1e-5, 1e-6)
    # a lone non-identity scaler is silently not applied (dgn_layer.py:170)
    assert np.array_equal(g["id/y"], g["amp_only/y"])


def _layer_case(g, name):
    meta = g[f"{name}/meta"].tolist()
    cfg = dict(aggregators=meta[3], scalers=meta[4], avg_log=torch.tensor(float(meta[5])), towers=int(meta[6]),
               divide_input=bool(int(meta[7])), edge_features=bool(int(meta[8])), graph_norm=True, batch_norm=True,
               residual=True)
    return meta[0], cfg, bool(int(meta[12]))


def test_layers(golden):
    g = golden("g4_layers")
    src, dst, N = T(g["src"]), T(g["dst"]), int(g["N"])
    snorm = T(g["snorm_n"])
    for name in g["cases"].tolist():
        type_net, cfg, train = _layer_case(g, name)
        sd = {k[len(name) + 5:]: T(g[k]).clone() for k in g.files if k.startswith(f"{name}/sd::")}
        pnames = [k[len(name) + 5:] for k in g.files if k.startswith(f"{name}/gp::")]
        for k in pnames:
            sd[k].requires_grad_(True)
        h = T(g[f"{name}/h"]).clone().requires_grad_(True)
        e = T(g[f"{name}/e"]).clone().requires_grad_(True)
        y, stats = orc.layer_forward(type_net, sd, cfg, src, dst, N, T(g[f"{name}/eig"]), h, e, snorm, training=train)
        _close(y.detach(), g[f"{name}/y"], 2e-5, 2e-6)
        grads = torch.autograd.grad(y, [h, e] + [sd[k] for k in pnames], T(g[f"{name}/cot"]), allow_unused=True)
        _close(grads[0], g[f"{name}/gh"], 1e-4, 1e-5)
        if grads[1] is not None:
            _close(grads[1], g[f"{name}/ge"], 1e-4, 1e-5)
        for k, gr in zip(pnames, grads[2:]):
            ref = g[f"{name}/gp::{k}"]
            _close(gr if gr is not None else np.zeros_like(ref), ref, 1e-4, 2e-5)
        for k, v in stats.items():
            _close(v, g[f"{name}/after::{k}"], 1e-5, 1e-6)
