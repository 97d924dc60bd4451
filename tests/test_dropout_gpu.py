"""Dropout of the layer tails (F.dropout at nets/dgn_layer.py:130, :201; configs HIV / PCBA / CIFAR10 ship dropout 0.3) on the bit-mask
kernels (dgn_dropout_forward / _backward), and the whole-layer fast path with dropout on:
(a) the saved keep bits reproduce the output and the gradient exactly, and a simple / complex layer with dropout 0.3 equals the oracle
    fed the SAME mask (values, d h, parameter gradients); the same seed gives the same mask;
(b) keep rate, scaling and independence statistically."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(mask, n):
    m = mask.cpu().numpy()
    return torch.from_numpy(np.unpackbits(m, bitorder="little")[:n].astype(np.float32))


@pytest.mark.parametrize("shape", [(1000, 70), (37, 75), (5, 3), (8192, 128)])
def test_dropout_kernel_mask_scaling_and_backward(shape):
    from dgn_amd import ops
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(shape, generator=gen).to(dev).requires_grad_(True)
    seed = torch.tensor([12345], dtype=torch.int64, device=dev)
    p = 0.3
    y = ops.dropout(x, p, True, seed=seed)
    keep = _bits(ops.LAST_DROPOUT_MASK, x.numel()).reshape(shape).to(dev)
    assert torch.equal(y.detach(), torch.where(keep > 0, x.detach() * (1.0 / (1.0 - p)), torch.zeros_like(x)))
    ct = torch.randn(shape, generator=gen).to(dev)
    (g,) = torch.autograd.grad(y, x, ct)
    assert torch.equal(g, torch.where(keep > 0, ct * (1.0 / (1.0 - p)), torch.zeros_like(ct)))
    # the same (seed, offset): the same mask; another offset or seed: another one
    y2 = ops.dropout(x, p, True, seed=seed)
    assert torch.equal(y2, y)
    if x.numel() > 500:
        assert not torch.equal(ops.dropout(x, p, True, seed=seed, offset=1), y)
        assert not torch.equal(ops.dropout(x, p, True, seed=seed + 1), y)
    # eval mode / p = 0: the identity
    assert ops.dropout(x, p, False) is x and ops.dropout(x, 0.0, True) is x


def test_dropout_statistics():
    from dgn_amd import ops
    dev = torch.device("cuda")
    N, F_ = 200000, 70
    x = torch.ones(N, F_, device=dev)
    torch.manual_seed(7)
    for p in (0.1, 0.3, 0.5):
        y = ops.dropout(x, p, True)
        keep = (y > 0).float()
        n = N * F_
        rate = float(keep.mean())
        assert abs(rate - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5, (p, rate)
        assert float(y.max()) == pytest.approx(1 / (1 - p), rel=1e-6)
        assert abs(float(y.mean()) - 1.0) < 5e-3                                       # E[dropout(x)] = x
        col = keep.mean(0)
        assert float((col - (1 - p)).abs().max()) < 6 * (p * (1 - p) / N) ** 0.5          # every column at the rate
        flat = keep.flatten() - (1 - p)
        for lag in (1, 7, 8, 70):                                                      # neighbours within a byte, across bytes, across rows
            c = float((flat[:-lag] * flat[lag:]).mean()) / (p * (1 - p))
            assert abs(c) < 5 / n ** 0.5, (p, lag, c)
    # torch.manual_seed reproduces the draw
    torch.manual_seed(3)
    a = ops.dropout(x, 0.3, True)
    torch.manual_seed(3)
    b = ops.dropout(x, 0.3, True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("route", ["whole-layer", "graph-block"])
@pytest.mark.parametrize("kind,F_,aggs", [("simple", 70, "mean max min dir1-dx dir1-av"), ("complex", 65, "mean dir1-dx dir2-dx"),
                                          ("simple", 65, "mean dir1-dx dir2-dx")])
def test_layer_with_dropout_vs_oracle_with_the_same_mask(kind, F_, aggs, route, monkeypatch):
    """HIV json (simple, hidden 70, 3 scalers, dropout 0.3, no graph norm) and CIFAR10 json (hidden 65, one scaler, dropout 0.3) through
    the WHOLE-LAYER calls (dropout = ops.dropout behind the call) and -- round 6, last session -- on the GRAPH-BLOCK route, where the layer's final
    F.dropout (nets/dgn_layer.py:130, :201) runs inside the tail kernels (DgnBlockLayer.drop_* for types 0 / 1): against the oracle fed the keep
    mask the kernel drew; on the block route the mask must also be, bit for bit, what dgn_dropout_forward draws for the same key and offset."""
    import dgn_amd
    from dgn_amd import ops, synth
    from oracle import dgn_oracle as orc
    dev = torch.device("cuda")
    monkeypatch.setattr(ops, "WIDE_MIN_ROWS", 0)
    monkeypatch.setenv("DGN_DC_MIN_NODES", "0")
    # (the graph-block route's LDS plan does not hold k-NN graphs of hidden 65: its variants run the same layers on a molecule batch)
    b = synth.molecule_batch(30, seed=5) if (F_ == 70 or route == "graph-block") else synth.knn_batch(4, seed=5)
    src, dst, N, eig, snorm = b["src"], b["dst"], int(b["num_nodes"]), b["eig"], b["snorm_n"]
    scalers = "identity amplification attenuation" if F_ == 70 else "identity"
    p = 0.3
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, p, F_ != 70, True, aggs, scalers, {"log": torch.tensor(1.2)}, kind, True, towers=1,
                             edge_features=False, edge_dim=0).model
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for q in layer.parameters():
            if q.dim() == 2:
                q.copy_(torch.randn(q.shape, generator=gen) / q.shape[1] ** 0.5)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    h = torch.randn(N, F_, generator=gen)
    ct = torch.randn(N, F_, generator=gen)
    layer = layer.to(dev).train()
    calls = []
    if route == "whole-layer":
        whole = type(layer)._whole_layer
        monkeypatch.setattr(type(layer), "_whole_layer", lambda self, *a: calls.append(whole(self, *a)) or calls[-1])
        monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_NODES", 0)
    else:
        real = ops.block_layer
        monkeypatch.setattr(ops, "block_layer", lambda *a, **k: calls.append(real(*a, **k)) or calls[-1])
        monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_NODES", 8192)
        monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_POST", 1 << 30)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    hd = h.to(dev).requires_grad_(True)
    ops.LAST_DROPOUT_MASK = None
    y = layer(graph, hd, None, snorm.to(dev))
    assert calls and calls[-1] is not None, f"the {route} call did not take the layer with dropout on"
    mask_bytes = ops.LAST_DROPOUT_MASK.clone()
    keep = _bits(mask_bytes, N * F_).reshape(N, F_)
    assert 0.6 < float(keep.mean()) < 0.8
    if route == "graph-block":      # the tail draws the bits dgn_dropout_forward draws for (key, offset) over the dense [N, f_out] tensor
        key, n_calls = dgn_amd.dgn_layer._DROP_STATE[torch.device("cuda", torch.cuda.current_device())]
        ops.dropout(torch.ones(N, F_, device=dev), p, True, seed=key, offset=n_calls)
        assert torch.equal(ops.LAST_DROPOUT_MASK[:(N * F_ + 7) // 8], mask_bytes[:(N * F_ + 7) // 8])
    (y * ct.to(dev)).sum().backward()
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(1.2), graph_norm=F_ != 70, batch_norm=True, residual=True, towers=1,
               divide_input=True, edge_features=False)
    res = {}
    for dt in (torch.float32, torch.float64):
        sdt = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
        ho = h.to(dt).requires_grad_(True)
        leaves = [ho] + [v.requires_grad_(True) for k, v in sdt.items() if v.is_floating_point() and "running" not in k]
        yo, _ = orc.layer_forward(kind, sdt, dict(cfg, avg_log=cfg["avg_log"].to(dt)), src, dst, N, eig.to(dt), ho, None, snorm.to(dt),
                                  training=True, dropout=(p, keep.to(dt)))
        res[dt] = (yo.detach(), torch.autograd.grad((yo * ct.to(dt)).sum(), leaves))
    yo32, go32 = res[torch.float32]
    yo64, go64 = res[torch.float64]

    def close(a, r32, r64, tol, what):
        a = a.detach().cpu().double()
        err = (a - r64).abs()
        bound = tol * (1.0 + r64.abs()) + 4 * (r32.double() - r64).abs()
        assert bool((err <= bound).all()), (what, float(err.max()))

    close(y, yo32, yo64, 2e-5, "y")
    assert torch.equal((y.detach().cpu() == 0), (keep == 0) | (yo32 == 0)), "dropped entries differ from the mask"
    close(hd.grad, go32[0], go64[0], 1e-4 * float(go64[0].abs().max()), "d h")
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    params = dict(layer.named_parameters())
    for i, k in enumerate(names):
        close(params[k].grad, go32[1 + i], go64[1 + i], 1e-4 * max(1.0, float(go64[1 + i].abs().max())), k)


def test_captured_dropout_draws_new_masks_per_replay():
    """A HIP graph freezes host-side arguments: the layers' dropout therefore advances its Philox key ON THE DEVICE inside the captured
    region, so that replays do not repeat one mask."""
    import dgn_amd
    from dgn_amd import ops
    from dgn_amd.dgn_layer import _dropout
    dev = torch.device("cuda")
    dgn_amd.reset_dropout_state()
    x = torch.ones(4096, 64, device=dev)
    _dropout(x, 0.3, True)                              # (warm-up: creates the per-device key outside the capture)
    out = torch.empty_like(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out.copy_(_dropout(x, 0.3, True))
    torch.cuda.current_stream().wait_stream(s)
    masks = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        masks.append((out > 0).clone())
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    for m in masks:
        assert abs(float(m.float().mean()) - 0.7) < 0.01


def test_dropout_state_get_set_replays_masks():
    """The dropout stream lives outside torch's RNG state: (key, calls) saved with ``get_dropout_state`` and installed with
    ``set_dropout_state`` replays the masks from that point on -- what a checkpoint resume or an activation recompute needs."""
    import dgn_amd
    from dgn_amd.dgn_layer import _dropout
    dev = torch.device("cuda")
    dgn_amd.reset_dropout_state()
    assert dgn_amd.get_dropout_state(dev) is None
    x = torch.ones(2048, 64, device=dev)
    _dropout(x, 0.3, True)
    key, calls = dgn_amd.get_dropout_state(dev)
    assert calls == 1
    a = [_dropout(x, 0.3, True).clone() for _ in range(3)]
    assert not torch.equal(a[0], a[1])
    dgn_amd.set_dropout_state(dev, key, calls)
    b = [_dropout(x, 0.3, True).clone() for _ in range(3)]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert dgn_amd.get_dropout_state(dev) == (key, calls + 3)


def test_first_dropout_inside_a_capture_is_refused():
    import dgn_amd
    from dgn_amd.dgn_layer import _dropout
    dev = torch.device("cuda")
    dgn_amd.reset_dropout_state()
    x = torch.ones(1024, 64, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    raised = False
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                _dropout(x, 0.3, True)
        except RuntimeError as exc:
            raised = "outside a stream capture" in str(exc)
    torch.cuda.current_stream().wait_stream(s)
    assert raised
    dgn_amd.reset_dropout_state()


@pytest.mark.parametrize("route", ["whole-layer", "graph-block"])
def test_towers_layer_with_dropout_in_the_whole_layer_call_vs_oracle_with_the_same_mask(monkeypatch, route):
    """``DGNTower``'s F.dropout between BatchNorm and the mixing network (nets/dgn_layer.py:275) inside dgn_towers_layer_forward /
    _backward (DgnTowersLayer.drop_*) and -- round 6 -- inside the graph-block route's tail kernels (DgnBlockLayer.drop_*): values, d h
    and every parameter gradient against the oracle fed the keep mask the kernel drew; on the block route the mask must also be, bit for
    bit, what dgn_dropout_forward draws for the same key and offset."""
    import dgn_amd
    from dgn_amd import ops, synth
    from oracle import dgn_oracle as orc
    dev = torch.device("cuda")
    b = synth.molecule_batch(40, seed=9)
    src, dst, N, eig, snorm = b["src"], b["dst"], int(b["num_nodes"]), b["eig"], b["snorm_n"]
    F_, p = 70, 0.3
    aggs, scalers = "mean max min dir1-av dir1-dx", "identity amplification attenuation"
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, p, True, True, aggs, scalers, {"log": torch.tensor(1.2)}, "towers", True, towers=5, edge_features=False,
                             edge_dim=0).model
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for q in layer.parameters():
            if q.dim() == 2:
                q.copy_(torch.randn(q.shape, generator=gen) / q.shape[1] ** 0.5)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    h = torch.randn(N, F_, generator=gen)
    ct = torch.randn(N, F_, generator=gen)
    layer = layer.to(dev).train()
    calls = []
    if route == "whole-layer":
        whole = type(layer)._whole_layer
        monkeypatch.setattr(type(layer), "_whole_layer", lambda self, *a: calls.append(whole(self, *a)) or calls[-1])
        monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_NODES", 0)
    else:
        real = ops.block_layer
        monkeypatch.setattr(ops, "block_layer", lambda *a, **k: calls.append(real(*a, **k)) or calls[-1])
        monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_NODES", 8192)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    hd = h.to(dev).requires_grad_(True)
    ops.LAST_DROPOUT_MASK = None
    y = layer(graph, hd, None, snorm.to(dev))
    assert calls and calls[-1] is not None, f"the {route} call did not take the towers layer with dropout on"
    mask_bytes = ops.LAST_DROPOUT_MASK.clone()
    keep = _bits(mask_bytes, N * F_).reshape(N, F_)
    assert 0.6 < float(keep.mean()) < 0.8
    if route == "graph-block":      # the tails draw the bits dgn_dropout_forward draws for (key, offset) over the dense [N, T f_out] tensor
        key, n_calls = dgn_amd.dgn_layer._DROP_STATE[torch.device("cuda", torch.cuda.current_device())]
        ops.dropout(torch.ones(N, F_, device=dev), p, True, seed=key, offset=n_calls)
        assert torch.equal(ops.LAST_DROPOUT_MASK[:(N * F_ + 7) // 8], mask_bytes[:(N * F_ + 7) // 8])
    (y * ct.to(dev)).sum().backward()
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(1.2), graph_norm=True, batch_norm=True, residual=True, towers=5,
               divide_input=True, edge_features=False)
    res = {}
    for dt in (torch.float32, torch.float64):
        sdt = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
        ho = h.to(dt).requires_grad_(True)
        leaves = [ho] + [v.requires_grad_(True) for k, v in sdt.items() if v.is_floating_point() and "running" not in k]
        yo, _ = orc.layer_forward("towers", sdt, dict(cfg, avg_log=cfg["avg_log"].to(dt)), src, dst, N, eig.to(dt), ho, None, snorm.to(dt),
                                  training=True, dropout=(p, keep.to(dt)))
        res[dt] = (yo.detach(), torch.autograd.grad((yo * ct.to(dt)).sum(), leaves))
    (yo32, go32), (yo64, go64) = res[torch.float32], res[torch.float64]

    def close(a, r32, r64, tol, what):
        a = a.detach().cpu().double()
        err = (a - r64).abs()
        bound = tol * (1.0 + r64.abs()) + 4 * (r32.double() - r64).abs()
        assert float((err > bound).float().mean()) <= 2e-4, (what, float(err.max()))      # (a LeakyReLU kink within rounding flips a slope)

    close(y, yo32, yo64, 2e-5, "y")
    close(hd.grad, go32[0], go64[0], 1e-4 * float(go64[0].abs().max()), "d h")
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    params = dict(layer.named_parameters())
    for i, k in enumerate(names):
        close(params[k].grad, go32[1 + i], go64[1 + i], 1e-4 * max(1.0, float(go64[1 + i].abs().max())), k)
    # a second step draws another mask; evaluation mode applies none
    y2 = layer(graph, hd, None, snorm.to(dev))
    assert not torch.equal(_bits(ops.LAST_DROPOUT_MASK, N * F_).reshape(N, F_), keep)
    layer.eval()
    with torch.no_grad():
        ye, ye2 = layer(graph, hd, None, snorm.to(dev)), layer(graph, hd, None, snorm.to(dev))
    assert torch.equal(ye, ye2) and y2.shape == ye.shape
