"""Batches held at a fixed capacity (DGNGraph.padded / hipgraph.PaddedBatch): the padded layer step -- eager and as ONE captured HIP
graph replayed for batches of different sizes -- against the ordinary step on the exact-size graph: outputs of the real rows, input
and parameter gradients, BatchNorm running statistics."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches():
    from dgn_amd import synth
    return [synth.molecule_batch(n, seed=s, laplacian_eig=False) for n, s in ((60, 1), (75, 2), (48, 3), (70, 4))]


def _reference_step(layer, b, h, ct, dev):
    import dgn_amd
    N = int(b["num_nodes"])
    g = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
    hh = h.clone().requires_grad_(True)
    y = layer(g, hh, None, b["snorm_n"].to(dev))
    y.backward(ct)
    return y.detach(), hh.grad, {k: v.grad.clone() for k, v in layer.named_parameters()}


@pytest.mark.parametrize("type_net", ["towers", "simple"])
def test_padded_step_equals_exact_step(type_net):
    import dgn_amd
    from dgn_amd.hipgraph import PaddedBatch, capture
    dev = torch.device("cuda")
    bs = _batches()
    F_ = 70 if type_net == "towers" else 20
    n_cap = int(max(int(b["num_nodes"]) for b in bs) * 1.1) + 7
    e_cap = int(max(b["src"].numel() for b in bs) * 1.1) + 5
    torch.manual_seed(0)
    proto = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, "mean max min dir1-av dir1-dx", "identity amplification attenuation",
                             {"log": torch.tensor(1.1)}, type_net, True, towers=5, edge_features=False, edge_dim=0).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for p in proto.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=gen))
    hs = [torch.randn(int(b["num_nodes"]), F_, device=dev, generator=gen) for b in bs]
    cts = [torch.randn(int(b["num_nodes"]), F_, device=dev, generator=gen) for b in bs]

    ref_layer = copy.deepcopy(proto).train()
    refs = []
    for b, h, ct in zip(bs, hs, cts):
        for p in ref_layer.parameters():
            p.grad = None
        refs.append(_reference_step(ref_layer, b, h, ct, dev))
    ref_stats = {k: v.clone() for k, v in ref_layer.state_dict().items() if "running" in k or "num_batches" in k}

    for mode in ("eager", "captured"):
        layer = copy.deepcopy(proto).train()
        pb = PaddedBatch(n_cap, e_cap, dev, eig_dim=bs[0]["eig"].shape[1])
        h_buf = pb.add_node_tensor("h", F_, requires_grad=True)
        sn_buf = pb.add_node_tensor("snorm", 1)
        ct_buf = pb.add_node_tensor("ct", F_)
        out = {}

        def step():
            pb.graph.invalidate_caches()
            y = layer(pb.graph, h_buf, None, sn_buf)
            y.backward(ct_buf)
            out["y"] = y.detach()

        def load(i):
            b = bs[i]
            pb.load(b["src"].to(dev), b["dst"].to(dev), int(b["num_nodes"]), b["eig"].to(dev),
                    node=dict(h=hs[i], snorm=b["snorm_n"].to(dev), ct=cts[i]))

        graph = None
        if mode == "captured":
            load(0)
            # warm-up steps would advance the running statistics: snapshot / restore around the capture
            snap = copy.deepcopy(layer.state_dict())

            def reset():
                h_buf.grad = None
                for p in layer.parameters():
                    p.grad = None

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    reset()
                    step()
            torch.cuda.current_stream().wait_stream(side)
            reset()                                    # gradients None at capture: every replay then DEFINES them
            out.clear()
            graph = capture(step, warmup=0)
            layer.load_state_dict(snap)
        for i, b in enumerate(bs):
            N = int(b["num_nodes"])
            load(i)
            if graph is None:
                h_buf.grad = None
                for p in layer.parameters():
                    p.grad = None
                step()
            else:
                graph.replay()
            torch.cuda.synchronize()
            y_ref, gh_ref, gp_ref = refs[i]
            np.testing.assert_allclose(out["y"][:N].cpu().numpy(), y_ref.cpu().numpy(), rtol=2e-5, atol=2e-5, err_msg=f"{mode} batch {i} y")
            np.testing.assert_allclose(h_buf.grad[:N].cpu().numpy(), gh_ref.cpu().numpy(), rtol=1e-4, atol=2e-5, err_msg=f"{mode} batch {i} gh")
            for k, p in layer.named_parameters():
                r = gp_ref[k]
                np.testing.assert_allclose(p.grad.cpu().numpy(), r.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(r.abs().max())),
                                           err_msg=f"{mode} batch {i} grad {k}")
        for k, v in layer.state_dict().items():
            if "running" in k:
                np.testing.assert_allclose(v.cpu().numpy(), ref_stats[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=f"{mode} {k}")


@pytest.mark.parametrize("type_net,as_types", [("towers", False), ("complex", False), ("towers", True)])
def test_padded_step_with_edge_features_equals_exact_step(type_net, as_types):
    """edge features on a padded graph.  The per-edge tensors are e_cap rows long while the batch fills E slots:
    the untouched gradient rows must read as zero and the slot -> edge map of the tail must not carry the previous batch's ids.
    Eager padded steps over batches of different sizes against the exact-size graph: output, d h, d ef, every parameter gradient."""
    import dgn_amd
    from dgn_amd.hipgraph import PaddedBatch
    dev = torch.device("cuda")
    bs = _batches()
    F_, ed, K = 20, 6, 4
    n_cap = int(max(int(b["num_nodes"]) for b in bs) * 1.1) + 7
    e_cap = int(max(b["src"].numel() for b in bs) * 1.1) + 5
    torch.manual_seed(0)
    proto = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, "mean max min dir1-av dir1-dx", "identity amplification attenuation",
                             {"log": torch.tensor(1.1)}, type_net, True, towers=5, edge_features=True, edge_dim=ed).model.to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for p in proto.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=gen))
    table = torch.randn(K, ed, device=dev, generator=gen)
    layer_ref, layer_pad = copy.deepcopy(proto).train(), copy.deepcopy(proto).train()
    pb = PaddedBatch(n_cap, e_cap, dev, eig_dim=bs[0]["eig"].shape[1])
    h_buf, sn_buf, ct_buf = pb.add_node_tensor("h", F_, requires_grad=True), pb.add_node_tensor("snorm", 1), pb.add_node_tensor("ct", F_)
    ef_buf = pb.add_edge_tensor("ef", ed, requires_grad=True)
    ty_buf = torch.zeros(e_cap, dtype=torch.int64, device=dev)
    for order in (0, 1, 2, 3, 1):              # (a smaller batch after a larger one leaves a stale tail behind)
        b = bs[order]
        N, E = int(b["num_nodes"]), b["src"].numel()
        h = torch.randn(N, F_, device=dev, generator=gen)
        ct = torch.randn(N, F_, device=dev, generator=gen)
        ef = torch.randn(E, ed, device=dev, generator=gen)
        types = torch.randint(0, K, (E,), device=dev, generator=gen)
        src, dst, sn = b["src"].to(dev), b["dst"].to(dev), b["snorm_n"].to(dev)
        # exact-size reference
        for p in layer_ref.parameters():
            p.grad = None
        g = dgn_amd.DGNGraph(src, dst, N, eig=b["eig"].to(dev))
        hh, ee, tt = h.clone().requires_grad_(True), ef.clone().requires_grad_(True), table.clone().requires_grad_(True)
        e_in = dgn_amd.EdgeTypeFeatures(tt, types) if as_types else ee
        y_ref = layer_ref(g, hh, e_in, sn)
        y_ref.backward(ct)
        # padded
        for p in layer_pad.parameters():
            p.grad = None
        h_buf.grad = ef_buf.grad = None
        pb.load(src, dst, N, b["eig"].to(dev), node=dict(h=h, snorm=sn, ct=ct), edge=dict(ef=ef))
        ty_buf[:E].copy_(types)
        ty_buf[E:].zero_()
        tp = table.clone().requires_grad_(True)
        pb.graph.invalidate_caches()
        y = layer_pad(pb.graph, h_buf, dgn_amd.EdgeTypeFeatures(tp, ty_buf) if as_types else ef_buf, sn_buf)
        y.backward(ct_buf)
        torch.cuda.synchronize()
        close = lambda a, r, m: np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(r.abs().max())), err_msg=m)
        close(y.detach()[:N], y_ref.detach(), f"batch {order} y")
        close(h_buf.grad[:N], hh.grad, f"batch {order} d h")
        if as_types:
            close(tp.grad, tt.grad, f"batch {order} d table")
        else:
            close(ef_buf.grad[:E], ee.grad, f"batch {order} d ef")
            assert float(ef_buf.grad[E:].abs().max()) == 0.0
        for (k, p), q in zip(layer_pad.named_parameters(), layer_ref.parameters()):
            assert torch.isfinite(p.grad).all(), k
            close(p.grad, q.grad, f"batch {order} grad {k}")


def test_rebuild_has_no_host_sync_and_reports_hub_rows_one_load_late():
    """DGNGraph.rebuild leaves the batch's statistics in pinned memory: a batch with a row beyond hub_threshold is computed correctly (a
    padded graph has no hub tables: the row kernels take it) and reported by check_deferred() / the next load."""
    import dgn_amd
    from dgn_amd import _lib
    from dgn_amd.ops import directional_aggregate
    dev = torch.device("cuda")
    n_hub_edges = dgn_amd.graph.HUB_THRESHOLD + 50
    N = 300
    gen = torch.Generator().manual_seed(0)
    src = torch.cat([torch.randint(0, N, (n_hub_edges,), generator=gen), torch.randint(0, N, (400,), generator=gen)])
    dst = torch.cat([torch.zeros(n_hub_edges, dtype=torch.long), torch.randint(1, N, (400,), generator=gen)])
    eig = torch.randn(N, 3, generator=gen)
    g = dgn_amd.DGNGraph.padded(N + 20, src.numel() + 64, dev, eig_dim=3)
    g.rebuild(src, dst, N, eig=eig.to(dev))                       # no exception here
    plan = dgn_amd.make_plan(["mean", "max", "dir1-dx"], ["identity"])
    x = torch.randn(N + 20, 6, generator=gen).to(dev)
    x[N:] = 0
    y = directional_aggregate(g, plan, 1.0, x_src=x, x_in=x)
    exact = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    assert exact.n_hub == 1
    y_ref = directional_aggregate(exact, plan, 1.0, x_src=x[:N].contiguous(), x_in=x[:N].contiguous())
    assert torch.allclose(y[:N], y_ref, rtol=1e-5, atol=1e-5)
    with pytest.raises(_lib.DgnError, match="hub rows"):
        g.check_deferred()
    g.check_deferred()                                               # reported once
    small = torch.randint(0, N, (200,), generator=gen)
    g.rebuild(small, small.roll(1), N, eig=eig.to(dev))
    g.check_deferred()
    assert g.max_in_degree > 0
