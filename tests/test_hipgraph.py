"""A training step captured in a HIP graph must reproduce the eager step: every launch of the package goes to the
caller's stream without host synchronisation (the C ABI's contract), which is what makes the capture legal."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("type_net,F_,min_rows", [("towers", 20, None), ("towers", 20, 0), ("simple", 7, None), ("complex", 8, 0)])
def test_captured_step_equals_eager_step(monkeypatch, type_net, F_, min_rows):
    if min_rows is not None:        # small test graphs take the library GEMMs by default: also capture the streaming Linear kernels
        import dgn_amd.ops
        monkeypatch.setattr(dgn_amd.ops, "LINEAR_MIN_ROWS", min_rows)
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.hipgraph import capture
    dev = torch.device("cuda")
    b = synth.molecule_batch(64, seed=5, laplacian_eig=False)
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), b["num_nodes"], eig=b["eig"].to(dev))
    N = graph.num_nodes
    torch.manual_seed(1)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, "mean max std dir1-dx dir1-av", "identity amplification attenuation",
                             {"log": torch.tensor(1.2)}, type_net, True, towers=5, edge_features=False, edge_dim=0).model.to(dev).train()
    gen = torch.Generator(device=dev).manual_seed(2)
    h = torch.randn(N, F_, device=dev, generator=gen).requires_grad_(True)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = b["snorm_n"].to(dev)
    params = list(layer.parameters())
    out = {}

    def step():
        graph._wcache.clear()
        y = layer(graph, h, None, snorm)
        y.backward(ct)
        out["y"] = y.detach()        # (an autograd-attached tensor kept from an earlier step and released inside the capture crashes capture_end)

    def reset():
        h.grad = None
        for p in params:
            p.grad = None

    reset(); step()
    ref_y = out["y"].detach().clone()
    ref_g = [h.grad.clone()] + [p.grad.clone() for p in params]
    bn0 = {k: v.clone() for k, v in layer.state_dict().items() if "running" in k}
    reset()
    g = capture(lambda: (reset(), step()), warmup=2)         # warm-up steps also advance the BN running statistics
    reset_done = [h.grad] + [p.grad for p in params]
    assert all(t is not None for t in reset_done)            # the captured backward produced the gradient tensors
    with torch.no_grad():
        h.add_(0.0)                                          # (inputs are read in place at replay time)
    g.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(out["y"], ref_y, rtol=1e-6, atol=1e-6)
    for a, r in zip([h.grad] + [p.grad for p in params], ref_g):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-6)
    # new input values, same shapes: the replay follows them
    with torch.no_grad():
        h.mul_(0.5)
    g.replay()
    y_graph = out["y"].detach().clone()
    g_graph = h.grad.clone()
    reset(); step()
    torch.testing.assert_close(y_graph, out["y"].detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g_graph, h.grad, rtol=1e-4, atol=1e-5)
    assert bn0 is not None
