"""dgn_graph_build / _csc (the batch preparation as a few kernels behind the C ABI) against the torch-op build it
replaces: every array of the CSR and of the transposed view must be identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graphs():
    from dgn_amd import synth
    rng = np.random.default_rng(0)
    b = synth.molecule_batch(300, seed=4, laplacian_eig=False)
    yield "molecules", b["src"], b["dst"], int(b["num_nodes"]), {}
    k = synth.knn_batch(n_graphs=6, seed=2)
    yield "knn", k["src"], k["dst"], int(k["num_nodes"]), {}
    N, E = 500, 9000                                           # multigraph with duplicates, self loops, zero in-degree rows, hubs
    dst = rng.integers(0, N - 7, E)
    dst[rng.random(E) < 0.3] = rng.integers(0, 3, 1)[0]
    yield "hubs", torch.from_numpy(rng.integers(0, N, E)), torch.from_numpy(dst), N, dict(hub_threshold=64, hub_chunk=16)
    yield "edgeless", torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 5, {}


@pytest.mark.parametrize("case", list(_graphs()), ids=lambda c: c[0])
def test_native_build_equals_torch_build(monkeypatch, case):
    import dgn_amd
    import dgn_amd.graph as G
    name, src, dst, N, kw = case
    dev = torch.device("cuda")
    built = {}
    for native in (True, False):
        monkeypatch.setattr(G, "NATIVE_BUILD", native)
        g = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, **kw)
        g.ensure_csc()
        built[native] = g
    a, b = built[True], built[False]
    assert (a.num_nodes, a.num_edges, a.max_in_degree, a.n_hub, a.n_chunks) == (b.num_nodes, b.num_edges, b.max_in_degree, b.n_hub, b.n_chunks)
    for f in ("indptr", "src", "log_deg", "csc_ptr", "csc_pos"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert torch.equal(a.in_degree, b.in_degree)
    if a.num_edges:
        assert torch.equal(a.eid, b.eid)
        assert torch.equal(a.dst_csr.long(), torch.repeat_interleave(torch.arange(N, device=dev), a.in_degree))


def test_native_build_matches_the_layer_results():
    """a layer step on a natively built graph equals the step on the torch-built one (same arrays -> same bits)"""
    import dgn_amd
    import dgn_amd.graph as G
    from dgn_amd import synth
    dev = torch.device("cuda")
    b = synth.molecule_batch(64, seed=8, laplacian_eig=False)
    N = int(b["num_nodes"])
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(20, 20, 0.0, True, True, "mean max dir1-dx dir1-av", "identity amplification attenuation",
                             {"log": torch.tensor(1.0)}, "simple", True).model.to(dev)
    h0 = torch.randn(N, 20, device=dev)
    outs = []
    for native in (True, False):
        G.NATIVE_BUILD = native
        try:
            g = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
        finally:
            G.NATIVE_BUILD = True
        h = h0.clone().requires_grad_(True)
        y = layer(g, h, None, b["snorm_n"].to(dev))
        y.sum().backward()
        outs.append((y.detach(), h.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
