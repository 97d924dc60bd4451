"""Block backward of the sweep (csrc/dgn_agg_block.hpp; DgnGraph.blk_cut / blk_gap): one wave owns a run of whole graphs of the
batch and accumulates d x_src in its LDS rows instead of staging an [E, F] per-edge gradient row.  Checked here:
the closed cuts against a numpy restatement; the gradients against the staged two-phase scatter (same per-edge rows, same summation
order: bit-identical unless d x_in aliases d x_src) on the list / layout of every layer type, with isolated nodes, long rows, ragged
boundaries, several block sizes; run-to-run reproducibility; against the oracle (the reference's autograd through
nets/dgn_layer.py:183-186).  The oracle suites of the other files run with the block backward ON (its default)."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda")


def _cuts_numpy(src, dst, N):
    diff = np.zeros(N + 2, dtype=np.int64)
    a, b = np.minimum(src, dst), np.maximum(src, dst)
    m = a < b
    np.add.at(diff, a[m] + 1, 1)
    np.add.at(diff, b[m] + 1, -1)
    cross = np.cumsum(diff)[: N + 1]
    closed = cross == 0
    last = np.maximum.accumulate(np.where(closed, np.arange(N + 1), 0))
    cuts = np.flatnonzero(closed)
    gap = int(np.diff(cuts).max()) if cuts.size > 1 else 0
    return last, gap


@pytest.mark.parametrize("kind", ["molecules", "chain_across", "isolated"])
def test_closed_cuts_match_numpy(kind):
    import dgn_amd
    from dgn_amd import synth
    if kind == "molecules":
        b = synth.molecule_batch(300, seed=3, laplacian_eig=False)
        src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    elif kind == "chain_across":      # an edge between two neighbouring graphs: that boundary is not a closed cut
        b = synth.molecule_batch(50, seed=4, laplacian_eig=False)
        src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
        n0 = int(b["sizes"][0])
        src = torch.cat([src, torch.tensor([n0 - 1])])
        dst = torch.cat([dst, torch.tensor([n0 + 2])])
    else:                             # isolated nodes: every one of them is a graph of its own; chains of 21 nodes
        N = 500
        src = torch.tensor([i for i in range(100) if i % 21 != 20])
        dst = src + 1
    g = dgn_amd.DGNGraph(src.to(_dev()), dst.to(_dev()), N)
    attached = g.ensure_blocks(True)
    cut, gap = g._blk
    last, gap_ref = _cuts_numpy(src.numpy(), dst.numpy(), N)
    assert gap == gap_ref
    np.testing.assert_array_equal(cut.cpu().numpy(), last)
    from dgn_amd.graph import BLOCK_MAX_GAP
    assert attached == (0 < gap <= BLOCK_MAX_GAP)            # (two molecules joined by an edge can exceed a wave's block)
    if attached:
        assert g.c_graph.blk_gap == gap and g.c_graph.blk_cut == cut.data_ptr()
    assert not g.ensure_blocks(False) and not g.c_graph.blk_cut and g.c_graph.blk_gap == 0


def _case(case):
    import dgn_amd
    from dgn_amd.dgn_layer import X_IN_NAME
    if case == "towers":          # headline list: P|Q messages, tower-major, h_in pass-through block, aux table
        return 70, 5, dgn_amd.make_plan(["mean", "max", "min", "dir1-av", "dir1-dx", X_IN_NAME], ["identity"]), True
    if case == "complex":
        return 70, 1, dgn_amd.make_plan(["mean", "max", "min", "dir1-dx", "dir1-av", X_IN_NAME], ["identity"]), True
    if case == "simple":          # x_src = x_in = h: d x_in joins d x_src in the LDS rows
        return 76, 1, dgn_amd.make_plan(["mean", "dir1-dx-no-abs"], ["identity"]), False
    if case == "simple_hiv":
        return 70, 1, dgn_amd.make_plan(["mean", "max", "min", "dir1-dx", "dir1-av"], ["identity"]), False
    if case == "cifar":
        return 66, 1, dgn_amd.make_plan(["mean", "dir1-dx", "dir2-dx"], ["identity"]), False
    if case == "cifar_complex":
        return 66, 1, dgn_amd.make_plan(["mean", "dir1-dx", "dir2-dx", X_IN_NAME], ["identity"]), True
    raise KeyError(case)


def _grads(graph, plan, F_, T, pair, X, PQ, ct_seed=1):
    from dgn_amd.ops import directional_aggregate
    dev = _dev()
    x = X.to(dev).requires_grad_(True)
    if pair:
        pq = PQ.to(dev).requires_grad_(True)
        y = directional_aggregate(graph, plan, 1.1, x_pair=pq, x_in=x, n_towers=T, tower_major=T > 1)
        leaves = [pq, x]
    else:
        y = directional_aggregate(graph, plan, 1.1, x_src=x, x_in=x)
        leaves = [x]
    ct = torch.randn(y.shape, generator=torch.Generator().manual_seed(ct_seed)).to(dev)
    return y, torch.autograd.grad(y, leaves, ct)


@pytest.mark.parametrize("lds_kb", ["13", "11", "24"])
@pytest.mark.parametrize("case", ["towers", "complex", "simple", "simple_hiv", "cifar", "cifar_complex"])
def test_block_backward_matches_staged(monkeypatch, case, lds_kb):
    import dgn_amd
    monkeypatch.setattr(dgn_amd._lib.options, "blk_lds_kb", int(lds_kb))
    monkeypatch.setattr(dgn_amd._lib.options, "blk_min_nodes", 0)
    run_matches_staged(case)


def run_matches_staged(case):
    import dgn_amd
    from dgn_amd import ops, synth
    dev = _dev()
    b = synth.molecule_batch(257, seed=17, laplacian_eig=False)
    src, dst, N = b["src"], b["dst"], int(b["num_nodes"])
    # two isolated nodes and one node with 9 in-edges from the LAST graph appended (per-row fallback inside the grouped kernel), and a
    # node with 70 in-edges from 7 of them (a multigraph row of more than one slot batch: the emit_range path with the LDS adds)
    last0 = N - int(b["sizes"][-1])
    hub, dense = N + 2, N + 3
    src = torch.cat([src, torch.arange(last0, last0 + 9), torch.arange(last0, last0 + 7).repeat(10)])
    dst = torch.cat([dst, torch.full((9,), hub), torch.full((70,), dense)])
    N = N + 4
    F_, T, plan, pair = _case(case)
    gen = torch.Generator().manual_seed(8)
    eig = torch.randn(N, 4, generator=gen)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    X, PQ = torch.randn(N, F_, generator=gen), torch.randn(N, 2 * F_, generator=gen)
    ops.BLOCK_BACKWARD = False
    y_ref, ref = _grads(graph, plan, F_, T, pair, X, PQ)
    assert not graph.c_graph.blk_cut
    ops.BLOCK_BACKWARD = True
    y, got = _grads(graph, plan, F_, T, pair, X, PQ)
    assert graph.c_graph.blk_cut, "the block description was not attached"
    assert torch.equal(y, y_ref)
    again = _grads(graph, plan, F_, T, pair, X, PQ)[1]
    for a, r, a2 in zip(got, ref, again):
        assert torch.isfinite(a).all()
        assert torch.equal(a, a2), "the block backward is not run-to-run reproducible"
        if pair:
            assert torch.equal(a, r), (case, float((a - r).abs().max()))          # the same adds in the same order
        else:                                                                      # d x_in joins d x_src in another position of the sum
            scale = float(r.abs().max())
            assert float((a - r).abs().max()) <= 2e-6 * scale, (case, float((a - r).abs().max()), scale)


def test_graphs_the_block_backward_does_not_take():
    """k-NN batches (8 in-edges per row, graphs of 85-150 nodes) keep the staged backward: no block description is attached."""
    import dgn_amd
    from dgn_amd import synth
    b = synth.knn_batch(6, seed=9)
    g = dgn_amd.DGNGraph(b["src"].to(_dev()), b["dst"].to(_dev()), int(b["num_nodes"]))
    assert not g.ensure_blocks(True) and not g.c_graph.blk_cut


@pytest.mark.parametrize("kind", ["simple", "towers"])
def test_block_backward_layer_vs_oracle(kind, monkeypatch):
    """Whole layers (one C call per direction) with the block backward on, against the oracle's autograd."""
    import dgn_amd
    from dgn_amd import ops, synth
    from oracle import dgn_oracle as orc
    dev = _dev()
    monkeypatch.setenv("DGN_DC_MIN_NODES", "0")
    monkeypatch.setattr(dgn_amd._lib.options, "blk_min_nodes", 0)
    b = synth.molecule_batch(40, seed=11)
    src, dst, N, eig, snorm = b["src"], b["dst"], int(b["num_nodes"]), b["eig"], b["snorm_n"]
    F_ = 70 if kind == "towers" else 76
    aggs = "mean max min dir1-av dir1-dx" if kind == "towers" else "mean dir1-dx-no-abs"
    scalers = "identity amplification attenuation"
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, aggs, scalers, {"log": torch.tensor(1.2)}, kind, True, towers=5,
                             edge_features=False, edge_dim=0).model
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    cfg = dict(aggregators=aggs, scalers=scalers, avg_log=torch.tensor(1.2), graph_norm=True, batch_norm=True, residual=True, towers=5,
               divide_input=True, edge_features=False)
    h = torch.randn(N, F_, generator=gen)
    ct = torch.randn(N, F_, generator=gen)
    ho = h.clone().requires_grad_(True)
    yo, _ = orc.layer_forward(kind, sd, cfg, src, dst, N, eig, ho, None, snorm, training=True)
    (yo * ct).sum().backward()
    layer = layer.to(dev)
    graph = dgn_amd.DGNGraph(src.to(dev), dst.to(dev), N, eig=eig.to(dev))
    hd = h.to(dev).requires_grad_(True)
    y = layer(graph, hd, None, snorm.to(dev))
    (y * ct.to(dev)).sum().backward()
    assert graph.c_graph.blk_cut, "the layer's backward did not attach the block description"
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(hd.grad.cpu().numpy(), ho.grad.numpy(), rtol=1e-4, atol=1e-4 * float(ho.grad.abs().max()))


# ---- graph backward (csrc/dgn_agg_graph.hpp): a workgroup per graph, coefficient rows in LDS, sources gather their out-edges --------------
@pytest.mark.parametrize("gen,case", [("knn", "cifar"), ("knn", "cifar_complex"), ("sbm", "cifar"), ("sbm", "cifar_complex"), ("knn", "zinc_json")])
def test_graph_backward_matches_staged_and_is_reproducible(monkeypatch, gen, case):
    """k-NN (in-degrees 0 .. ~25, zero in-degree rows) and SBM (~25 - 50 in-edges: several slot batches per source) batches: the
    gradients of the graph backward against the staged two-phase scatter -- the same per-edge rows added in the same (source, slot)
    order; d x_dst is a closed form of the row's coefficients there, hence fp32 rounding, not bits --, taken (the staging kernel is
    never launched: checked through the attached description), and bit-reproducible run to run."""
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.dgn_layer import X_IN_NAME
    dev = _dev()
    b = synth.knn_batch(12, seed=5) if gen == "knn" else synth.sbm_batch(6, seed=5, n_lo=44, n_hi=90)
    N = int(b["num_nodes"])
    if case == "zinc_json":
        F_, T, plan, pair = 46, 1, dgn_amd.make_plan(["mean", "dir1-dx", "dir1-av", X_IN_NAME], ["identity"]), True
    else:
        F_, T, plan, pair = _case(case)
    gg = torch.Generator().manual_seed(9)
    X, PQ = torch.randn(N, F_, generator=gg), torch.randn(N, 2 * F_, generator=gg)

    def run(attach):
        graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
        if not attach:
            monkeypatch.setattr(dgn_amd.graph.DGNGraph, "_ensure_graph_blocks", lambda self, enabled=True: False)
        y, g = _grads(graph, plan, F_, T, pair, X, PQ)
        monkeypatch.undo()
        return graph, y, g

    graph, y1, g1 = run(True)
    assert graph.c_graph.gblk_desc and graph.c_graph.n_gblk == len(b["sizes"]), "the graph description was not attached"
    _, y1b, g1b = run(True)
    assert all(torch.equal(a, c) for a, c in zip(g1, g1b)), "the graph backward is not run-to-run reproducible"
    graph0, y0, g0 = run(False)
    assert not graph0.c_graph.gblk_desc
    assert torch.equal(y1, y0)
    for a, c in zip(g1, g0):
        scale = max(1.0, float(c.abs().max()))
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)


@pytest.mark.parametrize("gen,case", [("knn", "cifar"), ("sbm", "cifar_complex")])
def test_graph_backward_feature_tiles_agree_bitwise(monkeypatch, gen, case):
    """The graph backward's feature tiles (blockIdx.y: 1, 2, 3 or 4 column ranges per graph, chosen by batch size and LDS) only decide
    which workgroup owns a column: every tiling gives the same bits."""
    import dgn_amd
    from dgn_amd import _lib, synth
    dev = _dev()
    b = synth.knn_batch(12, seed=5) if gen == "knn" else synth.sbm_batch(6, seed=5, n_lo=44, n_hi=90)
    N = int(b["num_nodes"])
    F_, T, plan, pair = _case(case)
    gg = torch.Generator().manual_seed(9)
    X, PQ = torch.randn(N, F_, generator=gg), torch.randn(N, 2 * F_, generator=gg)
    ref = None
    for tiles in (1, 2, 3, 4):
        monkeypatch.setattr(_lib.options, "graph_bwd_tiles", tiles)
        graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
        _, g = _grads(graph, plan, F_, T, pair, X, PQ)
        assert graph.c_graph.gblk_desc
        if ref is None:
            ref = g
        else:
            assert all(torch.equal(a, c) for a, c in zip(g, ref)), f"tiles = {tiles} changed the gradients"


_GRAPH_BWD_LISTS = [(["mean", "dir1-dx", "dir2-dx"], False), (["mean", "dir1-dx", "dir2-dx"], True), (["mean", "dir1-dx", "dir1-av"], False),
                    (["mean", "dir1-dx", "dir1-av"], True), (["mean", "dir1-av", "dir1-dx"], False), (["mean", "dir1-dx-no-abs"], False), (["mean"], False),
                    (["mean", "dir1-dx"], False), (["mean", "dir1-av"], False)]


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DGN_FUZZ_CASES", "24"))))
def test_graph_backward_fuzz_vs_staged(monkeypatch, seed):
    """Seeded shapes for the graph backward (csrc/dgn_agg_graph.hpp): every baked-in list without max / min / std / var, simple and pair
    messages, even widths 2 .. 126 (one to 63 feature pairs: every tiling), k-NN graphs of 9 .. 150 nodes with 2 .. 12 neighbours and SBM
    graphs, 1 .. 40 graphs (feature tiles on and off) -- against the staged scatter, and reproducible."""
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.dgn_layer import X_IN_NAME
    rng = np.random.default_rng(7000 + seed)
    names, pair = _GRAPH_BWD_LISTS[int(rng.integers(0, len(_GRAPH_BWD_LISTS)))]
    F_ = 2 * int(rng.integers(1, 64))
    n_graphs = int(rng.integers(1, 41))
    if rng.integers(0, 3) == 0:
        b = synth.sbm_batch(max(1, n_graphs // 6), seed=int(rng.integers(0, 1 << 20)), n_lo=20, n_hi=int(rng.integers(30, 120)))
    else:
        n_lo = int(rng.integers(9, 60))
        b = synth.knn_batch(n_graphs, seed=int(rng.integers(0, 1 << 20)), n_lo=n_lo, n_hi=n_lo + int(rng.integers(0, 90)), k=int(rng.integers(4, 13)))
    N = int(b["num_nodes"])
    if b["src"].dim() != 1 or b["src"].shape != b["dst"].shape or b["src"].numel() <= 3 * N:
        pytest.skip("fewer than three edges per node (or a degenerate synthetic batch): not a graph-backward batch")
    plan = dgn_amd.make_plan(names + ([X_IN_NAME] if pair else []), ["identity"])
    dev = _dev()
    gg = torch.Generator().manual_seed(seed)
    X, PQ = torch.randn(N, F_, generator=gg), torch.randn(N, 2 * F_, generator=gg)
    eig = b["eig"] if b["eig"].shape[1] >= 3 else torch.cat([b["eig"], torch.randn(N, 3 - b["eig"].shape[1], generator=gg)], dim=1)

    def run(attach):
        graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=eig.to(dev))
        if not attach:
            monkeypatch.setattr(dgn_amd.graph.DGNGraph, "_ensure_graph_blocks", lambda self, enabled=True: False)
        y, g = _grads(graph, plan, F_, 1, pair, X, PQ)
        monkeypatch.undo()
        return graph, y, g

    graph, y1, g1 = run(True)
    if not graph.c_graph.gblk_desc:
        pytest.skip("the graph description was not attached (graphs beyond 512 nodes)")
    _, _, g1b = run(True)
    assert all(torch.equal(a, c) for a, c in zip(g1, g1b)), "not run-to-run reproducible"
    _, y0, g0 = run(False)
    assert torch.equal(y1, y0)
    for a, c in zip(g1, g0):
        scale = max(1.0, float(c.abs().max()))
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=2e-5, atol=4e-6 * scale, err_msg=f"{names} pair={pair} F={F_} N={N}")
