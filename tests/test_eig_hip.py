"""Batched Laplacian eigenvectors on the GPU vs the dense per-graph oracle: eigen-subspaces, residuals, padding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("norm", ["none", "sym"])
def test_laplacian_eigvecs_vs_oracle(norm):
    from dgn_amd import synth
    from dgn_amd.eig import laplacian_eigvecs
    from oracle import eig_oracle
    b = synth.molecule_batch(300, seed=7, laplacian_eig=False)
    sizes = b["sizes"].tolist() + [2, 1]                       # plus two graphs smaller than k
    n0 = int(b["num_nodes"])
    src = torch.cat([b["src"], torch.tensor([n0, n0 + 1])])     # the 2-node graph has one bond, the 1-node graph none
    dst = torch.cat([b["dst"], torch.tensor([n0 + 1, n0])])
    k = 4
    eig = laplacian_eigvecs(src.cuda(), dst.cuda(), sizes, k, norm=norm).cpu().double().numpy()
    assert eig.shape == (n0 + 3, k)
    ref = eig_oracle.eigvecs(src.numpy(), dst.numpy(), sizes, k, norm)
    off = 0
    for n, (w, v) in zip(sizes, ref):
        blk = eig[off:off + n]
        kk = min(k, n)
        assert np.all(blk[:, kk:] == 0)                          # fewer nodes than k: zero columns
        # group the k lowest eigenvalues into clusters (degenerate eigenvalues span a subspace: compare projectors)
        j = 0
        while j < kk:
            e = j + 1
            while e < n and abs(w[e] - w[j]) < 1e-6:
                e += 1
            if e <= kk:                                          # the whole cluster lies inside the first k columns
                P_ref = v[:, j:e] @ v[:, j:e].T
                P = blk[:, j:e] @ blk[:, j:e].T
                np.testing.assert_allclose(P, P_ref, atol=2e-5)
            j = e
        L = eig_oracle.graph_laplacian(src.numpy()[(dst.numpy() >= off) & (dst.numpy() < off + n)] - off,
                                       dst.numpy()[(dst.numpy() >= off) & (dst.numpy() < off + n)] - off, n, norm)
        for c in range(kk):                                      # every column is an eigenvector of its eigenvalue
            np.testing.assert_allclose(L @ blk[:, c], w[c] * blk[:, c], atol=5e-5)
        off += n


def test_augmentations_keep_norms():
    from dgn_amd.eig import flip_sign, rotate
    gen = torch.Generator(device="cuda").manual_seed(0)
    eig = torch.randn(1000, 4, device="cuda", generator=gen)
    f = flip_sign(eig, col=2, generator=gen)
    assert torch.equal(f.abs(), eig.abs()) and torch.equal(f[:, [0, 1, 3]], eig[:, [0, 1, 3]]) and (f[:, 2] != eig[:, 2]).any()
    r = rotate(eig, 30.0, generator=gen)
    torch.testing.assert_close(r[:, 1] ** 2 + r[:, 2] ** 2, eig[:, 1] ** 2 + eig[:, 2] ** 2, rtol=1e-5, atol=1e-5)
    assert torch.equal(r[:, [0, 3]], eig[:, [0, 3]])


def test_distort_on_device():
    from dgn_amd.eig import distort
    gen = torch.Generator(device="cuda").manual_seed(1)
    eig = torch.randn(1000, 4, device="cuda", generator=gen)
    d = distort(eig, 0.3, generator=gen)
    assert torch.equal(d[:, [0, 3]], eig[:, [0, 3]])
    for c in (1, 2):
        assert float((d[:, c] - eig[:, c]).abs().max()) <= 0.3 * float(eig[:, c].abs().mean()) + 1e-6
    assert float((d[:, 1] - eig[:, 1]).abs().max()) > 0


@pytest.mark.parametrize("norm", ["none", "sym", "walk"])
def test_batched_eigvecs_vs_reference_fixture(golden, norm):
    """laplacian_eigvecs on the G9 graphs against what the reference's get_eig stored (data/molecules.py:100-116)."""
    from dgn_amd.eig import laplacian_eigvecs
    g = golden("g9_laplacian")
    k = int(g["pos_enc_dim"])
    srcs, dsts, sizes, off = [], [], [], 0
    for i in range(int(g["n_graphs"])):
        srcs.append(torch.from_numpy(g[f"g{i}/src"]) + off)
        dsts.append(torch.from_numpy(g[f"g{i}/dst"]) + off)
        sizes.append(int(g[f"g{i}/n"]))
        off += sizes[-1]
    eig = laplacian_eigvecs(torch.cat(srcs).cuda(), torch.cat(dsts).cuda(), sizes, k, norm=norm).cpu().double().numpy()
    off = 0
    for i, n in enumerate(sizes):
        L = g[f"g{i}/{norm}/L"]
        ref = g[f"g{i}/{norm}/eig"].astype(np.float64)
        blk = eig[off:off + n]
        w = np.sort(np.linalg.eigvals(L).real)
        for c in range(k):
            np.testing.assert_allclose(L @ blk[:, c], w[c] * blk[:, c], atol=5e-5)       # an eigenvector of the reference's L
        j = 0
        while j < k:                                                                     # same subspaces as the stored columns
            e = j + 1
            while e < n and abs(w[e] - w[j]) < 1e-6:
                e += 1
            if e <= k:
                coef, *_ = np.linalg.lstsq(blk[:, j:e], ref[:, j:e], rcond=None)
                np.testing.assert_allclose(blk[:, j:e] @ coef, ref[:, j:e], atol=5e-5)
            j = e
        off += n


def test_sweep_matvec_is_the_reference_laplacian(golden):
    """L X through the aggregation kernels (sum aggregator forward = A X, its backward = A^T X) against the dense L the
    reference built, on the batched G9 graphs; the transposed product changes nothing on symmetric graphs."""
    import dgn_amd
    from dgn_amd.eig import sweep_laplacian_matvec
    g = golden("g9_laplacian")
    srcs, dsts, sizes, off = [], [], [], 0
    for i in range(int(g["n_graphs"])):
        srcs.append(torch.from_numpy(g[f"g{i}/src"]) + off)
        dsts.append(torch.from_numpy(g[f"g{i}/dst"]) + off)
        sizes.append(int(g[f"g{i}/n"]))
        off += sizes[-1]
    N = off
    graph = dgn_amd.DGNGraph(torch.cat(srcs).cuda(), torch.cat(dsts).cuda(), N)
    X = torch.randn(N, 5, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    for norm in ("none", "sym"):
        Ld = torch.zeros(N, N, dtype=torch.float64)
        o = 0
        for i, n in enumerate(sizes):
            Ld[o:o + n, o:o + n] = torch.from_numpy(g[f"g{i}/{norm}/L"])
            o += n
        want = (Ld @ X.cpu())
        for symmetric in (True, False):
            mv, diag = sweep_laplacian_matvec(graph, norm, symmetric=symmetric)
            torch.testing.assert_close(mv(X).cpu(), want, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(diag.cpu(), torch.diagonal(Ld), rtol=0, atol=1e-12)


def test_lobpcg_on_a_large_graph():
    """k lowest eigenpairs of a 200k-node power-law graph (directed: L = D - (A + A^T)/2, as the batched dense path defines
    it) by LOBPCG on the sweep's products; residuals checked with an independent sparse fp64 product."""
    import dgn_amd
    from dgn_amd import synth
    from dgn_amd.eig import lobpcg_eigvecs
    dev = torch.device("cuda")
    indptr, src, _ = synth.powerlaw_csr(200_000, 4_000_000, dev, seed=5)
    graph = dgn_amd.DGNGraph.from_csr(indptr, src)
    N, k = graph.num_nodes, 4
    eig, lam, it, res = lobpcg_eigvecs(graph, k, iters=120, tol=1e-4, generator=torch.Generator(device=dev).manual_seed(0))
    assert eig.shape == (N, k) and eig.dtype == torch.float32
    dst = torch.repeat_interleave(torch.arange(N, device=dev), graph.in_degree)
    A = torch.sparse_coo_tensor(torch.stack([dst, src.long()]), torch.ones(src.numel(), dtype=torch.float64, device=dev), (N, N)).coalesce()
    deg = graph.in_degree.double().clamp(min=1.0)
    X = eig.double()
    LX = X * deg.unsqueeze(1) - 0.5 * (torch.sparse.mm(A, X) + torch.sparse.mm(A.t(), X))
    lam_max = float(2 * deg.max())
    r = (LX - X * lam.unsqueeze(0)).norm(dim=0)
    assert float(r.max()) <= 1e-3 * lam_max, (r.tolist(), lam.tolist(), it)
    torch.testing.assert_close(X.T @ X, torch.eye(k, dtype=torch.float64, device=dev), rtol=0, atol=1e-5)
    assert bool((lam[1:] >= lam[:-1] - 1e-9).all())
    ritz = torch.diagonal(X.T @ LX)
    torch.testing.assert_close(ritz, lam, rtol=1e-3, atol=1e-3 * lam_max)
