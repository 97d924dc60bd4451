"""Fixture G9 (the reference's unmodified ``MoleculeDGL.get_eig``, data/molecules.py:100-116, driven by a fake graph with the
solver call intercepted) pins the Laplacian construction and the eigenvector bookkeeping; the numpy restatement in
oracle/eig_oracle.py and the LOBPCG core of dgn_amd.eig are held to it.  CPU only."""
import numpy as np
import pytest
import torch


def _clusters(w, kk, tol=1e-6):
    j = 0
    while j < kk:
        e = j + 1
        while e < len(w) and abs(w[e] - w[j]) < tol:
            e += 1
        yield j, e
        j = e


@pytest.mark.parametrize("norm", ["none", "sym", "walk"])
def test_laplacian_and_eigvecs_match_the_reference(golden, norm):
    from oracle import eig_oracle
    g = golden("g9_laplacian")
    k = int(g["pos_enc_dim"])
    for i in range(int(g["n_graphs"])):
        src, dst, n = g[f"g{i}/src"], g[f"g{i}/dst"], int(g[f"g{i}/n"])
        L_ref = g[f"g{i}/{norm}/L"]
        np.testing.assert_allclose(eig_oracle.graph_laplacian(src, dst, n, norm), L_ref, rtol=0, atol=1e-12)
        (w, v), = eig_oracle.eigvecs(src, dst, [n], k, norm)
        ref = g[f"g{i}/{norm}/eig"].astype(np.float64)                 # [n, k] fp32, increasing eigenvalue order
        assert ref.shape == (n, k)
        for c in range(k):                                              # every stored column is an eigenvector of L for w[c]
            np.testing.assert_allclose(L_ref @ ref[:, c], w[c] * ref[:, c], atol=2e-5)
        for j, e in _clusters(w, k):                                    # and spans the restatement's subspace
            if e <= k:
                A, B = v[:, j:e], ref[:, j:e]
                # (walk: eigenvectors of a non-symmetric matrix are not orthogonal: compare through least squares)
                coef, res, *_ = np.linalg.lstsq(A, B, rcond=None)
                np.testing.assert_allclose(A @ coef, B, atol=2e-5)


def test_lobpcg_core_on_the_reference_laplacian(golden):
    from dgn_amd.eig import lobpcg_lowest
    g = golden("g9_laplacian")
    i = int(g["n_graphs"]) - 1                                          # the 37-atom graph
    L = torch.from_numpy(g[f"g{i}/none/L"])
    w = torch.linalg.eigvalsh(L)
    lam, X, it, res = lobpcg_lowest(lambda Z: L @ Z, L.shape[0], 4, diag=torch.diagonal(L).clone(), iters=300, tol=1e-7,
                                    generator=torch.Generator().manual_seed(0))
    np.testing.assert_allclose(lam.numpy(), w[:4].numpy(), atol=1e-7)
    assert float(res.max()) < 2e-6 and it < 300
    np.testing.assert_allclose((X.T @ X).numpy(), np.eye(4), atol=1e-8)


def test_distortion_augmentation():
    """train_superpixels_graph_classification.py:43-47 (intended form, see dgn_amd.eig.distort)"""
    from dgn_amd.eig import distort
    gen = torch.Generator().manual_seed(3)
    eig = torch.randn(500, 4, generator=gen)
    out = distort(eig, 0.25, generator=torch.Generator().manual_seed(5))
    dist = (torch.rand(500, generator=torch.Generator().manual_seed(5)) - 0.5) * 2 * 0.25
    torch.testing.assert_close(out[:, 1], dist * eig[:, 1].abs().mean() + eig[:, 1])
    torch.testing.assert_close(out[:, 2], dist * eig[:, 2].abs().mean() + eig[:, 2])
    assert torch.equal(out[:, [0, 3]], eig[:, [0, 3]])
    assert float((out[:, 1] - eig[:, 1]).abs().max()) <= 0.25 * float(eig[:, 1].abs().mean()) + 1e-6
