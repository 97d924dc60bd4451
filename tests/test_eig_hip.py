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
