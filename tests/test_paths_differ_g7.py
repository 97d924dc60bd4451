"""SURVEY.md section 8(c), fixture row G7: the dense path (models/pytorch) and the DGL path (realworld_benchmark/nets)
are DIFFERENT functions of the same graph -- acos-transformed field, epsilon inside the difference, a diagonal term
and no absolute value on the dense side.  The two implementations must therefore never be unified; this test keeps
the evidence executable: both oracles (each pinned to the reference by its own fixtures) on one graph."""
import torch

from oracle import dense_oracle, dgn_oracle


def test_dense_and_dgl_directional_derivatives_are_different_functions():
    gen = torch.Generator().manual_seed(0)
    N, F_ = 9, 4
    und = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0), (2, 6), (6, 7), (7, 8), (8, 3)]
    adj = torch.zeros(1, N, N)
    for a, b in und:
        adj[0, a, b] = adj[0, b, a] = 1.0
    h = torch.randn(N, F_, generator=gen)
    eig = torch.randn(N, 3, generator=gen)
    # DGL path: messages h[src] reduced at dst, `dir1-dx-no-abs` (signed, like the dense derivative)
    dst, src = adj[0].nonzero(as_tuple=True)
    sparse = dgn_oracle.aggregate_graph(src, dst, N, h[src], eig, h, ["dir1-dx-no-abs"], ["identity"], torch.tensor(1.0))
    # dense path: X[b, i, j] = h_j, `dir1-dx`
    X = h.unsqueeze(0).unsqueeze(1).expand(1, N, N, F_)
    dense = dense_oracle.aggregate("dir1-dx", X, adj, eigvec=eig.unsqueeze(0))[0]
    assert sparse.shape == dense.shape == (N, F_)
    rel = (sparse - dense).abs().max() / sparse.abs().max()
    assert rel > 1e-2, "the two paths agree: they should not (acos field, epsilon placement, diagonal term)"
    # what they do share: both are linear in h and vanish on constant features
    ones = torch.ones(N, F_)
    s1 = dgn_oracle.aggregate_graph(src, dst, N, ones[src], eig, ones, ["dir1-dx-no-abs"], ["identity"], torch.tensor(1.0))
    d1 = dense_oracle.aggregate("dir1-dx", ones.unsqueeze(0).unsqueeze(1).expand(1, N, N, F_), adj, eigvec=eig.unsqueeze(0))[0]
    assert s1.abs().max() < 1e-5 and d1.abs().max() < 1e-4
