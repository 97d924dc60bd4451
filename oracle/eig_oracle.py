"""CPU oracle for the Laplacian eigenvectors.  TEST INFRASTRUCTURE ONLY.

Restates realworld_benchmark/data/molecules.py:100-116 (``get_eig``) per graph with numpy: ``L = D - A`` with the
in-degrees clipped to 1 (``'none'``), ``I - D^-1/2 A D^-1/2`` (``'sym'``), eigenvectors in increasing eigenvalue order.
The reference solves with ARPACK (``sp.linalg.eigs(L, k, which='SR', tol=5e-1)``, random start vector), i.e. loosely
converged vectors with arbitrary signs: PARITY UNPINNED by construction -- there is nothing reproducible to pin
(and the function needs DGL graph methods, so it cannot be driven from the fixtures' fake graph).  The exact dense
solver used here is the published definition of what ARPACK approximates; tests compare eigen-SUBSPACES.
"""
import numpy as np


def graph_laplacian(src, dst, n, norm="none"):
    A = np.zeros((n, n))
    np.add.at(A, (dst, src), 1.0)
    deg = np.clip(A.sum(1), 1, None)                      # in-degrees (row = destination), clipped (:104)
    A = 0.5 * (A + A.T)
    if norm == "none":
        return np.diag(deg) - A                           # :104-105
    d = deg ** -0.5
    return np.eye(n) - d[:, None] * A * d[None, :]        # :106-108


def eigvecs(src, dst, sizes, k, norm="none"):
    """per-graph list of (eigenvalues [n], eigenvectors [n, n]) in increasing order"""
    out, off = [], 0
    for n in sizes:
        m = (dst >= off) & (dst < off + n)
        L = graph_laplacian(src[m] - off, dst[m] - off, n, norm)
        w, v = np.linalg.eigh(L)
        out.append((w, v))
        off += n
    return out
