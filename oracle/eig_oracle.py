"""CPU oracle for the Laplacian eigenvectors.  TEST INFRASTRUCTURE ONLY.

Restates realworld_benchmark/data/molecules.py:100-116 (``get_eig``) per graph with numpy: ``L = D - A`` with the
in-degrees clipped to 1 (``'none'``), ``I - D^-1/2 A D^-1/2`` (``'sym'``), eigenvectors in increasing eigenvalue order.
The reference solves with ARPACK (``sp.linalg.eigs(L, k, which='SR', tol=5e-1)``, random start vector), i.e. loosely
converged vectors with arbitrary signs -- nothing reproducible to pin there.  What IS pinned (fixture G9,
tests/golden/make_golden.py::g9_laplacian drives the unmodified ``get_eig`` with a fake graph and an intercepted solver):
the matrix L the reference builds for ``norm`` in none / sym / walk, and its own sort / truncate / cast of the solver's
output; ``tests/test_eig_oracle_vs_golden.py`` holds this restatement to both.  The exact dense solver used here is the
published definition of what ARPACK approximates; comparisons of eigenvectors are by eigen-SUBSPACE.  (Unpinned, DGL
0.4.2 absent: the orientation of ``adjacency_matrix_scipy`` -- rows = destinations -- which only matters for directed
graphs, where this restatement symmetrises A and the reference takes real parts of a non-symmetric solve.)
"""
import numpy as np


def graph_laplacian(src, dst, n, norm="none"):
    A = np.zeros((n, n))
    np.add.at(A, (dst, src), 1.0)
    deg = np.clip(A.sum(1), 1, None)                      # in-degrees (row = destination), clipped (:104)
    A = 0.5 * (A + A.T)
    if norm == "none":
        return np.diag(deg) - A                           # :104-105
    if norm == "walk":
        return np.eye(n) - A / deg[:, None]               # :109-111  I - D^-1 A (not symmetric)
    d = deg ** -0.5
    return np.eye(n) - d[:, None] * A * d[None, :]        # :106-108


def eigvecs(src, dst, sizes, k, norm="none"):
    """per-graph list of (eigenvalues [n], eigenvectors [n, n]) in increasing order"""
    out, off = [], 0
    for n in sizes:
        m = (dst >= off) & (dst < off + n)
        L = graph_laplacian(src[m] - off, dst[m] - off, n, norm)
        if norm == "walk":      # same eigenvalues as the sym form; eigenvectors D^-1/2 u, unit length (what np.linalg.eig returns)
            deg = np.clip(np.bincount(dst[m] - off, minlength=n).astype(float), 1, None)
            w, u = np.linalg.eigh(graph_laplacian(src[m] - off, dst[m] - off, n, "sym"))
            v = u / np.sqrt(deg)[:, None]
            v = v / np.linalg.norm(v, axis=0, keepdims=True)
        else:
            w, v = np.linalg.eigh(L)
        out.append((w, v))
        off += n
    return out
