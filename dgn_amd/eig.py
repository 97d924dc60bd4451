"""Laplacian eigenvectors of a batch of graphs on the GPU, and the eigenvector augmentations of the training loops
(SURVEY.md section 8(f) rank 3).

The reference computes ``g.ndata['eig']`` per graph on the CPU at dataset-load time: ``L = D - A`` (or the sym / walk
normalisation) and the k eigenvectors of smallest eigenvalue by ARPACK with ``tol=5e-1`` and a random start vector
(realworld_benchmark/data/molecules.py:100-116, data/PCBA.py:23-78, data/HIV.py likewise) -- minutes for a dataset,
and neither signs nor degenerate subspaces are reproducible.  Here a whole batch is one batched dense symmetric
eigendecomposition on the GPU (``torch.linalg.eigh`` -> rocSOLVER): graphs are bucketed by size, each Laplacian is
padded to the bucket's width with a large diagonal (the padding eigenpairs sort last and, L being block diagonal,
never mix with the graph's), and the k lowest eigenvectors are scattered back to ``[N, k]`` in node order.  Exact
eigenvectors instead of ARPACK's loosely converged ones: same subspaces, signs arbitrary (the training loops flip
them at random anyway, train_molecules_graph_regression.py:29-33).

Scope: graphs given with a symmetric edge list (molecules, SBMs); for a directed graph the symmetrised adjacency
``(A + A^T)/2`` is used (the reference takes the real part of a non-symmetric ARPACK solve there).  Giant single
graphs (config 5: 10 M nodes) cannot be densified: ``lobpcg_eigvecs`` finds the k lowest eigenpairs with a block
LOBPCG iteration whose only large operation is ``L X`` for a thin block ``X [N, m]`` -- evaluated by the aggregation
sweep itself (``A X`` is the ``sum`` aggregator of the block's rows, ``A^T X`` its backward), so the solver runs at the
sweep's HBM rate and needs no sparse-matrix library.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

_PAD = 1.0e4      # diagonal of the padding rows: far above any Laplacian eigenvalue (<= 2 * max degree)


def laplacian_eigvecs(src: torch.Tensor, dst: torch.Tensor, sizes: Sequence[int], k: int, norm: str = "none",
                      bucket: int = 8) -> torch.Tensor:
    """``[N, k]`` fp32: column j is the eigenvector of the j-th smallest eigenvalue of every graph's Laplacian (what
    ``get_eig(pos_enc_dim=k)`` stores, molecules.py:100-116); graphs with fewer than k nodes get zero columns.
    ``src``/``dst``: edges of the batched graph (global node ids, graphs occupy consecutive ranges ``sizes``)."""
    if norm not in ("none", "sym", "walk"):
        raise ValueError(norm)
    dev = src.device
    sizes_t = torch.as_tensor(list(sizes), dtype=torch.long, device=dev)
    G, N = sizes_t.numel(), int(sizes_t.sum().item())
    off = torch.zeros(G + 1, dtype=torch.long, device=dev)
    off[1:] = torch.cumsum(sizes_t, 0)
    gid = torch.repeat_interleave(torch.arange(G, device=dev), sizes_t)            # node -> graph
    loc = torch.arange(N, device=dev) - off[gid]                                    # node -> index inside its graph
    src, dst = src.long(), dst.long()
    deg = torch.bincount(dst, minlength=N).double().clamp_(min=1.0)                 # in-degrees, clipped like :104
    out = torch.zeros(N, k, dtype=torch.float32, device=dev)
    width = ((sizes_t + bucket - 1) // bucket) * bucket                              # bucket width of every graph
    for w in torch.unique(width).tolist():
        if w == 0:
            continue
        sel = torch.nonzero(width == w).flatten()                                    # graphs of this bucket
        slot = torch.full((G,), -1, dtype=torch.long, device=dev)
        slot[sel] = torch.arange(sel.numel(), device=dev)
        L = torch.zeros(sel.numel(), w, w, dtype=torch.float64, device=dev)
        e_ok = slot[gid[dst]] >= 0
        b, i, j = slot[gid[dst[e_ok]]], loc[dst[e_ok]], loc[src[e_ok]]
        a = torch.ones(b.numel(), dtype=torch.float64, device=dev)
        if norm == "sym":                                                           # I - D^-1/2 A D^-1/2   (:106-108)
            a = a / torch.sqrt(deg[dst[e_ok]] * deg[src[e_ok]])
        elif norm == "walk":                                                        # I - D^-1 A has the eigenvalues of the sym form,
            a = a / torch.sqrt(deg[dst[e_ok]] * deg[src[e_ok]])                     # eigenvectors D^-1/2 v (applied below)   (:109-111)
        L.index_put_((b, i, j), -0.5 * a, accumulate=True)                          # symmetrised adjacency
        L.index_put_((b, j, i), -0.5 * a, accumulate=True)
        n_ok = slot[gid] >= 0
        nb, ni = slot[gid[n_ok]], loc[n_ok]
        diag = deg[n_ok] if norm == "none" else torch.ones_like(deg[n_ok])
        L[nb, ni, ni] += diag
        pad = torch.arange(w, device=dev).unsqueeze(0) >= sizes_t[sel].unsqueeze(1)  # [g, w] padding positions
        L[:, torch.arange(w), torch.arange(w)] += pad.double() * _PAD
        _, vec = torch.linalg.eigh(L)                                                # ascending eigenvalues
        kk = min(k, w)
        v = vec[:, :, :kk]                                                           # [g, w, kk]
        if norm == "walk":
            dpad = torch.ones(sel.numel(), w, dtype=torch.float64, device=dev)
            dpad[nb, ni] = deg[n_ok]
            v = v / torch.sqrt(dpad).unsqueeze(-1)
            v = v / v.norm(dim=1, keepdim=True).clamp_min(1e-30)
        valid = (torch.arange(kk, device=dev).unsqueeze(0) < sizes_t[sel].unsqueeze(1)).unsqueeze(1)   # fewer nodes than k
        v = torch.where(valid, v, torch.zeros_like(v))
        out[n_ok, :kk] = v[nb, ni].float()
    return out


# ---- iterative solver for graphs that cannot be densified ---------------------------------------------------------------

def _gram_basis(S: torch.Tensor, rtol: float = 1e-10) -> torch.Tensor:
    """M [c, r] such that Q = S M has orthonormal columns spanning span(S) (eigen-decomposition of the Gram matrix,
    directions below ``rtol`` of the largest dropped: converged residuals make S rank deficient, which a Cholesky /
    Householder factor would turn into infinities).  The columns of S are scaled to unit length first."""
    d = S.norm(dim=0).clamp_min(1e-300)
    G = (S.T @ S) / (d.unsqueeze(0) * d.unsqueeze(1))
    sig, V = torch.linalg.eigh(0.5 * (G + G.T))
    keep = sig > rtol * sig[-1]
    return (V[:, keep] / torch.sqrt(sig[keep]).unsqueeze(0)) / d.unsqueeze(1)


def lobpcg_lowest(matvec, n: int, k: int, diag: Optional[torch.Tensor] = None, block: Optional[int] = None, iters: int = 80,
                  tol: float = 1e-4, device=None, generator: Optional[torch.Generator] = None, x0: Optional[torch.Tensor] = None):
    """The k lowest eigenpairs of a symmetric positive semi-definite operator given as ``matvec(X [n, m]) -> L X``
    (Knyazev's LOBPCG with a Jacobi preconditioner ``diag``; Rayleigh-Ritz on span[X, W, P] in fp64, the basis
    orthonormalised through its Gram matrix so that converged directions drop out instead of breaking the factorisation).
    One ``matvec`` of an m-column block per iteration.  Returns (eigenvalues [k], eigenvectors [n, k] fp64, iterations,
    residual norms [k])."""
    m = block or (k + 4 + (k % 2))
    dev = device or (x0.device if x0 is not None else "cpu")
    if x0 is not None:
        X = x0.double()
    else:
        X = torch.randn(n, m, dtype=torch.float64, device=dev, generator=generator)
        X[:, 0] = 1.0                           # the constant vector spans the null space of a connected graph's L
    m = X.shape[1]
    M0 = _gram_basis(X)
    X = X @ M0
    m = X.shape[1]
    LX = matvec(X)
    P = LP = None
    lam = res = None
    best = None                                                      # (residual, lam, X, iteration) of the best iterate so far
    it = 0
    for it in range(1, iters + 1):
        if it % 10 == 0:
            # L X and L P are carried along as linear combinations of earlier products: refresh them now and then so that
            # rounding does not accumulate (one extra product every ten iterations)
            X = X @ _gram_basis(X)
            LX = matvec(X)
            P = LP = None
        lam, C = torch.linalg.eigh(X.T @ LX)                        # Rayleigh-Ritz inside span(X): X becomes the Ritz vectors
        X, LX = X @ C, LX @ C
        R = LX - X * lam.unsqueeze(0)
        res = R.norm(dim=0)
        scale = max(float(lam.abs().max()), 1e-12)
        worst = float(res[:k].max())
        if best is None or worst < best[0]:
            best = (worst, lam[:k].clone(), X[:, :k].clone(), it, res[:k].clone())
        if worst <= tol * scale:
            break
        if it - best[3] >= 15:                                       # stagnation at the rounding floor of the products
            break
        W = R / diag.unsqueeze(1) if diag is not None else R
        W = W - X @ (X.T @ W)                                        # keep the search directions out of span(X)
        live = W.norm(dim=0) > 1e-14 * scale                         # (converged columns contribute no direction)
        W = W[:, live]
        if W.shape[1] == 0:
            break
        LW = matvec(W)
        S = torch.cat([X, W] + ([P] if P is not None else []), dim=1)
        LS = torch.cat([LX, LW] + ([LP] if P is not None else []), dim=1)
        M = _gram_basis(S)                                           # [cols(S), r]
        T = M.T @ (S.T @ LS) @ M
        _, Cq = torch.linalg.eigh(0.5 * (T + T.T))
        Y = M @ Cq[:, :m]                                            # new X = S Y
        Xn, LXn = S @ Y, LS @ Y
        # implicit conjugate direction: the part of the new X that does not come from the old X
        P, LP = S[:, m:] @ Y[m:], LS[:, m:] @ Y[m:]
        pn = P.norm(dim=0)
        ok = pn > 1e-14
        P, LP = P[:, ok] / pn[ok], LP[:, ok] / pn[ok]
        if P.shape[1] == 0:
            P = LP = None
        X, LX = Xn, LXn
    return best[1], best[2], it, best[4]


def sweep_laplacian_matvec(graph, norm: str = "none", symmetric: bool = False):
    """``X [N, m] fp64 -> L X`` for the graph's Laplacian, the product evaluated by the aggregation kernels in fp32:
    ``(A X)[i] = sum over in-edges of X[src]`` is the ``sum`` aggregator (dgn_agg_forward), ``A^T X`` its backward
    (dgn_agg_backward with the block as upstream gradient); L = D - (A + A^T)/2 with the in-degrees clipped to 1
    (molecules.py:104-105), or I - D^-1/2 (A + A^T)/2 D^-1/2 for ``norm='sym'`` (:106-108).  ``symmetric=True`` skips
    the transposed product (undirected graphs stored as symmetric edge lists).  Returns (matvec, diag(L))."""
    from .ops import launch_backward, launch_forward
    from .spec import make_plan
    if norm not in ("none", "sym"):
        raise ValueError(norm)
    plan = make_plan(["sum"], ["identity"])
    N, dev = graph.num_nodes, graph.device
    deg = graph.in_degree.to(torch.float64).clamp(min=1.0)
    dinv = deg.rsqrt() if norm == "sym" else None

    def adj(X32):
        out = torch.empty_like(X32)
        launch_forward(graph, plan, 1, 1.0, None, X32, None, None, None, out)
        if symmetric:
            return out
        gt = torch.empty_like(X32)
        launch_backward(graph, plan, 1, 1.0, None, X32, None, None, None, X32, gt, None, None, None, accumulate=False)
        return 0.5 * (out + gt)

    def matvec(X):
        pad = X.shape[1] % 2                                          # (even widths: 8-byte lanes, atomic-free scatter)
        Z = X * dinv.unsqueeze(1) if dinv is not None else X
        Z32 = Z.float()
        if pad:
            Z32 = torch.nn.functional.pad(Z32, (0, 1))
        AZ = adj(Z32.contiguous())[:, :X.shape[1]].double()
        if dinv is not None:
            return X - AZ * dinv.unsqueeze(1)
        return X * deg.unsqueeze(1) - AZ

    diag = torch.ones_like(deg) if norm == "sym" else deg
    return matvec, diag


def lobpcg_eigvecs(graph, k: int, norm: str = "none", symmetric: bool = False, iters: int = 80, tol: float = 1e-3,
                   generator: Optional[torch.Generator] = None):
    """``[N, k]`` fp32 eigenvectors of the k smallest Laplacian eigenvalues of ONE (large) graph given as a ``DGNGraph``
    -- what ``get_eig`` (molecules.py:100-116) would store in ``g.ndata['eig']`` -- plus (eigenvalues, iterations, residuals).
    The reference's ARPACK call stops at ``tol=5e-1``; ``tol`` here is the relative residual ``||L x - lambda x|| / lambda_max``."""
    matvec, diag = sweep_laplacian_matvec(graph, norm, symmetric)
    lam, X, it, res = lobpcg_lowest(matvec, graph.num_nodes, k, diag=diag, iters=iters, tol=tol, device=graph.device,
                                    generator=generator)
    return X.float(), lam, it, res


# ---- augmentations of the training loops (train/train_superpixels_graph_classification.py:29-48) -------------------

def flip_sign(eig: torch.Tensor, col: Optional[int] = None, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Random +-1 per ENTRY (as the reference draws it: ``torch.rand(eig.size())``), on one column or on all
    (train_superpixels...:39-43, train_molecules_graph_regression.py:29-33)."""
    tgt = eig if col is None else eig[:, col]
    sign = torch.where(torch.rand(tgt.shape, device=eig.device, generator=generator) >= 0.5, 1.0, -1.0).to(eig.dtype)
    out = eig.clone()
    if col is None:
        return out * sign
    out[:, col] = tgt * sign
    return out


def distort(eig: torch.Tensor, distortion: float, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Per-node random offset of the (eig1, eig2) pair, scaled by the batch mean of |column| (train_superpixels...:43-47):
    ``eig[:, c] += dist * mean(|eig[:, c]|)`` with ONE ``dist ~ U(-distortion, distortion)`` per node for both columns.
    (The reference's line for column 2 adds the whole ``[N, K]`` tensor instead of column 2 -- a shape error unless
    ``N == K``; the evident intent, the same formula as for column 1, is what is implemented.)"""
    dist = (torch.rand(eig.shape[0], device=eig.device, generator=generator) - 0.5) * 2 * distortion
    out = eig.clone()
    out[:, 1] = dist * eig[:, 1].abs().mean() + eig[:, 1]
    out[:, 2] = dist * eig[:, 2].abs().mean() + eig[:, 2]
    return out


def rotate(eig: torch.Tensor, augmentation_deg: float, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Per-node random rotation of the (eig1, eig2) pair by up to +-augmentation degrees (train_superpixels...:29-37)."""
    angle = (torch.rand(eig.shape[0], device=eig.device, generator=generator) - 0.5) * 2 * augmentation_deg
    sine = torch.sin(angle * math.pi / 180)
    cos = (1 - sine ** 2) ** 0.5
    out = eig.clone()
    out[:, 1] = cos * eig[:, 1] + sine * eig[:, 2]
    out[:, 2] = cos * eig[:, 2] - sine * eig[:, 1]
    return out
