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
graphs (config 5) would need an iterative solver (LOBPCG) -- not provided.
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


def rotate(eig: torch.Tensor, augmentation_deg: float, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Per-node random rotation of the (eig1, eig2) pair by up to +-augmentation degrees (train_superpixels...:29-37)."""
    angle = (torch.rand(eig.shape[0], device=eig.device, generator=generator) - 0.5) * 2 * augmentation_deg
    sine = torch.sin(angle * math.pi / 180)
    cos = (1 - sine ** 2) ** 0.5
    out = eig.clone()
    out[:, 1] = cos * eig[:, 1] + sine * eig[:, 2]
    out[:, 2] = cos * eig[:, 2] - sine * eig[:, 1]
    return out
