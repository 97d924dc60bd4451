"""Seeded synthetic graph batches shaped like the BASELINE configs (the datasets themselves --
ZINC.pkl, CIFAR10.pkl, ogbg-molhiv -- are not available offline).

Every generator returns a dict of CPU tensors
    src, dst [E] int64 (edge-id order), num_nodes, sizes [n_graphs], eig [N, K] fp32, snorm_n [N, 1]
which ``DGNGraph(src, dst, num_nodes, eig)`` turns into the CSR batch.  ``powerlaw_csr`` builds the
10M-node / 200M-edge graph directly as a destination-major CSR on the device.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch


def _laplacian_eig(n: int, und_edges: np.ndarray, k: int) -> np.ndarray:
    """k lowest eigenvectors of L = D - A (data/molecules.py:100-116 of the reference, norm='none'),
    dense eigh instead of scipy eigs (tiny graphs)."""
    A = np.zeros((n, n))
    A[und_edges[:, 0], und_edges[:, 1]] = 1.0
    A[und_edges[:, 1], und_edges[:, 0]] = 1.0
    L = np.diag(np.clip(A.sum(1), 1, None)) - A
    _, vec = np.linalg.eigh(L)
    out = np.zeros((n, k), dtype=np.float32)
    kk = min(k, n)
    out[:, :kk] = vec[:, :kk]
    return out


def molecule_batch(n_graphs: int, seed: int = 41, n_lo: int = 9, n_hi: int = 37, extra_bonds: float = 2.7,
                   eig_dim: int = 6, laplacian_eig: bool = True) -> Dict[str, torch.Tensor]:
    """Molecule-like graphs: random spanning tree with max degree 4 plus a few ring-closing bonds,
    stored as symmetric directed edges (ZINC: n ~ U{9..37}, ~24.9 bonds -> ~49.8 directed edges)."""
    rng = np.random.default_rng(seed)
    srcs, dsts, sizes, eigs = [], [], [], []
    off = 0
    for _ in range(n_graphs):
        n = int(rng.integers(n_lo, n_hi + 1))
        deg = np.zeros(n, dtype=np.int64)
        und = []
        for v in range(1, n):
            cand = np.flatnonzero(deg[:v] < 4)
            u = int(cand[rng.integers(0, len(cand))]) if len(cand) else int(rng.integers(0, v))
            und.append((u, v))
            deg[u] += 1
            deg[v] += 1
        have = set(und)
        for _ in range(int(rng.poisson(extra_bonds))):
            a, b = (int(x) for x in rng.integers(0, n, 2))
            a, b = min(a, b), max(a, b)
            if a != b and (a, b) not in have and deg[a] < 4 and deg[b] < 4:
                have.add((a, b))
                und.append((a, b))
                deg[a] += 1
                deg[b] += 1
        und = np.asarray(und, dtype=np.int64).reshape(-1, 2)
        s = np.concatenate([und[:, 0], und[:, 1]]) + off
        d = np.concatenate([und[:, 1], und[:, 0]]) + off
        srcs.append(s)
        dsts.append(d)
        sizes.append(n)
        eigs.append(_laplacian_eig(n, und, eig_dim) if laplacian_eig
                    else rng.standard_normal((n, eig_dim)).astype(np.float32))
        off += n
    sizes_t = torch.tensor(sizes)
    snorm = torch.cat([torch.full((n, 1), 1.0 / n) for n in sizes]).sqrt()
    return dict(src=torch.from_numpy(np.concatenate(srcs)), dst=torch.from_numpy(np.concatenate(dsts)), num_nodes=off,
                sizes=sizes_t, eig=torch.from_numpy(np.concatenate(eigs)), snorm_n=snorm)


def subset_batch(b: Dict[str, torch.Tensor], graph_ids) -> Dict[str, torch.Tensor]:
    """The sub-batch holding the given graphs of ``b`` (in the given order, nodes relabelled consecutively): what a
    data-parallel rank keeps of a global batch (``dist.shard_by_edges``)."""
    sizes = b["sizes"].long()
    offs = torch.zeros(sizes.numel() + 1, dtype=torch.long)
    offs[1:] = torch.cumsum(sizes, 0)
    gids = torch.as_tensor(list(graph_ids), dtype=torch.long)
    keep_sizes = sizes[gids]
    new_offs = torch.zeros(gids.numel() + 1, dtype=torch.long)
    new_offs[1:] = torch.cumsum(keep_sizes, 0)
    n_new = int(new_offs[-1])
    # old node id of every kept node, and the inverse map (-1 = dropped)
    nodes = torch.repeat_interleave(offs[gids] - new_offs[:-1], keep_sizes) + torch.arange(n_new)
    new_id = torch.full((int(offs[-1]),), -1, dtype=torch.long)
    new_id[nodes] = torch.arange(n_new)
    keep = new_id[b["dst"]] >= 0                      # edges never cross graphs
    return dict(src=new_id[b["src"][keep]], dst=new_id[b["dst"][keep]], num_nodes=n_new, sizes=keep_sizes,
                eig=b["eig"][nodes], snorm_n=b["snorm_n"][nodes])


def edges_per_graph(b: Dict[str, torch.Tensor]) -> torch.Tensor:
    gid = torch.repeat_interleave(torch.arange(b["sizes"].numel()), b["sizes"].long())
    return torch.bincount(gid[b["dst"]], minlength=b["sizes"].numel())


def knn_batch(n_graphs: int = 128, seed: int = 41, n_lo: int = 85, n_hi: int = 150, k: int = 8) -> Dict[str, torch.Tensor]:
    """CIFAR10-superpixel-like graphs: 2-D points, every node sends an edge to its k nearest
    (data/superpixels.py:139-145), so in-degree varies and can be 0; eig = [0, x, y] (coord_eig mode,
    data/superpixels.py:423-428)."""
    rng = np.random.default_rng(seed)
    srcs, dsts, sizes, eigs = [], [], [], []
    off = 0
    for _ in range(n_graphs):
        n = int(rng.integers(n_lo, n_hi + 1))
        pts = rng.random((n, 2))
        d2 = ((pts[:, None] - pts[None]) ** 2).sum(-1)
        np.fill_diagonal(d2, np.inf)
        nbr = np.argsort(d2, axis=1)[:, :k]
        srcs.append(np.repeat(np.arange(n), k) + off)
        dsts.append(nbr.reshape(-1) + off)
        sizes.append(n)
        eigs.append(np.concatenate([np.zeros((n, 1)), pts], axis=1).astype(np.float32))
        off += n
    snorm = torch.cat([torch.full((n, 1), 1.0 / n) for n in sizes]).sqrt()
    return dict(src=torch.from_numpy(np.concatenate(srcs)), dst=torch.from_numpy(np.concatenate(dsts)), num_nodes=off,
                sizes=torch.tensor(sizes), eig=torch.from_numpy(np.concatenate(eigs)), snorm_n=snorm)


def sbm_batch(n_graphs: int = 128, seed: int = 41, n_lo: int = 44, n_hi: int = 188, communities: int = 5, p_in: float = 0.5,
              p_out: float = 0.35, eig_dim: int = 3) -> Dict[str, torch.Tensor]:
    """PATTERN-like graphs (Benchmarking-GNNs SBM_PATTERN: stochastic block model, 5 communities, p = 0.5 inside / q = 0.35
    between; 44..188 nodes, mean ~119 nodes and ~6 100 directed edges per graph), symmetric directed edges, Laplacian eig."""
    rng = np.random.default_rng(seed)
    srcs, dsts, sizes, eigs = [], [], [], []
    off = 0
    for _ in range(n_graphs):
        n = int(rng.integers(n_lo, n_hi + 1))
        comm = rng.integers(0, communities, n)
        prob = np.where(comm[:, None] == comm[None, :], p_in, p_out)
        upper = np.triu(rng.random((n, n)) < prob, k=1)
        a, b = np.nonzero(upper)
        und = np.stack([a, b], axis=1).astype(np.int64)
        srcs.append(np.concatenate([und[:, 0], und[:, 1]]) + off)
        dsts.append(np.concatenate([und[:, 1], und[:, 0]]) + off)
        sizes.append(n)
        eigs.append(_laplacian_eig(n, und, eig_dim))
        off += n
    snorm = torch.cat([torch.full((n, 1), 1.0 / n) for n in sizes]).sqrt()
    return dict(src=torch.from_numpy(np.concatenate(srcs)), dst=torch.from_numpy(np.concatenate(dsts)), num_nodes=off,
                sizes=torch.tensor(sizes), eig=torch.from_numpy(np.concatenate(eigs)), snorm_n=snorm)


def powerlaw_csr(num_nodes: int, num_edges: int, device, seed: int = 0, alpha: float = 2.1, k_eig: int = 4,
                 generator: Optional[torch.Generator] = None):
    """Destination-major CSR of a power-law graph built on ``device``: in-degree ~ Zipf(alpha) clipped to
    N/10 and rescaled to ~num_edges (min 1), sources uniform over the nodes.
    Returns (indptr int64 [N+1], src int32 [E], eig fp32 [N, k_eig])."""
    gen = generator or torch.Generator(device=device).manual_seed(seed)
    u = torch.rand(num_nodes, device=device, generator=gen, dtype=torch.float64).clamp_min(1e-12)
    raw = torch.floor(u.pow(-1.0 / (alpha - 1.0))).clamp(max=num_nodes / 10)
    deg = torch.clamp((raw * (num_edges / raw.sum())).round(), min=1).long()
    indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(deg, 0)
    E = int(indptr[-1].item())
    src = torch.randint(0, num_nodes, (E,), device=device, generator=gen, dtype=torch.int32)
    eig = torch.randn(num_nodes, k_eig, device=device, generator=gen)
    return indptr, src, eig
