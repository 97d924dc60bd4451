#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference.

Runs only in the build container (needs /root/reference); never on the GPU box.
The fixtures are numeric arrays only -- inputs and the reference's outputs/grads.
No reference source or bytecode is copied (sys.dont_write_bytecode is set before
any import from the reference tree).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What drives the reference:
* ``realworld_benchmark/nets/{aggregators,scalers,layers}.py`` import as they are.
* ``realworld_benchmark/nets/dgn_layer.py`` needs ``dgl.nn.pytorch.glob`` only for
  ``VirtualNode``; the nets call ``dgl.{sum,mean,max}_nodes``: four stub modules are registered whose
  ``*_nodes`` functions reduce ``g.ndata[key]`` over the consecutive node blocks ``g.batch_num_nodes`` of a
  batched graph (DGL's documented meaning; DGL's own implementation is absent -> unpinned like the mailbox order).
* DGL itself (0.4.2) is not installed.  ``FakeGraph`` below implements the UDF
  protocol the layer relies on (ndata/edata, apply_edges, update_all with degree
  bucketing).  Two DGL-internal behaviours are NOT pinned by the reference and are
  fixed here by definition: zero-in-degree rows are zeros; a destination's mailbox
  is in ascending edge-id order.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch

torch.set_num_threads(1)


def _install_stubs():
    for name in ("dgl", "dgl.nn", "dgl.nn.pytorch", "dgl.nn.pytorch.glob"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    def seg(how):
        def fn(g, feat):
            x, outs, off = g.ndata[feat], [], 0
            for n in g.batch_num_nodes:
                blk = x[off:off + n]
                outs.append(blk.sum(0) if how == "sum" else (blk.mean(0) if how == "mean" else blk.max(0)[0]))
                off += n
            return torch.stack(outs)
        return fn
    glob, dgl = sys.modules["dgl.nn.pytorch.glob"], sys.modules["dgl"]
    for how in ("sum", "mean", "max"):
        setattr(dgl, how + "_nodes", seg(how))
    glob.mean_nodes, glob.sum_nodes = dgl.mean_nodes, dgl.sum_nodes
    sys.path.insert(0, os.path.join(REF, "realworld_benchmark"))
    sys.path.insert(1, REF)


class _Edges:
    def __init__(self, src, dst, data):
        self.src, self.dst, self.data = src, dst, data


class _Nodes:
    def __init__(self, data, mailbox):
        self.data, self.mailbox = data, mailbox


class FakeGraph:
    """Minimal stand-in for a (batched) DGLGraph 0.4.x, enough for nets/dgn_layer.py."""

    def __init__(self, src, dst, num_nodes):
        self.src = torch.as_tensor(src, dtype=torch.long)
        self.dst = torch.as_tensor(dst, dtype=torch.long)
        self.n = int(num_nodes)
        self.ndata, self.edata = {}, {}

    def number_of_nodes(self):
        return self.n

    def number_of_edges(self):
        return self.src.numel()

    def edges(self):
        return self.src, self.dst

    def apply_edges(self, func):
        e = _Edges({k: v[self.src] for k, v in self.ndata.items()},
                   {k: v[self.dst] for k, v in self.ndata.items()}, self.edata)
        self.edata.update(func(e))

    def update_all(self, message_func, reduce_func):
        e = _Edges({k: v[self.src] for k, v in self.ndata.items()},
                   {k: v[self.dst] for k, v in self.ndata.items()}, self.edata)
        msgs = message_func(e)
        order = torch.sort(self.dst, stable=True)[1]
        deg = torch.bincount(self.dst, minlength=self.n)
        ptr = torch.zeros(self.n + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(deg, 0)
        result = {}
        for D in torch.unique(deg).tolist():
            if D == 0:
                continue
            nodes = torch.nonzero(deg == D).flatten()
            slots = (ptr[nodes].unsqueeze(1) + torch.arange(D).unsqueeze(0)).reshape(-1)
            eids = order[slots]
            mailbox = {k: v[eids].reshape(len(nodes), D, *v.shape[1:]) for k, v in msgs.items()}
            out = reduce_func(_Nodes({k: v[nodes] for k, v in self.ndata.items()}, mailbox))
            for k, v in out.items():
                if k not in result:
                    result[k] = v.new_zeros((self.n,) + tuple(v.shape[1:]))
                result[k] = result[k].index_copy(0, nodes, v)
        self.ndata.update(result)


# ---------------------------------------------------------------------------- graphs


def make_test_graph(seed=0):
    """Batched graph: 2 molecule-like symmetric graphs + 1 directed kNN-like graph with a
    zero-in-degree node + one duplicated edge.  Returns src, dst, N, graph sizes."""
    rng = np.random.default_rng(seed)
    src, dst, sizes, off = [], [], [], 0
    for n in (17, 23):
        und = set()
        for v in range(1, n):          # random spanning tree
            und.add((int(rng.integers(0, v)), v))
        while len(und) < n + 2:         # a few rings
            a, b = rng.integers(0, n, 2)
            if a != b:
                und.add((int(min(a, b)), int(max(a, b))))
        for a, b in sorted(und):
            src += [a + off, b + off]
            dst += [b + off, a + off]
        sizes.append(n)
        off += n
    n = 20
    pts = rng.random((n, 2))
    d2 = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    d2[:, n - 1] = np.inf                # nobody points at the last node -> in-degree 0
    for v in range(n):
        for u in np.argsort(d2[v])[:4]:
            src.append(v + off)
            dst.append(int(u) + off)
    src.append(src[-1])                  # duplicated edge
    dst.append(dst[-1])
    sizes.append(n)
    off += n
    perm = rng.permutation(len(src))     # shuffle edge ids so CSR order != edge-id order
    return np.asarray(src)[perm], np.asarray(dst)[perm], off, sizes


def graph_inputs(seed, N, E, in_dim, edge_dim, K=6):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(N, in_dim, generator=g)
    e = torch.randn(E, edge_dim, generator=g) if edge_dim else torch.zeros(E, 1)
    eig = torch.randn(N, K, generator=g)
    return h, e, eig


# ---------------------------------------------------------------------------- fixtures


def g1_aggregators(out):
    from nets.aggregators import AGGREGATORS
    names = sorted(AGGREGATORS)
    out["names"] = np.array(names)
    case = 0
    for seed in range(3):
        for D in (1, 2, 3, 5, 8, 33):
            g = torch.Generator().manual_seed(1000 * seed + D)
            n, F_, K = 7, 5, 4
            h = torch.randn(n, D, F_, generator=g)
            es = torch.randn(n, D, K, generator=g)
            ed = torch.randn(n, 1, K, generator=g).expand(n, D, K).contiguous()
            hin = torch.randn(n, F_, generator=g)
            ct = torch.randn(n, F_, generator=g)
            pre = f"c{case}"
            out[f"{pre}/h"], out[f"{pre}/eig_s"], out[f"{pre}/eig_d"] = h.numpy(), es.numpy(), ed.numpy()
            out[f"{pre}/h_in"], out[f"{pre}/cot"] = hin.numpy(), ct.numpy()
            for name in names:
                hh = h.clone().requires_grad_(True)
                xx = hin.clone().requires_grad_(True)
                y = AGGREGATORS[name](hh, es, ed, xx)
                gh, gx = torch.autograd.grad(y, [hh, xx], ct, allow_unused=True)
                out[f"{pre}/{name}/y"] = y.detach().numpy()
                out[f"{pre}/{name}/gh"] = gh.numpy()
                out[f"{pre}/{name}/gx"] = (gx if gx is not None else torch.zeros_like(hin)).numpy()
            case += 1
    out["n_cases"] = np.array(case)


def g2_scalers(out):
    from nets.scalers import SCALERS
    h = torch.randn(4, 6, generator=torch.Generator().manual_seed(7))
    out["h"] = h.numpy()
    Ds = list(range(1, 11)) + [100, 47830]
    out["D"] = np.array(Ds)
    out["avg"] = np.array([0.9, 1.3], dtype=np.float32)
    for ai, avg in enumerate((0.9, 1.3)):
        avg_d = {"log": torch.tensor(avg)}
        for D in Ds:
            for name in sorted(SCALERS):
                out[f"a{ai}/D{D}/{name}"] = SCALERS[name](h, D=D, avg_d=avg_d).numpy()


def _build_layer(type_net, in_dim, out_dim, aggs, scalers, avg_log, residual=True, towers=5,
                 edge_features=False, edge_dim=0, graph_norm=True, batch_norm=True, posttrans_layers=1,
                 pretrans_layers=1, big_weights=True, seed=0):
    from nets.dgn_layer import DGNLayer
    torch.manual_seed(seed)
    layer = DGNLayer(in_dim=in_dim, out_dim=out_dim, dropout=0.0, graph_norm=graph_norm, batch_norm=batch_norm,
                     aggregators=aggs, scalers=scalers, avg_d={"log": torch.tensor(avg_log)}, type_net=type_net,
                     residual=residual, towers=towers, edge_features=edge_features, edge_dim=edge_dim,
                     pretrans_layers=pretrans_layers, posttrans_layers=posttrans_layers).model
    if big_weights:                       # the reference's xavier gain 1/in_size gives ~1e-3 weights
        g = torch.Generator().manual_seed(seed + 99)
        with torch.no_grad():
            for name, p in layer.named_parameters():
                if name.endswith("linear.weight"):
                    p.copy_(torch.randn(p.shape, generator=g) / p.shape[1] ** 0.5)
                elif name.endswith("linear.bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                elif "batchnorm" in name and name.endswith("weight"):
                    p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
                elif "batchnorm" in name and name.endswith("bias"):
                    p.copy_(0.2 * torch.randn(p.shape, generator=g))
    return layer


def _run_layer_case(out, pre, layer, src, dst, N, h, e, eig, snorm, train):
    g = FakeGraph(src, dst, N)
    g.ndata["eig"] = eig
    layer.train(train)
    for k, v in layer.state_dict().items():
        out[f"{pre}/sd::{k}"] = v.detach().numpy().copy()
    hh = h.clone().requires_grad_(True)
    ee = e.clone().requires_grad_(True)
    y = layer(g, hh, ee, snorm)
    ct = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    params = [p for p in layer.parameters()]
    grads = torch.autograd.grad(y, [hh, ee] + params, ct, allow_unused=True)
    out[f"{pre}/y"] = y.detach().numpy()
    out[f"{pre}/cot"] = ct.numpy()
    out[f"{pre}/gh"] = grads[0].numpy()
    out[f"{pre}/ge"] = (grads[1] if grads[1] is not None else torch.zeros_like(e)).numpy()
    for (name, p), gr in zip(layer.named_parameters(), grads[2:]):
        out[f"{pre}/gp::{name}"] = (gr if gr is not None else torch.zeros_like(p)).numpy()
    for k, v in layer.state_dict().items():     # BN running stats after the step
        if "running" in k:
            out[f"{pre}/after::{k}"] = v.detach().numpy().copy()


LAYER_CASES = [
    # name, type_net, in, out, aggregators, scalers, kwargs
    ("simple_c1", "simple", 10, 10, "mean dir1-dx-no-abs", "identity amplification attenuation", {}),
    ("simple_all", "simple", 6, 8, "mean sum max min std var dir1-av dir2-dx dir3-dx-no-abs dir1-dx-balanced dir2-0.1 dir3-neg-0.1",
     "identity amplification attenuation", {}),
    ("simple_single_scaler", "simple", 6, 6, "mean max dir1-dx", "amplification", {}),
    ("simple_eval", "simple", 10, 10, "mean std dir1-dx", "identity attenuation", {"train": False}),
    ("simple_post2", "simple", 6, 6, "mean dir1-av", "identity", {"posttrans_layers": 2}),
    ("complex_noef", "complex", 8, 8, "mean max min dir1-av dir1-dx", "identity amplification attenuation", {}),
    ("complex_ef", "complex", 8, 8, "mean max min dir1-av dir1-dx", "identity amplification attenuation",
     {"edge_features": True, "edge_dim": 3}),
    ("complex_pre2", "complex", 8, 6, "mean std dir2-dx", "identity", {"pretrans_layers": 2, "edge_features": True, "edge_dim": 3}),
    ("towers_c2", "towers", 10, 10, "mean max min dir1-av dir1-dx", "identity amplification attenuation", {"towers": 5}),
    ("towers_ef", "towers", 10, 10, "mean max min dir1-av dir1-dx", "identity amplification attenuation",
     {"towers": 5, "edge_features": True, "edge_dim": 3}),
    ("towers_nodiv", "towers", 6, 9, "mean dir1-dx-no-abs", "identity amplification", {"towers": 3, "divide_input": False}),
    ("towers_one", "towers", 6, 6, "sum var dir1-dx", "identity", {"towers": 1}),
    ("simple_fresh_init", "simple", 10, 10, "mean dir1-dx", "identity", {"big_weights": False}),
]


def g4_layers(out):
    src, dst, N, sizes = make_test_graph(0)
    out["src"], out["dst"], out["N"], out["sizes"] = src, dst, np.array(N), np.array(sizes)
    snorm = torch.cat([torch.full((n, 1), 1.0 / n) for n in sizes]).sqrt()
    out["snorm_n"] = snorm.numpy()
    out["cases"] = np.array([c[0] for c in LAYER_CASES])
    for name, type_net, din, dout, aggs, scalers, kw in LAYER_CASES:
        kw = dict(kw)
        train = kw.pop("train", True)
        divide_input = kw.pop("divide_input", True)
        from nets import dgn_layer as ref_layer
        layer = _build_layer(type_net, din, dout, aggs, scalers, 1.1, **{k: v for k, v in kw.items()}) \
            if divide_input else None
        if layer is None:               # DGNLayer factory always passes divide_input through; do it directly
            torch.manual_seed(0)
            layer = ref_layer.DGNLayer(in_dim=din, out_dim=dout, dropout=0.0, graph_norm=True, batch_norm=True,
                                       aggregators=aggs, scalers=scalers, avg_d={"log": torch.tensor(1.1)},
                                       type_net=type_net, residual=True, towers=kw.get("towers", 5),
                                       divide_input=False, edge_features=False, edge_dim=0).model
            g = torch.Generator().manual_seed(99)
            with torch.no_grad():
                for pn, p in layer.named_parameters():
                    if pn.endswith("linear.weight"):
                        p.copy_(torch.randn(p.shape, generator=g) / p.shape[1] ** 0.5)
        edge_dim = kw.get("edge_dim", 0)
        h, e, eig = graph_inputs(3, N, len(src), din, edge_dim)
        out[f"{name}/h"], out[f"{name}/e"], out[f"{name}/eig"] = h.numpy(), e.numpy(), eig.numpy()
        out[f"{name}/meta"] = np.array([type_net, str(din), str(dout), aggs, scalers, "1.1",
                                        str(kw.get("towers", 5)), str(int(divide_input)),
                                        str(int(bool(kw.get("edge_features", False)))), str(edge_dim),
                                        str(kw.get("pretrans_layers", 1)), str(kw.get("posttrans_layers", 1)),
                                        str(int(train))])
        _run_layer_case(out, name, layer, src, dst, N, h, e, eig, snorm, train)


def g3_reduce(out):
    """reduce_func concat order + the skipped-single-scaler rule (dgn_layer.py:161-173),
    captured through the aggregation output of an (otherwise unused) simple layer."""
    from nets.dgn_layer import DGNLayer
    src, dst, N, sizes = make_test_graph(1)
    out["src"], out["dst"], out["N"] = src, dst, np.array(N)
    h, _, eig = graph_inputs(11, N, len(src), 5, 0)
    out["h"], out["eig"] = h.numpy(), eig.numpy()
    aggs = "mean max dir1-dx"
    out["aggregators"] = np.array(aggs)
    ct = None
    for tag, scalers in (("id", "identity"), ("amp_only", "amplification"),
                         ("three", "identity amplification attenuation"),
                         ("att_amp", "attenuation amplification")):
        layer = DGNLayer(in_dim=5, out_dim=5, dropout=0.0, graph_norm=False, batch_norm=False, aggregators=aggs,
                         scalers=scalers, avg_d={"log": torch.tensor(0.8)}, type_net="simple", residual=False).model
        g = FakeGraph(src, dst, N)
        g.ndata["eig"] = eig
        hh = h.clone().requires_grad_(True)
        g.ndata["h"] = hh
        g.apply_edges(layer.pretrans_edges)
        g.update_all(layer.message_func, layer.reduce_func)
        y = g.ndata["h"]
        ct = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
        out[f"{tag}/scalers"] = np.array(scalers)
        out[f"{tag}/y"] = y.detach().numpy()
        out[f"{tag}/cot"] = ct.numpy()
        out[f"{tag}/gh"] = torch.autograd.grad(y, hh, ct)[0].numpy()


def g5_edge_cases(out):
    from nets.aggregators import AGGREGATORS
    names = sorted(AGGREGATORS)
    out["names"] = np.array(names)
    n, F_, K = 4, 3, 4
    g = torch.Generator().manual_seed(21)
    cases = {}
    # all-equal eig: every delta is 0
    h = torch.randn(n, 4, F_, generator=g)
    es = torch.ones(n, 4, K)
    cases["zero_delta"] = (h, es, es.clone(), torch.randn(n, F_, generator=g))
    # degree 1
    cases["deg1"] = (torch.randn(n, 1, F_, generator=g), torch.randn(n, 1, K, generator=g),
                     torch.randn(n, 1, K, generator=g), torch.randn(n, F_, generator=g))
    # max/min ties (integer-valued features, as after an nn.Embedding of few atom types)
    h = torch.randint(0, 2, (n, 6, F_), generator=g).float()
    ed = torch.randn(n, 1, K, generator=g).expand(n, 6, K).contiguous()
    cases["ties"] = (h, torch.randn(n, 6, K, generator=g), ed, torch.randn(n, F_, generator=g))
    # identical messages -> variance rounds to ~0 (possibly negative before relu)
    h = torch.randn(n, 1, F_, generator=g).expand(n, 5, F_).contiguous() * 1000.0
    ed = torch.randn(n, 1, K, generator=g).expand(n, 5, K).contiguous()
    cases["const_msgs"] = (h, torch.randn(n, 5, K, generator=g), ed, torch.randn(n, F_, generator=g))
    # dx exactly 0: messages all equal to h_in, weights sum to 0 by antisymmetric deltas
    x = torch.randn(n, F_, generator=g)
    h = x.unsqueeze(1).expand(n, 2, F_).contiguous()
    ed = torch.zeros(n, 2, K)
    es = torch.stack([torch.ones(n, K), -torch.ones(n, K)], dim=1)
    cases["dx_zero"] = (h, es, ed, x)
    out["cases"] = np.array(sorted(cases))
    for cname, (h, es, ed, x) in cases.items():
        ct = torch.randn(n, F_, generator=torch.Generator().manual_seed(3))
        out[f"{cname}/h"], out[f"{cname}/eig_s"], out[f"{cname}/eig_d"] = h.numpy(), es.numpy(), ed.numpy()
        out[f"{cname}/h_in"], out[f"{cname}/cot"] = x.numpy(), ct.numpy()
        for name in names:
            hh = h.clone().requires_grad_(True)
            xx = x.clone().requires_grad_(True)
            y = AGGREGATORS[name](hh, es, ed, xx)
            gh, gx = torch.autograd.grad(y, [hh, xx], ct, allow_unused=True)
            out[f"{cname}/{name}/y"] = y.detach().numpy()
            out[f"{cname}/{name}/gh"] = gh.numpy()
            out[f"{cname}/{name}/gx"] = (gx if gx is not None else torch.zeros_like(x)).numpy()


def _dense_inputs(seed, B, N, F_, K, weighted):
    g = torch.Generator().manual_seed(seed)
    adj = (torch.rand(B, N, N, generator=g) < 0.45).float()
    idx = torch.arange(N)
    adj[:, idx, idx] = 0
    adj[:, idx, (idx + 1) % N] = 1          # a ring: no isolated row / column
    if weighted:
        adj = adj * (0.5 + torch.rand(B, N, N, generator=g))      # asymmetric positive weights
    else:
        adj = ((adj + adj.transpose(1, 2)) > 0).float()
    X = torch.randn(B, N, N, F_, generator=g)
    eig = torch.rand(B, N, K, generator=g) * 2 - 1
    return X, adj, eig


def g6_dense(out):
    """Dense formulation models/pytorch/*: every aggregator that runs on a modern torch, all scalers, and the
    dense DGNLayer with 1 and 2 towers."""
    from models.pytorch.aggregators import AGGREGATORS
    from models.pytorch.scalers import SCALERS
    from models.pytorch.dgn_layer import DGNLayer
    names = [n for n in sorted(AGGREGATORS) if n not in ("softmax", "softmin")]
    out["names"] = np.array(names)
    out["broken"] = np.array(["softmax", "softmin"])
    avg_d = {"log": torch.tensor(1.2), "lin": torch.tensor(3.5)}
    out["avg_log"], out["avg_lin"] = np.array(1.2, dtype=np.float32), np.array(3.5, dtype=np.float32)
    cases = {"bin6": (11, 2, 6, 3, 6, False), "wgt10": (12, 2, 10, 3, 6, True)}
    out["cases"] = np.array(sorted(cases))
    for cname, (seed, B, N, F_, K, weighted) in cases.items():
        X, adj, eig = _dense_inputs(seed, B, N, F_, K, weighted)
        out[f"{cname}/X"], out[f"{cname}/adj"], out[f"{cname}/eig"] = X.numpy(), adj.numpy(), eig.numpy()
        for self_loop in (False, True):
            for name in names:
                XX = X.clone().requires_grad_(True)
                y = AGGREGATORS[name](XX, adj, eigvec=eig, self_loop=self_loop, device="cpu", avg_d=avg_d)
                ct = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
                (gX,) = torch.autograd.grad(y, XX, ct)
                tag = f"{cname}/sl{int(self_loop)}/{name}"
                out[f"{tag}/y"], out[f"{tag}/cot"], out[f"{tag}/gX"] = y.detach().numpy(), ct.numpy(), gX.numpy()
        m = torch.randn(B, N, 4, generator=torch.Generator().manual_seed(2))
        out[f"{cname}/scaler_in"] = m.numpy()
        for sname in sorted(SCALERS):
            out[f"{cname}/scaler/{sname}"] = SCALERS[sname](m, adj, avg_d=avg_d).numpy()
    # dense layers
    layer_cases = {"dense_t1": (1, ["mean", "max", "min", "std", "dir1-dx", "dir2-smooth"], ["identity", "amplification", "attenuation"], 6, 8, "bin6"),
                   "dense_t2": (2, ["sum", "var", "dir1-both", "moment3", "normalised_mean"], ["identity", "linear", "inverse_linear"], 6, 6, "wgt10"),
                   "dense_t2_nodiv": (2, ["mean", "dir2-dx"], ["attenuation"], 4, 6, "wgt10")}
    out["layer_cases"] = np.array(sorted(layer_cases))
    for lname, (towers, aggs, scalers, fin, fout, cname) in layer_cases.items():
        torch.manual_seed(0)
        layer = DGNLayer(in_features=fin, out_features=fout, aggregators=aggs, scalers=scalers, NN_eig=False, avg_d=avg_d,
                         eigs=None, towers=towers, self_loop=False, pretrans_layers=1, posttrans_layers=1,
                         divide_input=not lname.endswith("nodiv"), device="cpu")
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for pn, p_ in layer.named_parameters():
                if pn.endswith("weight"):
                    p_.copy_(torch.randn(p_.shape, generator=g) / p_.shape[1] ** 0.5)
                else:
                    p_.copy_(0.1 * torch.randn(p_.shape, generator=g))
        _, adj, eig = _dense_inputs(*cases[cname][:1], *cases[cname][1:])
        B, N = adj.shape[:2]
        inp = torch.randn(B, N, fin, generator=g).requires_grad_(True)
        y = layer(inp, adj, eig)
        ct = torch.randn(y.shape, generator=g)
        params = list(layer.parameters())
        grads = torch.autograd.grad(y, [inp] + params, ct)
        out[f"{lname}/meta"] = np.array([str(towers), " ".join(aggs), " ".join(scalers), str(fin), str(fout), cname,
                                        str(int(not lname.endswith("nodiv")))])
        out[f"{lname}/input"], out[f"{lname}/y"], out[f"{lname}/cot"] = inp.detach().numpy(), y.detach().numpy(), ct.numpy()
        out[f"{lname}/ginput"] = grads[0].numpy()
        for k, v in layer.state_dict().items():
            out[f"{lname}/sd::{k}"] = v.detach().numpy().copy()
        for (pn, _), gr in zip(layer.named_parameters(), grads[1:]):
            out[f"{lname}/gp::{pn}"] = gr.numpy()


def g8_readouts(out):
    """Graph-level readouts of nets/molecules_graph_regression/dgn_net.py:71-86 (captured at the input of the
    net's MLP_layer) and VirtualNode of nets/dgn_layer.py:12-49."""
    from nets.molecules_graph_regression.dgn_net import DGNNet
    from nets.dgn_layer import VirtualNode
    src, dst, N, sizes = make_test_graph(seed=3)
    out["src"], out["dst"], out["N"], out["sizes"] = src, dst, np.array(N), np.array(sizes)
    gen = torch.Generator().manual_seed(11)
    eig = torch.randn(N, 4, generator=gen)
    atoms = torch.randint(0, 5, (N,), generator=gen)
    snorm = torch.rand(N, 1, generator=gen) + 0.5
    out["eig"] = eig.numpy()
    for mode in ("sum", "max", "mean", "directional", "directional_abs"):
        torch.manual_seed(1)
        net = DGNNet(dict(num_atom_type=5, num_bond_type=3, hidden_dim=8, out_dim=8, in_feat_dropout=0.0, dropout=0.0, L=2,
                          type_net="simple", pos_enc_dim=0, readout=mode, graph_norm=True, batch_norm=True,
                          aggregators="mean dir1-dx", scalers="identity", avg_d={"log": torch.tensor(1.0)}, residual=True,
                          edge_feat=False, edge_dim=0, pretrans_layers=1, posttrans_layers=1, device="cpu"))
        g = FakeGraph(src, dst, N)
        g.batch_num_nodes = list(sizes)
        g.ndata["eig"] = eig
        seen = {}
        net.MLP_layer.register_forward_pre_hook(lambda m, inp: seen.__setitem__("hg", inp[0]))
        net(g, atoms, None, snorm, None)
        h_last, hg = g.ndata["h"], seen["hg"]
        ct = torch.randn(hg.shape, generator=torch.Generator().manual_seed(2))
        gh, = torch.autograd.grad(hg, [h_last], ct)
        out[f"readout/{mode}/h"], out[f"readout/{mode}/hg"] = h_last.detach().numpy(), hg.detach().numpy()
        out[f"readout/{mode}/cot"], out[f"readout/{mode}/gh"] = ct.numpy(), gh.numpy()
    G, D = len(sizes), 8
    case = 0
    for vn_type in ("mean", "sum", "logsum"):
        for b_norm, residual in ((False, True), (True, True), (False, False)):
            torch.manual_seed(20 + case)
            vn = VirtualNode(dim=D, dropout=0.0, batch_norm=b_norm, bias=True, residual=residual, vn_type=vn_type)
            with torch.no_grad():
                for q in vn.parameters():
                    q.mul_(3.0).add_(0.1 * torch.randn(q.shape, generator=gen))
            vn.train(True)
            pre = f"vn/c{case}"
            out[f"{pre}/cfg"] = np.array([vn_type, str(int(b_norm)), str(int(residual))])
            for k, v in vn.state_dict().items():
                out[f"{pre}/sd::{k}"] = v.detach().numpy().copy()
            h = torch.randn(N, D, generator=gen).requires_grad_(True)
            vh = torch.randn(G, D, generator=gen).requires_grad_(True)
            g = FakeGraph(src, dst, N)
            g.batch_num_nodes = list(sizes)
            vn_out, h_out = vn(g, h, vh)
            ct_v = torch.randn(G, D, generator=torch.Generator().manual_seed(3))
            ct_h = torch.randn(N, D, generator=torch.Generator().manual_seed(4))
            params = list(vn.parameters())
            grads = torch.autograd.grad([vn_out, h_out], [h, vh] + params, [ct_v, ct_h])
            out[f"{pre}/h"], out[f"{pre}/vn_h"] = h.detach().numpy(), vh.detach().numpy()
            out[f"{pre}/vn_out"], out[f"{pre}/h_out"] = vn_out.detach().numpy(), h_out.detach().numpy()
            out[f"{pre}/cot_v"], out[f"{pre}/cot_h"] = ct_v.numpy(), ct_h.numpy()
            out[f"{pre}/gh"], out[f"{pre}/gvn"] = grads[0].numpy(), grads[1].numpy()
            for (pn, _), gr in zip(vn.named_parameters(), grads[2:]):
                out[f"{pre}/gp::{pn}"] = gr.numpy()
            for k, v in vn.state_dict().items():
                if "running" in k:
                    out[f"{pre}/after::{k}"] = v.detach().numpy().copy()
            case += 1
    out["vn/n_cases"] = np.array(case)


def g10_net(out):
    """The whole graph-regression net of nets/molecules_graph_regression/dgn_net.py (embeddings, L towers layers with and
    without bond features, readout, MLPReadout, L1 loss): scores, loss, and every parameter gradient, in training mode."""
    from nets.molecules_graph_regression.dgn_net import DGNNet
    src, dst, N, sizes = make_test_graph(seed=5)
    out["src"], out["dst"], out["N"], out["sizes"] = src, dst, np.array(N), np.array(sizes)
    gen = torch.Generator().manual_seed(21)
    eig = torch.randn(N, 4, generator=gen)
    atoms = torch.randint(0, 6, (N,), generator=gen)
    bonds = torch.randint(0, 4, (len(src),), generator=gen)
    snorm = torch.rand(N, 1, generator=gen) + 0.5
    targets = torch.randn(len(sizes), 1, generator=gen)
    out["eig"], out["atoms"], out["bonds"], out["snorm"], out["targets"] = eig.numpy(), atoms.numpy(), bonds.numpy(), snorm.numpy(), targets.numpy()
    cases = [("towers_edge", "towers", True, "mean"), ("towers", "towers", False, "directional"), ("simple", "simple", False, "sum")]
    out["cases"] = np.array([c[0] for c in cases])
    for name, type_net, edge_feat, mode in cases:
        torch.manual_seed(7)
        params = dict(num_atom_type=6, num_bond_type=4, hidden_dim=20, out_dim=20, in_feat_dropout=0.0, dropout=0.0, L=3,
                      type_net=type_net, pos_enc_dim=0, readout=mode, graph_norm=True, batch_norm=True,
                      aggregators="mean max dir1-av dir1-dx", scalers="identity amplification", avg_d={"log": torch.tensor(1.1)},
                      residual=True, edge_feat=edge_feat, edge_dim=6 if edge_feat else 0, pretrans_layers=1, posttrans_layers=1, device="cpu")
        net = DGNNet(params)
        net.train(True)
        for k, v in net.state_dict().items():
            out[f"{name}/sd::{k}"] = v.detach().numpy().copy()
        g = FakeGraph(src, dst, N)
        g.batch_num_nodes = list(sizes)
        g.ndata["eig"] = eig
        scores = net(g, atoms, bonds if edge_feat else None, snorm, None)
        loss = net.loss(scores, targets)
        names = [k for k, q in net.named_parameters()]
        grads = torch.autograd.grad(loss, [q for _, q in net.named_parameters()], allow_unused=True)
        out[f"{name}/cfg"] = np.array([type_net, str(int(edge_feat)), mode])
        out[f"{name}/scores"], out[f"{name}/loss"] = scores.detach().numpy(), loss.detach().numpy()
        for k, gr in zip(names, grads):
            if gr is not None:
                out[f"{name}/gp::{k}"] = gr.numpy()
        for k, v in net.state_dict().items():
            if "running" in k:
                out[f"{name}/after::{k}"] = v.detach().numpy().copy()


def g9_laplacian(out):
    """Laplacian construction + eigenvector bookkeeping of ``MoleculeDGL.get_eig`` (data/molecules.py:100-116), UNMODIFIED,
    driven by a fake graph that supplies the three DGL methods it calls.  ``scipy.sparse.linalg.eigs`` is ARPACK with
    tol=5e-1 and a random start vector -- irreproducible by construction -- so the solver call is intercepted: the matrix L
    the reference built is RECORDED (that pins the construction: in-degree clipping, the three normalisations, A's
    orientation) and an exact dense solve is handed back, which the reference then sorts / truncates / casts itself
    (that pins column order, the ``pos_enc_dim`` cut and the fp32 cast).  Unpinned (DGL 0.4.2 absent): the orientation of
    ``adjacency_matrix_scipy`` (rows = destinations, DGL's documented default) -- irrelevant for the symmetric graphs here."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    dgl = sys.modules["dgl"]
    dgl.backend = types.SimpleNamespace(asnumpy=lambda t: t.numpy() if torch.is_tensor(t) else np.asarray(t))
    dgl.DGLGraph = object
    import data.molecules as M

    class G:
        def __init__(self, src, dst, n):
            self.src, self.dst, self.n, self.ndata = np.asarray(src), np.asarray(dst), n, {}

        def number_of_nodes(self):
            return self.n

        def in_degrees(self):
            return torch.from_numpy(np.bincount(self.dst, minlength=self.n))

        def adjacency_matrix_scipy(self, return_edge_ids=False):
            return sp.coo_matrix((np.ones(len(self.src)), (self.dst, self.src)), shape=(self.n, self.n)).tocsr()

    recorded = []
    real_eigs = spl.eigs

    def fake_eigs(L, k, which="SR", tol=0):
        dense = np.asarray(L.todense())
        recorded.append(dense)
        w, v = np.linalg.eig(dense)
        idx = np.argsort(w.real)[:k]
        # (hand the pairs back in a scrambled order: the reference's own argsort must restore increasing order)
        idx = idx[::-1]
        return w[idx], v[:, idx]

    rng = np.random.default_rng(9)
    graphs = []
    for n in (9, 14, 23, 37):
        und = [(int(rng.integers(0, v)), v) for v in range(1, n)]
        for _ in range(3):
            a, b = sorted(int(x) for x in rng.integers(0, n, 2))
            if a != b and (a, b) not in und:
                und.append((a, b))
        und = np.asarray(und)
        graphs.append((np.concatenate([und[:, 0], und[:, 1]]), np.concatenate([und[:, 1], und[:, 0]]), n))
    out["n_graphs"] = np.array(len(graphs))
    out["pos_enc_dim"] = np.array(6)
    M.sp.linalg.eigs = fake_eigs
    try:
        for norm in ("none", "sym", "walk"):
            ds = types.SimpleNamespace(graph_lists=[G(*g) for g in graphs])
            recorded.clear()
            M.MoleculeDGL.get_eig(ds, pos_enc_dim=6, norm=norm)
            for i, (g, fg) in enumerate(zip(graphs, ds.graph_lists)):
                out[f"g{i}/src"], out[f"g{i}/dst"], out[f"g{i}/n"] = g[0], g[1], np.array(g[2])
                out[f"g{i}/{norm}/L"] = recorded[i]
                out[f"g{i}/{norm}/eig"] = fg.ndata["eig"].numpy()
    finally:
        M.sp.linalg.eigs = real_eigs


def main():
    _install_stubs()
    only = sys.argv[1:]
    for fname, fn in (("g1_aggregators", g1_aggregators), ("g2_scalers", g2_scalers), ("g3_reduce", g3_reduce),
                      ("g4_layers", g4_layers), ("g5_edge_cases", g5_edge_cases), ("g6_dense", g6_dense),
                      ("g8_readouts", g8_readouts), ("g9_laplacian", g9_laplacian), ("g10_net", g10_net)):
        if only and fname not in only:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, fname + ".npz")
        np.savez_compressed(path, **out)
        print(f"{fname}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
