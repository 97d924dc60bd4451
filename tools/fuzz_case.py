#!/usr/bin/env python3
"""One case of tests/test_block_layer_fuzz_gpu.py in detail: both routes and the oracle, per tensor.  Usage: fuzz_case.py <seed> [aggs override]"""
import importlib.util, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dgn_amd
from dgn_amd import synth
from oracle import dgn_oracle as orc
spec = importlib.util.spec_from_file_location("f", os.path.join(ROOT, "tests", "test_block_layer_fuzz_gpu.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
seed = int(sys.argv[1])
c = m._case(seed)
if len(sys.argv) > 2:
    c["aggs"] = sys.argv[2]
if len(sys.argv) > 3:
    c["scalers"] = sys.argv[3]
print(c)
dev = torch.device("cuda")
b = synth.knn_batch(max(2, c["n_graphs"] // 4), seed=c["gseed"], n_lo=12, n_hi=40, k=4) if c["knn"] else synth.molecule_batch(c["n_graphs"], seed=c["gseed"], extra_bonds=3.9, eig_dim=6)
N = int(b["num_nodes"])
if b["eig"].shape[1] < 4:
    b["eig"] = torch.cat([b["eig"], torch.randn(N, 4 - b["eig"].shape[1], generator=torch.Generator().manual_seed(seed))], dim=1)
avg = float(torch.log(torch.bincount(b["dst"], minlength=N).float() + 1).mean())
torch.manual_seed(seed)
layer = dgn_amd.DGNLayer(c["F"], c["F"], 0.0, c["graph_norm"], True, c["aggs"], c["scalers"], {"log": torch.tensor(avg)}, c["type_net"], True, towers=c["T"], edge_features=False, edge_dim=0).model
gen = torch.Generator().manual_seed(seed)
with torch.no_grad():
    for p in layer.parameters():
        if p.dim() == 2:
            p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
        else:
            p.add_(0.1 * torch.randn(p.shape, generator=gen))
layer = layer.to(dev).train()
sd0 = {k: v.clone() for k, v in layer.state_dict().items()}
graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), N, eig=b["eig"].to(dev))
h, ct, snorm = torch.randn(N, c["F"], generator=gen).to(dev), torch.randn(N, c["F"], generator=gen).to(dev), b["snorm_n"].to(dev)
dgn_amd.ops.BLOCK_LAYER_MAX_POST = 1 << 30
def run(max_nodes):
    dgn_amd.ops.BLOCK_LAYER_MAX_NODES = max_nodes
    layer.load_state_dict(sd0)
    hh = h.clone().requires_grad_(True)
    y = layer(graph, hh, None, snorm)
    g = torch.autograd.grad(y, [hh] + list(layer.parameters()), ct)
    return dict(zip(["y", "h"] + [k for k, _ in layer.named_parameters()], [y.detach()] + list(g)))
rb, rs = run(1 << 20), run(0)
ref = {}
for dtype in (torch.float32, torch.float64):
    sd = {k: (v.detach().cpu().to(dtype).requires_grad_("running" not in k) if v.dtype.is_floating_point else v.cpu().clone()) for k, v in sd0.items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and v.requires_grad]
    cfg = dict(aggregators=c["aggs"], scalers=c["scalers"], avg_log=torch.tensor(avg, dtype=dtype), graph_norm=c["graph_norm"], batch_norm=True, residual=True, towers=c["T"], divide_input=True, edge_features=False)
    hh = h.cpu().to(dtype).requires_grad_(True)
    y, _ = orc.layer_forward(c["type_net"], sd, cfg, b["src"], b["dst"], N, b["eig"].to(dtype), hh, None, b["snorm_n"].to(dtype), training=True)
    g = torch.autograd.grad(y, [hh] + [sd[k] for k in names], ct.cpu().to(dtype))
    ref[dtype] = dict(zip(["y", "h"] + names, [y.detach()] + list(g)))
for k in rb:
    r64 = ref[torch.float64][k]
    e = lambda t: float((t.cpu().double() - r64).abs().max())
    nb = int(((rb[k] - rs[k]).abs() > 5e-5 * max(1.0, float(rs[k].abs().max())) + 2e-4 * rs[k].abs()).sum())
    print(f"{k:55s} scale {float(r64.abs().max()):9.3g}  block {e(rb[k]):.3e}  streaming {e(rs[k]):.3e}  fp32 oracle {e(ref[torch.float32][k]):.3e}  entries block != streaming: {nb} / {rb[k].numel()}")
d = (rb["h"].cpu().double() - ref[torch.float64]["h"]).abs().max(dim=1).values
rows = torch.nonzero(d > 1e-3).flatten()
deg = torch.bincount(b["dst"], minlength=N)
odeg = torch.bincount(b["src"], minlength=N)
print("rows of d h off (block):", rows[:20].tolist(), "in-degree", deg[rows[:20]].tolist(), "out-degree", odeg[rows[:20]].tolist(), "of N", N)
for k in ("batchnorm_h.bias", "batchnorm_h.weight", "posttrans.fully_connected.0.linear.bias", "pretrans.fully_connected.0.linear.bias"):
    if k in rb:
        d = (rb[k].cpu().double() - ref[torch.float64][k]).abs()
        idx = torch.nonzero(d > 1e-3).flatten().tolist()
        print(k, "wrong columns:", idx[:20], [round(float(rb[k][i]), 4) for i in idx[:6]], "expected", [round(float(ref[torch.float64][k][i]), 4) for i in idx[:6]])
k = "posttrans.fully_connected.0.linear.weight"
d = (rb[k].cpu().double() - ref[torch.float64][k]).abs()
bad = torch.nonzero(d > 1e-3)
print("posttrans weight wrong entries: rows", sorted(set(bad[:, 0].tolist()))[:30], "cols min/max", int(bad[:, 1].min()), int(bad[:, 1].max()), "n", bad.shape[0])
k = "pretrans.fully_connected.0.linear.weight"
d = (rb[k].cpu().double() - ref[torch.float64][k]).abs()
bad = torch.nonzero(d > 1e-3)
print("pretrans weight wrong entries: rows", sorted(set(bad[:, 0].tolist()))[:30], "cols min/max", int(bad[:, 1].min()), int(bad[:, 1].max()), "n", bad.shape[0])
sizes = [int(x) for x in b["sizes"]]
cuts = np.concatenate([[0], np.cumsum(sizes)])
print("graph sizes", sizes)
print("graphs of wrong d h rows:", sorted(set(int(np.searchsorted(cuts, r, side="right") - 1) for r in rows.tolist())))
# expected d beta for simple / complex layers: sum over rows of ct where BatchNorm's output is positive (y = relu(y1) + h)
if c["type_net"] != "towers":
    yb = rb["y"]
    mask = (yb - h) > 0
    exp = (ct * mask).sum(0)
    got = rb["batchnorm_h.bias"]
    col = int((got - exp).abs().argmax())
    print("d beta: worst column", col, "got", float(got[col]), "sum ct*mask", float(exp[col]), "oracle", float(ref[torch.float64]["batchnorm_h.bias"][col]))
    diff = float(got[col] - exp[col])
    cand = torch.nonzero(((ct[:, col] - diff).abs() < 2e-3) | ((ct[:, col] + diff).abs() < 2e-3)).flatten().tolist()
    print("rows whose cotangent at that column equals the difference:", cand, [(float(ct[r, col]), bool(mask[r, col]), float((yb - h)[r, col])) for r in cand[:5]])
    for r in cand[:3]:
        g = int(np.searchsorted(cuts, r, side="right") - 1)
        print("row", r, "graph", g, "size", sizes[g], "row in graph", r - int(cuts[g]), "tail workgroup", r // 16, "row in it", r % 16)
