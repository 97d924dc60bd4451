#!/usr/bin/env python3
"""Micro-timing of the layer-tail streaming kernels at the c2 shape (experiment helper)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgn_amd import ops

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

N, T, S, fo = 275167, 5, 3, 14
dev = "cuda"
z = torch.randn(T, N, S * fo, device=dev, requires_grad=True)
sc = torch.rand(N, S, device=dev) + 0.5
rs = torch.rand(N, device=dev) + 0.5
bias = torch.randn(T * fo, device=dev, requires_grad=True)
gy = torch.randn(N, T * fo, device=dev)
x = torch.randn(N, T * fo, device=dev, requires_grad=True)
bn = torch.nn.BatchNorm1d(T * fo).to(dev)
res = {}
with torch.no_grad():
    res["combine_fwd"] = timeit(lambda: ops.scale_combine(z, sc, bias, rs))
y = ops.scale_combine(z, sc, bias, rs)
res["combine_bwd+bias"] = timeit(lambda: torch.autograd.grad(y, [z, bias], gy, retain_graph=True))
zb = ops.scale_combine(z, sc, bias.detach(), rs)
res["combine_bwd"] = timeit(lambda: torch.autograd.grad(zb, [z], gy, retain_graph=True))
with torch.no_grad():
    res["bn_fwd(stats+apply)"] = timeit(lambda: ops.bn_tail(x, bn, True))
yb = ops.bn_tail(x, bn, True)
res["bn_bwd(stats+apply)"] = timeit(lambda: torch.autograd.grad(yb, [x, bn.weight, bn.bias], gy, retain_graph=True))
print(os.environ.get("DGN_EXP_ROWS"), os.environ.get("DGN_EXP_BNROWS"), " ".join(f"{k}={v:.0f}us" for k, v in res.items()))
