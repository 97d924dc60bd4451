#!/usr/bin/env python3
"""Fused sweep + posttrans forward (dgn_layer_fused_forward) vs the two kernels it replaces, on the c2 shape: max error, timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgn_amd
from dgn_amd import synth, ops
from dgn_amd.spec import X_IN_NAME

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
dev = torch.device("cuda")
b = synth.molecule_batch(n_graphs, seed=41, extra_bonds=3.9, laplacian_eig=False)
g = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), b["num_nodes"], eig=b["eig"].to(dev))
N, F, T, S, fo = g.num_nodes, 70, 5, 3, 14
plan = dgn_amd.make_plan(["mean", "max", "min", "dir1-av", "dir1-dx", X_IN_NAME], ["identity"])
K = plan.n_agg * (F // T)
gen = torch.Generator(device=dev).manual_seed(0)
pq = torch.randn(N, 2 * F, device=dev, generator=gen)
h = torch.randn(N, F, device=dev, generator=gen)
W = torch.randn(T, S * fo, K, device=dev, generator=gen) / K ** 0.5
sc = torch.rand(N, S, device=dev, generator=gen) + 0.5
bias = torch.randn(T * fo, device=dev, generator=gen)
rs = torch.rand(N, device=dev, generator=gen) + 0.5
w = g.edge_weights(plan)
print("supported:", ops.fused_sweep_posttrans_supported(g, plan, T, F, S, fo))

def unfused():
    aggx = ops.directional_aggregate(g, plan, 1.0, x_pair=pq, x_in=h, n_towers=T, weights=w, tower_major=True)
    z = ops.linear(aggx, W)
    return ops.scale_combine(z, sc, bias, rs)

def fused():
    return ops.fused_sweep_posttrans_forward(g, plan, T, 1.0, w, pq, h, W, sc, bias, rs)

with torch.no_grad():
    ya, yb = unfused(), fused()
    torch.cuda.synchronize()
    print("max abs diff", float((ya - yb).abs().max()), "scale", float(ya.abs().max()))
    for name, fn in (("unfused", unfused), ("fused", fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        print(name, "ms", a.elapsed_time(e) / 20)
