#!/usr/bin/env python3
"""Forward-sweep timing at the c2 shape for output layouts (tower-major vs node-major) and tower counts (experiment helper)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgn_amd
from dgn_amd import synth, ops
from dgn_amd.spec import make_plan, X_IN_NAME

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = "cuda"
b = synth.molecule_batch(12000, seed=41)
g = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), b["num_nodes"], eig=b["eig"].to(dev))
N = b["num_nodes"]
F = int(os.environ.get("F", 70))
pq = torch.randn(N, 2 * F, device=dev)
h = torch.randn(N, F, device=dev)
plan = make_plan("mean max min dir1-av dir1-dx".split() + [X_IN_NAME], ["identity"])
w = g.edge_weights(plan)
with torch.no_grad():
    for T, tm in ((5, True), (5, False), (1, False)):
        if F % T: continue
        t = timeit(lambda: ops.directional_aggregate(g, plan, 1.0, x_src=pq[:, :F], x_dst=pq[:, F:], x_in=h, n_towers=T, weights=w, tower_major=tm))
        print(f"F={F} towers={T} tower_major={tm}: {t:.0f} us")
