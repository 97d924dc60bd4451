#!/usr/bin/env python3
"""Per-step wall time of the c2 step from a cold start (after the setup's idle time): how long the first steps take to reach the steady state."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dgn_amd
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["c2"])
batch, graph = bench.build_batch(wl, 41, dev)
F_, N = wl["hidden"], graph.num_nodes
avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
torch.manual_seed(0)
layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)}, "towers", True, towers=5,
                         edge_features=False, edge_dim=0).model.to(dev).train()
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(N, F_, device=dev, generator=gen).requires_grad_(True)
ct = torch.randn(N, F_, device=dev, generator=gen)
snorm = batch["snorm_n"].to(dev)
params = list(layer.parameters())
def step():
    graph._wcache.clear()
    h.grad = None
    for p in params:
        p.grad = None
    layer(graph, h, None, snorm).backward(ct)
for trial in range(2):
    torch.cuda.synchronize(); time.sleep(1.0)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
    evs[0].record()
    for i in range(80):
        step(); evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(80)]
    print("trial", trial, "steps 1-5:", " ".join(f"{t:.3f}" for t in ts[:5]), "| 6-25 mean", f"{sum(ts[5:25]) / 20:.4f}", "| 26-45", f"{sum(ts[25:45]) / 20:.4f}",
          "| 46-80", f"{sum(ts[45:]) / 35:.4f}")
