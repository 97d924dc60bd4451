#!/usr/bin/env python3
"""cProfile of the host side of an eager layer step of a bench workload (which Python lines a launch-bound step spends its time in).
Usage: host_profile.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dgn_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2_b128"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = dict(bench.WORKLOADS[name])
dev = torch.device("cuda")
batch, graph = bench.build_batch(wl, 41, dev)
F_, N = wl["hidden"], graph.num_nodes
avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
layer = dgn_amd.DGNLayer(F_, F_, 0.0, wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)}, wl["type_net"],
                         True, towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).train()
h = torch.randn(N, F_, device=dev, requires_grad=True)
ct = torch.randn(N, F_, device=dev)
snorm = batch["snorm_n"].to(dev)
params = list(layer.parameters())


def step():
    graph._wcache.clear()
    h.grad = None
    for p in params:
        p.grad = None
    layer(graph, h, None, snorm).backward(ct)


for _ in range(30):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"{name}: per-step host totals (us) over {steps} steps")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
for (fn, line, func), (cc, nc, tt, ct_, callers) in rows:
    print(f"{tt / steps * 1e6:8.1f} us  calls/step {nc / steps:6.1f}  {os.path.basename(fn)}:{line} {func}")
