#!/usr/bin/env python3
"""Host-side (Python) cost of a layer step at a launch-bound batch size: cProfile over N steps of the c2_b128 workload."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgn_amd
from dgn_amd import synth
import bench

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2_b128"])
dev = torch.device("cuda")
batch, graph = bench.build_batch(wl, 41, dev)
F_ = wl["hidden"]
torch.manual_seed(0)
layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(1.0)}, wl["type_net"], True,
                         towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev)
N = graph.num_nodes
h = torch.randn(N, F_, device=dev, requires_grad=True)
snorm = batch["snorm_n"].to(dev)
ct = torch.randn(N, F_, device=dev)

params = list(layer.parameters())

def step():
    graph._wcache.clear()
    h.grad = None
    for p in params:
        p.grad = None
    layer(graph, h, None, snorm).backward(ct)

for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 5)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "tottime")).print_stats(34); print(s.getvalue()[:6000])
