#!/usr/bin/env python3
"""Which host-side op launches each SMALL device activity (< 15 us) of a layer step: torch.profiler, device activities joined to the
innermost CPU op whose time range contains the launch (by correlation through the profiler's kineto events)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dgn_amd
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda", 0)
tag = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = dict(bench.WORKLOADS[tag])
batch, graph = bench.build_batch(wl, 41, dev)
F_, N, E = wl["hidden"], graph.num_nodes, graph.num_edges
torch.manual_seed(0)
avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
layer = dgn_amd.DGNLayer(F_, F_, wl.get("dropout", 0.0), wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"],
                         {"log": torch.tensor(avg_log)}, wl["type_net"], True, towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev)
layer.train()
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
for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
evs = prof.profiler.kineto_results.events()
cpu_ops = [(e.start_ns(), e.start_ns() + e.duration_ns(), e.name()) for e in evs if str(e.device_type()).endswith("CPU") and "hipLaunch" not in e.name() and "hipMemcpy" not in e.name() and "hipExtModule" not in e.name()]
launches = {e.correlation_id(): e for e in evs if str(e.device_type()).endswith("CPU") and ("hipLaunch" in e.name() or "hipMemcpy" in e.name() or "hipExtModule" in e.name() or "hipMemset" in e.name())}
count = collections.Counter()
for e in evs:
    if not str(e.device_type()).endswith("CUDA") or e.duration_ns() > 15000:
        continue
    l = launches.get(e.correlation_id())
    parent = "?"
    if l is not None:
        t = l.start_ns()
        inner = [(b - a, n) for a, b, n in cpu_ops if a <= t <= b]
        inner.sort()
        parent = " < ".join(n for _, n in inner[:3])
    count[(e.name()[:70], parent[:150])] += 1
for (k, par), c in sorted(count.items(), key=lambda kv: -kv[1]):
    print(f"{c / 3:5.1f}/step  {k:70s}  <- {par}")
