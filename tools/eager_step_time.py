#!/usr/bin/env python3
"""Wall time of an eager layer step of a bench workload, autograd's device thread on / off (torch.autograd.set_multithreading_enabled).
Usage: eager_step_time.py [workload] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dgn_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2_b128"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
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


def step(keep_grads):
    graph._wcache.clear()
    if not keep_grads:
        h.grad = None
        for p in params:
            p.grad = None
    layer(graph, h, None, snorm).backward(ct)


for mt in (True, False):
    for keep in (False, True):
        torch.autograd.set_multithreading_enabled(mt)
        for _ in range(50):
            step(keep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(keep)
        torch.cuda.synchronize()
        print(f"{name}: autograd multithreading {mt}, grads {'accumulated' if keep else 'reset to None'}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms / step")
