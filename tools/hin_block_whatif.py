#!/usr/bin/env python3
"""What the h_in pass-through block costs the forward sweep: the c2 sweep, tower-major, with the six-block list
(aggregators + __x_in__) against the five-block list, same inputs; and the backward sweep likewise."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dgn_amd
from dgn_amd import ops
from dgn_amd.ops import launch_forward, launch_backward
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["c2"])
batch, graph = bench.build_batch(wl, 41, dev)
N, F_, T = graph.num_nodes, 70, 5
avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
torch.manual_seed(0)
layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)}, "towers", True, towers=5,
                         edge_features=False, edge_dim=0).model.to(dev)
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(N, F_, device=dev, generator=gen)
pq = torch.randn(N, 2 * F_, device=dev, generator=gen)
for name, plan in (("six blocks (with __x_in__)", layer._kplan_x), ("five blocks", layer._kplan)):
    w = graph.edge_weights(plan)
    K = plan.out_width(F_) // T
    Np = N + (N & 1)                                            # (tower planes 16-byte aligned also for K = 70)
    out = torch.empty(T, Np, K, device=dev)[:, :N]
    g_out = torch.randn(T, Np, K, device=dev, generator=gen)[:, :N]
    n_aux = ops.agg_aux_bytes(graph, plan, T, F_, pq[:, :F_], pq[:, F_:], None, h)
    aux = torch.empty(n_aux, dtype=torch.uint8, device=dev) if n_aux else None
    g_src, g_dst, g_in = (torch.zeros(N, F_, device=dev) for _ in range(3))
    f = lambda: launch_forward(graph, plan, T, avg_log, w, pq[:, :F_], pq[:, F_:], None, h, out, aux=aux)
    b = lambda: launch_backward(graph, plan, T, avg_log, w, pq[:, :F_], pq[:, F_:], None, h, g_out, g_src, g_dst, None, g_in, accumulate=False, aux=aux)
    sf, sb = bench.event_stats(f, dev), bench.event_stats(b, dev)
    print(f"{name}: forward {sf['median'] * 1e3:.1f} us, backward {sb['median'] * 1e3:.1f} us  (out {4 * T * N * K / 1e6:.0f} MB)")
