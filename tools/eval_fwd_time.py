#!/usr/bin/env python3
"""No-grad (evaluation) forward of a bench workload's layer: eager and captured."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dgn_amd  # noqa: E402
from dgn_amd.hipgraph import capture  # noqa: E402
dev = torch.device("cuda")
for name in sys.argv[1:] or ["c2_b128", "zinc_json_b128", "c1_b128"]:
    wl = dict(bench.WORKLOADS[name])
    batch, graph = bench.build_batch(wl, 41, dev)
    F_, N = wl["hidden"], graph.num_nodes
    avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)}, wl["type_net"],
                             True, towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).eval()
    h = torch.randn(N, F_, device=dev)
    snorm = batch["snorm_n"].to(dev)

    def step():
        graph._wcache.clear()
        with torch.no_grad():
            return layer(graph, h, None, snorm)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 300 * 1e3
    g = capture(step, warmup=3)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name}: eval forward eager {eager:.4f} ms, captured {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms", flush=True)
