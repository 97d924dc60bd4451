#!/usr/bin/env python3
"""Phase durations inside blk_forward / blk_backward (wall-clock stamps, csrc/dgn_blk_layer_kernels.hpp: BLK_STAMP) for a bench workload.
Usage: blk_phases.py [workload]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dgn_amd  # noqa: E402
from dgn_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2_b128"
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
for it in range(3):
    ops._BLK_DBG = {} if it == 2 else None
    layer(graph, h, None, snorm).backward(ct)
torch.cuda.synchronize()
d = ops._BLK_DBG
ops._BLK_DBG = None
t = graph.block_table()
print(f"{name}: N={N} blocks={t['n_blocks']} max_rows={t['max_rows']} max_edges={t['max_edges']}")
for key in ("t_fwd", "t_bwd"):
    ts = d[key].cpu().double()
    t0 = ts[:, 0:1]
    rel = (ts - t0) * 0.01          # 100 MHz -> us
    rel[ts == 0] = float("nan")
    mean = torch.nanmean(rel, dim=0)
    mx = torch.nan_to_num(rel, nan=0.0).max(dim=0).values
    span = (ts[ts > 0].max() - ts[ts > 0].min()) * 0.01
    starts = (ts[:, 0] - ts[:, 0].min()) * 0.01
    nb = t["n_blocks"]
    print(key, "start offsets of the towers' workgroups (mean per tower):", [round(float(starts[i * nb:(i + 1) * nb].mean()), 2) for i in range(ts.shape[0] // nb)],
          "latest start:", round(float(starts.max()), 2))
    print(key, "mean us since kernel start per stamp:", [None if m != m else round(float(m), 2) for m in mean])
    print(key, "max:", [round(float(m), 2) for m in mx], "first start -> last stamp:", round(float(span), 2))
