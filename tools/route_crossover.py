#!/usr/bin/env python3
"""Captured layer step, graph-block route vs streaming kernels, over batch sizes (where ops.BLOCK_LAYER_MAX_NODES should sit)."""
import argparse, copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dgn_amd import ops  # noqa: E402
dev = torch.device("cuda")
base = sys.argv[1] if len(sys.argv) > 1 else "c2_b128"
for n_graphs in (128, 256, 512, 1024, 1400):
    out = []
    for lim in (1 << 30, 0):
        ops.BLOCK_LAYER_MAX_NODES = lim
        wl = copy.deepcopy(bench.WORKLOADS[base])
        wl["gen"] = (wl["gen"][0], dict(wl["gen"][1], n_graphs=n_graphs))
        args = argparse.Namespace(steps=100, warmup=20, hipgraph=True, scaling=None, workload=base, no_cpu_baseline=True, aggregators=None, scalers=None)
        r, _ = bench.run_layer_workload(args, wl, 0, 1, dev, steps=100, warmup=20, tag=None)
        out.append(r["ms_per_step"])
    print(f"{base} x {n_graphs} graphs ({r['nodes_per_rank']} nodes): block route {out[0]:.4f} ms, streaming {out[1]:.4f} ms", flush=True)
