#!/usr/bin/env python3
"""A/B of a library option (dgn_set_option) on a bench workload inside ONE process: the step is timed `rounds` times per value,
alternating (box-to-box and run-to-run noise on this pool is +-2 %: separate processes cannot resolve a 20-us change).
usage: tools/ab_option.py <option> <value_a> <value_b> [--workload c2] [--steps 50] [--rounds 6] [--kernels]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dgn_amd
from dgn_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("option"); ap.add_argument("a", type=int); ap.add_argument("b", type=int)
ap.add_argument("--workload", default="c2"); ap.add_argument("--steps", type=int, default=50); ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--kernels", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS[args.workload])
batch, graph = bench.build_batch(wl, 41, dev)
F_, N = wl["hidden"], graph.num_nodes
avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
torch.manual_seed(0)
layer = dgn_amd.DGNLayer(F_, F_, wl.get("dropout", 0.0), wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)},
                         wl["type_net"], True, towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).train()
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

def timed(n):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3 / n

res = {args.a: [], args.b: []}
for v in (args.a, args.b):
    setattr(_lib.options, args.option, v)
    timed(10)
for r in range(args.rounds):
    for v in (args.a, args.b):
        setattr(_lib.options, args.option, v)
        timed(3)
        res[v].append(timed(args.steps))
for v, t in res.items():
    t = sorted(t)
    print(f"{args.option}={v}: median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}  max {t[-1]:.4f}  ({args.workload}, {args.steps} steps x {args.rounds} rounds)")
if args.kernels:
    for v in (args.a, args.b):
        setattr(_lib.options, args.option, v)
        tab = bench.step_kernel_table(step, dev, steps=10)
        print(f"--- {args.option}={v}: kernels per step (us)")
        for row in tab[:28]:
            print(f"  {row['us_per_step']:9.2f}  x{row['calls_per_step']:.0f}  {row['kernel'][:110]}")
        print(f"  sum {sum(r['us_per_step'] for r in tab):.1f}")
