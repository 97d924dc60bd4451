#!/usr/bin/env python3
"""Training step (forward + backward) of a simple layer on a power-law graph WITH hub rows: posttrans per in-degree class on the rows below
32 + folded product on the hubs (ops.DC_SPLIT_TRAINING, round 6) against the folded product on every row (the whole-layer call).
usage: tools/hub_training_time.py [N] [E] [F]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgn_amd
from dgn_amd import ops, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
F_ = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
indptr, src, eig = synth.powerlaw_csr(N, E, dev, seed=0)
graph = dgn_amd.DGNGraph.from_csr(indptr, src, eig=eig)
avg = float(graph.log_deg.mean().item())
aggs, scalers = "mean max min sum std dir1-dx dir2-dx dir3-dx", "identity amplification attenuation"
torch.manual_seed(0)
layer = dgn_amd.DGNLayer(F_, F_, 0.0, False, True, aggs, scalers, {"log": torch.tensor(avg)}, "simple", True, towers=1, edge_features=False,
                         edge_dim=0).model.to(dev).train()
h = torch.randn(N, F_, device=dev, requires_grad=True)
ct = torch.randn(N, F_, device=dev)
params = list(layer.parameters())
print(f"N={N} E={graph.num_edges} F={F_} hub rows={graph.n_hub_rows_dc()} ({100.0 * graph.n_hub_rows_dc() / N:.2f} %)")


def step():
    h.grad = None
    for p in params:
        p.grad = None
    layer(graph, h, None, None).backward(ct)


for on in (True, False, True, False):
    ops.DC_SPLIT_TRAINING = on
    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize(dev)
    print(f"split training {'on ' if on else 'off'}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step (layer forward + backward)")
