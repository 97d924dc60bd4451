"""The RCCL path on a real device: a process group with backend "nccl" (= RCCL on ROCm), world size 1, runs the
two collectives of dgn_amd.dist on device buffers -- the flat-gradient all-reduce after a real layer backward and
the differentiable row all-gather -- and bench.py's own multi-rank launch either runs N ranks or fails loudly.
(World size > 1 is covered by the gloo tests on CPU and by the driver's multi-GPU runs.)"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_NCCL_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, os.environ["DGN_ROOT"])
import torch, torch.distributed as dist
import dgn_amd
from dgn_amd import dist as ddist, synth
rank, world, local = ddist.init_from_env("nccl")
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local)
b = synth.molecule_batch(64, seed=3, laplacian_eig=False)
g = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), b["num_nodes"], eig=b["eig"].to(dev))
torch.manual_seed(0)
layer = dgn_amd.DGNLayer(20, 20, 0.0, True, True, "mean max dir1-dx dir1-av", "identity amplification attenuation",
                         {"log": torch.tensor(1.0)}, "towers", True, towers=5, edge_features=False, edge_dim=0).model.to(dev)
h = torch.randn(b["num_nodes"], 20, device=dev)
y = layer(g, h, None, b["snorm_n"].to(dev))
(y * y).mean().backward()
before = {n: p.grad.clone() for n, p in layer.named_parameters()}
red = ddist.FlatGradAllReduce(layer.parameters())
assert red.active and red.flat.is_cuda
red()                                                # RCCL all-reduce of the flat buffer, world 1: the identity
torch.cuda.synchronize()
ok_grad = all(torch.equal(p.grad, before[n]) for n, p in layer.named_parameters())
# differentiable row all-gather on device rows
rows = torch.randn(37, 8, device=dev, requires_grad=True)
full = ddist.all_gather_rows(rows, [(0, 37)])
w = torch.randn(37, 8, device=dev)
(full * w).sum().backward()
ok_gather = torch.equal(full.detach(), rows.detach()) and torch.allclose(rows.grad, w)
# SyncBN: the two all-reduces of dist.sync_batch_norm on device statistics (world 1: the sums of this rank) against F.batch_norm
x = torch.randn(501, 70, device=dev, requires_grad=True)
x2 = x.detach().clone().requires_grad_(True)
ga, be = torch.rand(70, device=dev) + 0.5, torch.randn(70, device=dev)
rm, rv, rm2, rv2 = torch.zeros(70, device=dev), torch.ones(70, device=dev), torch.zeros(70, device=dev), torch.ones(70, device=dev)
ct = torch.randn(501, 70, device=dev)
ys = ddist.sync_batch_norm(x, ga, be, rm, rv, 0.1, 1e-5)
yr = torch.nn.functional.batch_norm(x2, rm2, rv2, ga, be, True, 0.1, 1e-5)
(ys * ct).sum().backward(); (yr * ct).sum().backward()
ok_syncbn = (torch.allclose(ys, yr, rtol=1e-5, atol=1e-5) and torch.allclose(x.grad, x2.grad, rtol=1e-4, atol=1e-5)
             and torch.allclose(rm, rm2, atol=1e-6) and torch.allclose(rv, rv2, rtol=1e-5))
# ... and a layer holding the converted modules leaves the fused calls in training (process group up) with the same results
torch.manual_seed(0)
layer2 = dgn_amd.DGNLayer(20, 20, 0.0, True, True, "mean max dir1-dx dir1-av", "identity amplification attenuation",
                          {"log": torch.tensor(1.0)}, "towers", True, towers=5, edge_features=False, edge_dim=0).model.to(dev)
layer2.load_state_dict({k: v for k, v in layer.state_dict().items()}, strict=True)
for m in layer2.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.reset_running_stats()
ddist.convert_sync_batchnorm(layer2)
from dgn_amd import ops
assert ops._spans_ranks(layer2.towers[0].batchnorm_h, True) and not ops.bn_tail_supported([layer2.towers[0].batchnorm_h], h, True)
y2 = layer2(g, h, None, b["snorm_n"].to(dev))
(y2 * y2).mean().backward()
ok_sync_layer = torch.allclose(y2, y, rtol=1e-4, atol=1e-5) and all(
    torch.allclose(p.grad, before[n], rtol=2e-3, atol=1e-5 * max(1.0, float(before[n].abs().max()))) for n, p in layer2.named_parameters())
ms = ddist.barrier_max_ms(1.25, dev)
print("RESULT " + json.dumps(dict(ok_grad=bool(ok_grad), ok_gather=bool(ok_gather), ok_syncbn=bool(ok_syncbn), ok_sync_layer=bool(ok_sync_layer),
                                  ms=ms, backend=dist.get_backend(), world=dist.get_world_size())))
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_collectives_world1():
    env = dict(os.environ, DGN_ROOT=ROOT, DGN_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _NCCL_SCRIPT], env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res == dict(ok_grad=True, ok_gather=True, ok_syncbn=True, ok_sync_layer=True, ms=1.25, backend="nccl", world=1)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` with no launcher: on a box with >= 2 devices two ranks run over RCCL and the line says
    n_gpus 2; on a 1-GPU box the run is refused (never a 1-rank run labelled 2)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c2_b128", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=850)
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0, p.stderr[-4000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and "2 ranks" in line["config"]["parallelism"]
    else:
        assert p.returncode != 0 and "refusing" in p.stderr


def test_bench_refuses_more_ranks_than_devices():
    """CPU container (0 devices): the multi-rank request must fail before anything runs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    want = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(2, want))], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "refusing" in p.stderr
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")      # a launcher that started fewer ranks than --gpus says
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


_SHARDED_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, os.environ["DGN_ROOT"])
import numpy as np, torch, torch.distributed as dist
import dgn_amd
from dgn_amd import dist as ddist, synth
from dgn_amd.ops import directional_aggregate
from oracle import dgn_oracle as orc
rank, world, local = ddist.init_from_env("nccl")
assert dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
N, F_ = 1500, 16
indptr, src, eig = synth.powerlaw_csr(num_nodes=N, num_edges=24000, device=dev, seed=5)
E = int(indptr[-1])
aggs, scalers = "mean max min sum std dir1-dx dir2-dx dir3-dx".split(), "identity amplification attenuation".split()     # C5's list
plan = dgn_amd.make_plan(aggs, scalers)
deg = (indptr[1:] - indptr[:-1])
avg = float(torch.log(deg.float() + 1).mean())
gen = torch.Generator(device=dev).manual_seed(1)
X0 = torch.randn(N, F_, device=dev, generator=gen)
W1 = torch.randn(F_, plan.out_width(F_), device=dev, generator=gen) / plan.out_width(F_) ** 0.5
W2 = torch.randn(F_, plan.out_width(F_), device=dev, generator=gen) / plan.out_width(F_) ** 0.5
ct = torch.randn(N, F_, device=dev, generator=gen)
ranges = ddist.row_ranges_by_edges(indptr, 3)            # three destination-range shards, run one after the other by this one rank
shards = [ddist.shard_rows(indptr, src, r0, r1, hub_threshold=256, hub_chunk=64) for r0, r1 in ranges]
assert any(s.n_hub > 0 for s in shards)

def sharded_layer(H, W):
    # every shard: the sweep over its destination rows (features replicated), the F-wide post-transformation on its rows; the rows
    # then travel through the RCCL all-gather (world 1: one range) exactly as between two sharded layers of a net
    parts = []
    for (r0, r1), sh in zip(ranges, shards):
        y = directional_aggregate(sh, plan, avg, x_src=H, x_in=H[r0:r1], eig=eig)
        parts.append(torch.tanh(y @ W.t()))
    own = torch.cat(parts)
    return ddist.all_gather_rows(own, [(0, N)])

X = X0.clone().requires_grad_(True)
out = sharded_layer(sharded_layer(X, W1), W2)
(gX,) = torch.autograd.grad(out, X, ct)
torch.cuda.synchronize()

# the oracle: the same two layers on the WHOLE graph, CPU, fp32 and fp64
dst = torch.repeat_interleave(torch.arange(N, device=dev), deg).cpu()
srcc = src.long().cpu()
res = {}
for dt in (torch.float32, torch.float64):
    Xo = X0.cpu().to(dt).requires_grad_(True)
    H = Xo
    for W in (W1, W2):
        y = orc.aggregate_graph(srcc, dst, N, H[srcc], eig.cpu().to(dt), H, aggs, scalers, torch.tensor(avg, dtype=dt))
        H = torch.tanh(y @ W.cpu().to(dt).t())
    res[dt] = (H.detach(), torch.autograd.grad(H, Xo, ct.cpu().to(dt))[0])
rows = torch.unique(torch.cat([torch.randint(0, N, (200,)), torch.tensor([0, N - 1]), torch.tensor([r for rr in ranges for r in (rr[0], max(rr[1] - 1, 0))]),
                               torch.argsort(deg.cpu())[-5:]]))
def worst(ours, r32, r64):
    ours, r32, r64 = ours.cpu().double(), r32.double(), r64.double()
    scale = max(1.0, float(r64.abs().max()))
    tol = 1e-5 * scale + 1e-4 * r64.abs()
    ok = ((ours - r32).abs() <= tol) | ((ours - r64).abs() <= tol + 4 * float((r32 - r64).abs().max()))
    return int((~ok).sum()), float((ours - r64).abs().max()) / scale
bad_y, err_y = worst(out.detach()[rows.to(dev)], res[torch.float32][0][rows], res[torch.float64][0][rows])
bad_g, err_g = worst(gX, res[torch.float32][1], res[torch.float64][1])
print("RESULT " + json.dumps(dict(bad_y=bad_y, err_y=err_y, bad_g=bad_g, err_g=err_g, n_rows=int(rows.numel()), hubs=sum(s.n_hub for s in shards))))
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_row_sharded_two_layer_path_vs_oracle_through_rccl():
    """f4 (one giant graph cut into destination-range shards) against the ORACLE, not against the unsharded kernels: a two-layer
    stack over a power-law graph -- three row shards with hub rows, features replicated, the layer outputs exchanged through
    dist.all_gather_rows on a real RCCL process group (DGN_FORCE_DIST=1, world 1) -- forward on sampled rows (incl. the shard
    boundaries and the heaviest hubs) and the full input gradient against the oracle's whole-graph evaluation."""
    env = dict(os.environ, DGN_ROOT=ROOT, DGN_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _SHARDED_SCRIPT], env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["bad_y"] == 0 and res["bad_g"] == 0 and res["hubs"] > 0, res
    assert res["err_y"] < 1e-3 and res["err_g"] < 1e-3, res
