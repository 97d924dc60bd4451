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
ms = ddist.barrier_max_ms(1.25, dev)
print("RESULT " + json.dumps(dict(ok_grad=bool(ok_grad), ok_gather=bool(ok_gather), ms=ms, backend=dist.get_backend(),
                                  world=dist.get_world_size())))
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
    assert res == dict(ok_grad=True, ok_gather=True, ms=1.25, backend="nccl", world=1)


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
