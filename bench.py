#!/usr/bin/env python3
"""DGN-layer forward+backward throughput on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c2_b128|c1|c3|c4|c5] [--scaling weak|strong]

With --gpus N > 1 and no launcher environment (RANK unset) the script re-executes itself under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` -- one rank per GPU over
RCCL -- and refuses to run if fewer than N devices are visible.

A "step" is one pass of the hot path over one batch of synthetic input: per-edge directional weights
from eig, one DGN layer forward, one backward (fixed random cotangent), and -- for N > 1 -- the flat
gradient all-reduce.  Inputs (graph CSR, features, eig) are resident in HBM before the timed region.
The default workload is BASELINE.json configs[1]: ZINC-12k (all 12 000 molecules as one batch),
DGN towers (5 towers, hidden 70, mean/max/min/dir1-av/dir1-dx, 3 PNA scalers).  For N > 1 every rank
processes its own 12k-molecule batch (weak scaling: seed 41 + rank), or -- ``--scaling strong`` -- its edge-balanced
shard of ONE global batch (``dist.shard_by_edges``; SURVEY 8(e): ogbg-molhiv batch 2048 split over the ranks).
The default single-GPU run also appends short sub-results (``extra``) for the other BASELINE configs (c1, c3, c4, c5) and for the
headline layer and the shipped ZINC json layer AT THE REFERENCE'S OWN BATCH SIZE (c2_b128, zinc_json_b128: 128 molecules, the graph-block
route of csrc/dgn_blk_layer.hip), the batch-sized legs both eager and as a captured HIP graph (``captured_ms_per_step``), and the C5 graph
through a whole simple layer forward (c5_layer: row f1; its ``frac`` is of the fp32 MFMA peak on the posttrans product); ``--all-extras``
adds every other leg.

Besides the contract keys the line carries
  roofline      the dominant aggregation kernel's algorithmic bytes / its measured launch duration
                (HIP events on the launching stream) against the 8 TB/s HBM peak
  cpu_baseline  the oracle (reference-structured CPU restatement) timed on the host cores on a bounded
                sample of the same workload -- rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dgn_amd  # noqa: E402
from dgn_amd import dist as ddist  # noqa: E402
from dgn_amd import synth  # noqa: E402
from dgn_amd.ops import launch_backward, launch_forward  # noqa: E402

AUX_LAYERS = ("towers", "simple", "complex")    # layer types whose whole-layer call passes the sweep's aux table from the forward to the backward
HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (generator kwargs, layer config)
    "c2": dict(desc="ZINC-12k (12000 molecules, one batch), DGN towers: 5 towers, hidden 70, "
                    "mean max min dir1-av dir1-dx x identity amplification attenuation",
               gen=("molecules", dict(n_graphs=12000, extra_bonds=3.9, eig_dim=6)), type_net="towers", hidden=70,
               aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation", towers=5),
    "c2c": dict(desc="ZINC-12k, DGN complex (the type_net of configs/molecules_graph_regression_DGN_ZINC.json): hidden 70, one tower, "
                     "mean max min dir1-av dir1-dx x identity amplification attenuation",
                gen=("molecules", dict(n_graphs=12000, extra_bonds=3.9, eig_dim=6)), type_net="complex", hidden=70,
                aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation", towers=1),
    "c2e": dict(desc="ZINC-12k towers as c2 WITH edge features (edge_dim 10: pretrans on [h_src || h_dst || ef])",
                gen=("molecules", dict(n_graphs=12000, extra_bonds=3.9, eig_dim=6)), type_net="towers", hidden=70,
                aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation", towers=5, edge_dim=10),
    "c2et": dict(desc="ZINC-12k towers as c2e, the edge features given as an embedding lookup (4 bond types x edge_dim 10, "
                      "dgn_net.py:53,75): EdgeTypeFeatures -> a 4 x 70 table in the sweep instead of an [E, 70] term",
                 gen=("molecules", dict(n_graphs=12000, extra_bonds=3.9, eig_dim=6)), type_net="towers", hidden=70,
                 aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation", towers=5, edge_dim=10,
                 edge_types=4),
    "c2_b128": dict(desc="ZINC batch of 128 molecules, DGN towers (as c2)",
                    gen=("molecules", dict(n_graphs=128, extra_bonds=3.9, eig_dim=6)), type_net="towers", hidden=70,
                    aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation", towers=5),
    "c1_b128": dict(desc="ZINC batch of 128 molecules, DGN simple as c1 (BASELINE configs[0] at the reference's batch size)",
                    gen=("molecules", dict(n_graphs=128, extra_bonds=3.9, eig_dim=6)), type_net="simple", hidden=75,
                    aggregators="mean dir1-dx-no-abs", scalers="identity amplification attenuation", towers=1),
    "c1": dict(desc="ZINC-12k, DGN simple: hidden 75, mean dir1-dx-no-abs x 3 scalers",
               gen=("molecules", dict(n_graphs=12000, extra_bonds=3.9, eig_dim=6)), type_net="simple", hidden=75,
               aggregators="mean dir1-dx-no-abs", scalers="identity amplification attenuation", towers=1),
    "c3": dict(desc="CIFAR10-superpixel-like, 128 graphs, 8-NN, DGN simple hidden 65, mean dir1-dx dir2-dx, identity",
               gen=("knn", dict(n_graphs=128)), type_net="simple", hidden=65,
               aggregators="mean dir1-dx dir2-dx", scalers="identity", towers=1),
    "c4": dict(desc="ogbg-molhiv-like, batch 2048, DGN simple hidden 70, mean max min dir1-dx dir1-av x 3 scalers",
               gen=("molecules", dict(n_graphs=2048, n_lo=10, n_hi=41, extra_bonds=4.3, eig_dim=4)), type_net="simple",
               hidden=70, aggregators="mean max min dir1-dx dir1-av", scalers="identity amplification attenuation", towers=1,
               graph_norm=False),      # configs/molecules_graph_classification_DGN_HIV.json:30
    # SURVEY 8(f) rank 1 / VERDICT r03 item 4: the C5 graph through a whole simple LAYER forward (sweep -> posttrans 3072 -> 128 -> tail)
    "c5_layer": dict(desc="power-law 10M / 200M, DGN simple layer forward (no grad): 8 aggregators x 3 scalers, hidden 128, posttrans + BatchNorm(eval) + ReLU + residual",
                     gen=("powerlaw", dict(num_nodes=10_000_000, num_edges=200_000_000)), type_net="layer_fwd", hidden=128,
                     aggregators="mean max min sum std dir1-dx dir2-dx dir3-dx", scalers="identity amplification attenuation", towers=1),
    # the reference's shipped dropout (configs HIV / CIFAR10 json: "dropout": 0.3; nets/dgn_layer.py:130,201): the same layers with it on
    "c3_drop": dict(desc="c3 with the CIFAR10 json's dropout 0.3 (bit-mask dropout kernels behind the whole-layer call)",
                    gen=("knn", dict(n_graphs=128)), type_net="simple", hidden=65, aggregators="mean dir1-dx dir2-dx", scalers="identity",
                    towers=1, dropout=0.3),
    "c4_drop": dict(desc="c4 with the HIV json's dropout 0.3",
                    gen=("molecules", dict(n_graphs=2048, n_lo=10, n_hi=41, extra_bonds=4.3, eig_dim=4)), type_net="simple",
                    hidden=70, aggregators="mean max min dir1-dx dir1-av", scalers="identity amplification attenuation", towers=1,
                    graph_norm=False, dropout=0.3),
    # "also run a saturating mega-batch so the roofline fraction is not just launch latency"
    "c3_mega": dict(desc="CIFAR10-superpixel-like, 8192 graphs in one batch (8-NN rows at a saturating size), layer as c3",
                    gen=("knn", dict(n_graphs=8192)), type_net="simple", hidden=65,
                    aggregators="mean dir1-dx dir2-dx", scalers="identity", towers=1),
    "c4_mega": dict(desc="ogbg-molhiv-like, the whole 41 127-graph dataset in one batch, layer as c4",
                    gen=("molecules", dict(n_graphs=41127, n_lo=10, n_hi=41, extra_bonds=4.3, eig_dim=4)), type_net="simple",
                    hidden=70, aggregators="mean max min dir1-dx dir1-av", scalers="identity amplification attenuation", towers=1,
                    graph_norm=False),
    # the reference's SHIPPED json configs, as written (odd hidden sizes: the layers pad the message path by one zero column)
    "zinc_json": dict(desc="configs/molecules_graph_regression_DGN_ZINC.json as shipped: complex, hidden 45, mean dir1-dx dir1-av x 3 scalers, "
                           "graph norm, ZINC-12k in one batch",
                      gen=("molecules", dict(n_graphs=12000, extra_bonds=3.9, eig_dim=6)), type_net="complex", hidden=45,
                      aggregators="mean dir1-dx dir1-av", scalers="identity amplification attenuation", towers=1),
    "zinc_json_b128": dict(desc="the same ZINC json layer at the json's batch size (128 molecules)",
                           gen=("molecules", dict(n_graphs=128, extra_bonds=3.9, eig_dim=6)), type_net="complex", hidden=45,
                           aggregators="mean dir1-dx dir1-av", scalers="identity amplification attenuation", towers=1),
    # BASELINE C4's layer (the HIV json's list WITH the three PNA scalers BASELINE.json names: "DGN + PNA scalers") at the json's batch size
    # and dropout: 73 500 posttrans weights, over ops.BLOCK_LAYER_MAX_POST -> streaming whole-layer route + bit-mask dropout
    "c4_b128": dict(desc="BASELINE C4's layer (simple, hidden 70, mean max min dir1-dx dir1-av x identity amplification attenuation) at batch 128 with the HIV json's "
                         "dropout 0.3 (streaming whole-layer route: its 73 500-weight posttrans is over the graph-block route's limit)",
                    gen=("molecules", dict(n_graphs=128, n_lo=10, n_hi=41, extra_bonds=4.3, eig_dim=4)), type_net="simple", hidden=70,
                    aggregators="mean max min dir1-dx dir1-av", scalers="identity amplification attenuation", towers=1, graph_norm=False, dropout=0.3),
    # the json AS SHIPPED (configs/molecules_graph_classification_DGN_HIV.json:23-34): scalers "identity" -> posttrans 5 x 70 x 70 = 24 500
    # weights: the graph-block route (round 5 benched the 3-scaler layer under this name)
    "hiv_json_b128": dict(desc="configs/molecules_graph_classification_DGN_HIV.json as shipped: simple, hidden 70, mean max min dir1-dx dir1-av x identity, "
                               "no graph norm, dropout 0.3, batch 128 (graph-block route + bit-mask dropout)",
                          gen=("molecules", dict(n_graphs=128, n_lo=10, n_hi=41, extra_bonds=4.3, eig_dim=4)), type_net="simple", hidden=70,
                          aggregators="mean max min dir1-dx dir1-av", scalers="identity", towers=1, graph_norm=False, dropout=0.3),
    "pattern_json": dict(desc="configs/SBMs_node_clustering_DGN_PATTERN.json as shipped: complex, hidden 47, mean dir1-dx dir2-dx x 3 scalers, "
                              "batch 128 of SBM graphs (~119 nodes, ~6.1 k directed edges each)",
                         gen=("sbm", dict(n_graphs=128)), type_net="complex", hidden=47,
                         aggregators="mean dir1-dx dir2-dx", scalers="identity amplification attenuation", towers=1),
    "c5": dict(desc="power-law 10M nodes / 200M edges, k=4 eig, hidden 128, forward only: "
                    "mean max min sum std dir1-dx dir2-dx dir3-dx x 3 scalers",
               gen=("powerlaw", dict(num_nodes=10_000_000, num_edges=200_000_000)), type_net="op", hidden=128,
               aggregators="mean max min sum std dir1-dx dir2-dx dir3-dx", scalers="identity amplification attenuation",
               towers=1),
}


def algorithmic_bytes(N, E, F, A, S, Ku, x, r):
    """SURVEY.md section 8(d) / BASELINE.md section 4: fp32 values, int32 indices, gather model."""
    fwd = E * (4 + 4 * F + 4 * Ku) + N * (4 + 4 * Ku + 4 * F * x + 4 * A * S * F)
    bwd = E * (4 + 4 * Ku + 4 * F * r + 4 * F) + N * (4 + 4 * Ku + 4 * A * S * F + 4 * F * x)
    return fwd, bwd


def plan_model(plan):
    ops = [op for l in plan.launches for op in l.ops]
    x = int(any(op in (8, 9, 10) for op in ops))
    r = int(any(op in (2, 3, 4, 5) for op in ops))
    return plan.n_agg, plan.n_scalers, plan.n_channels, x, r


TRAFFIC_SOURCE = ("committed rocprofv3 --pmc passes of the same command (profiles/pmc_traffic.json, builder-run); "
                  "not re-measured in this run")


def pmc_traffic(tag, kernels):
    """HBM bytes per launch of `kernels` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    produced by tools/gpu_profile.sh + tools/profile_report.py for the same bench command); None if absent.
    bytes = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE tallies 64 B per 128-B request) + WRITE_SIZE * 1024."""
    try:
        data = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[tag]["kernels"]
        have = [k for k in kernels if k in data]      # (an op is one of several kernels depending on the graph)
        if "agg_bwd_graph" in have:                   # (k-NN / SBM batches from round 5 on: the graph backward alone)
            have = ["agg_bwd_graph"]
        if not have:
            return None
        return sum(2 * 1024 * data[k]["FETCH_SIZE_KiB"] + 1024 * data[k]["WRITE_SIZE_KiB"] for k in have)
    except Exception:
        return None


TUNING_FILE = os.path.join(ROOT, "dgn_amd", "tunableop_gfx950.csv")


def configure_gemm_tuning(mode):
    """The skinny fp32 GEMMs around the sweep ([N,70]x[70,140], dW = dZ^T X with N ~ 3e5, per-tower batched
    GEMMs) hit poor default rocBLAS solutions (0.2-0.7 ms each where 0.1 ms is possible); TunableOp picks the
    best rocBLAS/hipBLASLt solution per shape.  'file' only replays the committed selection."""
    if mode == "off":
        return
    import torch.cuda.tunable as tn
    tn.enable(True)
    if mode == "tune":
        tn.set_filename(TUNING_FILE)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(150)
        tn.set_max_tuning_iterations(20)
    else:
        # replay only.  Whatever TunableOp writes at exit goes to a per-process scratch file, so that concurrent
        # ranks (torchrun) can never clobber the committed selection file.
        tn.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"dgn_tunableop_{os.getpid()}.csv"))
        tn.tuning_enable(False)
        if os.path.exists(TUNING_FILE):
            tn.read_file(TUNING_FILE)


def event_ms(fn, reps, dev, warm=1):
    """Average duration of fn() over `reps` back-to-back enqueues, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / reps


def event_stats(fn, dev, warm=20, groups=20, per_group=5):
    """SURVEY 8(d) timing protocol: `warm` warm-up calls, then groups x per_group = 100 timed calls, each group of back-to-back calls
    bracketed by its own HIP event pair on the launching stream; the per-call duration of a group = bracket / per_group (a bracket around
    ONE sub-50-us call would mostly measure the launch gap).  Returns ms: median (the figure the roofline uses), p10, p90, mean."""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    torch.cuda.synchronize(dev)
    for a, b in ev:
        a.record()
        for _ in range(per_group):
            fn()
        b.record()
    torch.cuda.synchronize(dev)
    t = sorted(a.elapsed_time(b) / per_group for a, b in ev)
    return dict(median=t[groups // 2], p10=t[groups // 10], p90=t[(groups * 9) // 10], mean=sum(t) / groups, warmup=warm,
                timed_calls=groups * per_group)


def _blk_name(kernel):
    """`blk_forward`, `blk_tail_fwd`, ... out of torch.profiler's kernel name."""
    import re
    m = re.search(r"blk_\w+", kernel)
    return m.group(0) if m else kernel[:40]


def step_min_bytes(N, E, F_in, F_out, Ku, n_params):
    """Mandatory HBM traffic of ONE layer step (forward + backward), whatever the kernels: inputs and outputs of both directions (h and
    g_out read, out and g_h written; h read again by the backward), the graph (row pointers and sources, both directions), the eig
    columns used (both directions), one saved activation of the output's size (what BatchNorm's adjoint needs: written once, read once),
    parameters read in both directions and their gradients written.  The denominator of ``roofline.step``."""
    act = 4 * N * (F_in + F_out + F_out + F_in + F_in) + 2 * 4 * N * F_out
    graph = 2 * (4 * E + 4 * (N + 1)) + 2 * 4 * N * Ku
    return act + graph + 3 * 4 * n_params


def step_kernel_table(step, dev, steps=5):
    """Per-kernel (calls per step, microseconds per step) of the timed step, from torch.profiler's device activity -- the table the
    committed rocprofv3 summary (profiles/) must agree with.  None where the profiler gives no device records."""
    try:
        from torch.profiler import ProfilerActivity, profile
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
        rows = {}
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and str(ev.device_type).endswith("CUDA"):
                d = rows.setdefault(ev.name, [0, 0.0])
                d[0] += 1
                d[1] += float(getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0) or 0.0)
        table = [dict(kernel=k[:96], calls_per_step=c / steps, us_per_step=t / steps) for k, (c, t) in rows.items() if t > 0]
        table.sort(key=lambda r: -r["us_per_step"])
        return table or None
    except Exception as exc:      # (the table is evidence, never a reason to lose the line)
        return [dict(error=f"{type(exc).__name__}: {exc}"[:160])]


def build_batch(wl, seed, dev, shard=None):
    """``shard = (rank, world)``: keep this rank's edge-balanced share of the batch's graphs (strong scaling)."""
    kind, kw = wl["gen"]
    if kind == "molecules":
        b = synth.molecule_batch(seed=seed, **kw)
    elif kind == "knn":
        b = synth.knn_batch(seed=seed, **kw)
    elif kind == "sbm":
        b = synth.sbm_batch(seed=seed, **kw)
    else:
        raise ValueError(kind)
    if shard is not None:
        rank, world = shard
        mine = ddist.shard_by_edges(synth.edges_per_graph(b).tolist(), world)[rank]
        b = synth.subset_batch(b, mine)
    graph = dgn_amd.DGNGraph(b["src"].to(dev), b["dst"].to(dev), b["num_nodes"], eig=b["eig"].to(dev))
    return b, graph


def cpu_baseline(wl, batch, sample_graphs, reps=5):
    """Oracle (reference-structured CPU restatement) layer fwd+bwd on the first `sample_graphs` graphs."""
    from oracle import dgn_oracle as orc
    sizes = batch["sizes"][:sample_graphs]
    n = int(sizes.sum())
    keep = batch["dst"] < n           # graphs are laid out consecutively; edges never cross graphs
    src, dst = batch["src"][keep], batch["dst"][keep]
    F_ = wl["hidden"]
    gen = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(1.0)},
                             wl["type_net"], True, towers=wl["towers"], edge_features=False, edge_dim=0).model
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in layer.state_dict().items()}
    cfg = dict(aggregators=wl["aggregators"], scalers=wl["scalers"], avg_log=torch.tensor(1.0), graph_norm=wl.get("graph_norm", True),
               batch_norm=True, residual=True, towers=wl["towers"], divide_input=True, edge_features=False)
    h = torch.randn(n, F_, generator=gen)
    eig, snorm = batch["eig"][:n], batch["snorm_n"][:n]
    ct = torch.randn(n, F_, generator=gen)

    def step():
        hh = h.clone().requires_grad_(True)
        y, _ = orc.layer_forward(wl["type_net"], sd, cfg, src, dst, n, eig, hh, None, snorm, training=True)
        y.backward(ct)

    def timed(threads, n):
        torch.set_num_threads(threads)
        step()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        return (time.perf_counter() - t0) / n

    # the CPU path with all threads torch uses by default AND with one thread; `value` is the faster
    # of the two (the reference's per-degree-bucket torch ops are small, so more threads do not always help)
    all_threads = torch.get_num_threads()
    runs = {all_threads: timed(all_threads, reps)}
    if all_threads > 1:
        runs[1] = timed(1, max(2, reps // 2))
        torch.set_num_threads(all_threads)
    cores, dt = min(runs.items(), key=lambda kv: kv[1])
    return dict(value=src.numel() / dt, unit="edges/s", cores=cores, kind="port",
                sample=f"first {sample_graphs} graphs of the batch ({n} nodes, {src.numel()} edges), layer fwd+bwd, "
                       f"{dt * 1e3:.1f} ms per pass with {cores} thread(s), torch {torch.__version__} CPU",
                edges_per_s_by_threads={str(k): src.numel() / v for k, v in runs.items()},
                host_cpus=os.cpu_count(), cpu_model=cpu_model())


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def hbm_triad_GBps(dev, n_bytes=1 << 30, reps=10):
    """Measured device-memory rate of c = a + s*b over three `n_bytes` fp32 arrays (the box's own
    stream-triad figure as a second denominator beside the 8 TB/s specification)."""
    n = n_bytes // 4
    a, b, c = (torch.empty(n, device=dev).normal_() for _ in range(3))
    ms = event_ms(lambda: torch.add(a, b, alpha=0.5, out=c), reps, dev)
    return 3 * n_bytes / ms / 1e6


def run_layer_workload(args, wl, rank, world, dev, steps=None, warmup=None, tag=None):
    """One layer workload: timed steps (contract timing) + the roofline of its aggregation kernels on rank 0."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    tag = tag or args.workload
    # weak scaling: every rank draws its OWN batch (generator seed 41 + rank; 41 = BASELINE config seed), so shapes differ
    # slightly between ranks as they do in a real data-parallel epoch; strong scaling: ONE global batch (seed 41 on every
    # rank), every rank keeps its edge-balanced share of the graphs.  Node features / cotangents are rank-specific.
    strong = world > 1 and args.scaling == "strong"
    batch, graph = build_batch(wl, 41 if strong else 41 + rank, dev, shard=(rank, world) if strong else None)
    F_ = wl["hidden"]
    N, E = graph.num_nodes, graph.num_edges
    torch.manual_seed(0)
    avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
    edge_dim = wl.get("edge_dim", 0)
    layer = dgn_amd.DGNLayer(F_, F_, wl.get("dropout", 0.0), wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"],
                             {"log": torch.tensor(avg_log)}, wl["type_net"], True, towers=wl["towers"], edge_features=edge_dim > 0,
                             edge_dim=edge_dim).model.to(dev)
    layer.train()
    gen = torch.Generator(device=dev).manual_seed(rank)
    h = torch.randn(N, F_, device=dev, generator=gen).requires_grad_(True)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = batch["snorm_n"].to(dev)
    ef = torch.randn(E, edge_dim, device=dev, generator=gen).requires_grad_(True) if edge_dim else None
    n_types = wl.get("edge_types", 0)
    ef_leaf = ef
    if n_types:     # the same features as (embedding table, bond type per edge)
        ef_leaf = torch.randn(n_types, edge_dim, device=dev, generator=gen).requires_grad_(True)
        ef = dgn_amd.EdgeTypeFeatures(ef_leaf, torch.randint(0, n_types, (E,), device=dev, generator=gen))
    reducer = ddist.FlatGradAllReduce(layer.parameters()) if torch.distributed.is_initialized() else None

    params = list(layer.parameters())      # (what an optimizer holds; walking the module tree costs 0.1 ms per step)

    def step():
        graph._wcache.clear()              # per-edge weights are recomputed every step (eig flips per batch)
        h.grad = None
        if ef_leaf is not None:
            ef_leaf.grad = None
        for p in params:
            p.grad = None
        y = layer(graph, h, ef, snorm)
        y.backward(ct)
        if reducer is not None:
            reducer()

    if args.hipgraph:
        # Launch-bound batches (the reference's batch of 128 molecules: ~80 launches around 0.4 ms of kernels): the whole
        # step -- edge weights, forward, backward -- is captured ONCE into a HIP graph and replayed.  Only valid for
        # a fixed batch shape (every kernel argument is frozen), so it is an option, not the headline mode.
        if reducer is not None and world > 1:
            raise SystemExit("--hipgraph is a single-GPU mode (the gradient all-reduce is not captured)")
        reducer = None      # (a forced single-rank process group: nothing to reduce)
        def bare_step():                   # (gradients stay None: the captured backward writes fresh ones per replay)
            graph._wcache.clear()
            layer(graph, h, ef, snorm).backward(ct)

        def reset():
            h.grad = None
            for p in params:
                p.grad = None

        def warm_step():
            reset()
            bare_step()

        from dgn_amd.hipgraph import capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                warm_step()                 # warm-up off the capture: allocator pools, csc view, cached tables
        torch.cuda.current_stream(dev).wait_stream(side)
        reset()
        hip_graph = capture(bare_step, warmup=0)
        eager_step, step = step, hip_graph.replay

    for _ in range(warmup):
        step()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    ms = ddist.barrier_max_ms(ms, dev)
    e_total = torch.tensor([float(E)], dtype=torch.float64, device=dev)
    if torch.distributed.is_initialized():
        torch.distributed.all_reduce(e_total)
    total_edges = float(e_total.item())

    result = dict(ms_per_step=ms, value=total_edges / (ms * 1e-3), edges_per_rank=E, nodes_per_rank=N,
                  scaling="strong" if strong else "weak")
    blk_rows = []
    if rank == 0:
        # the step against the layer's mandatory traffic (roofline.step), and where its time goes kernel by kernel (full record only)
        plan0 = layer._kplan_x if (wl["type_net"] != "simple" and hasattr(layer, "_kplan_x")) else layer._kplan
        n_par = sum(p.numel() for p in params)
        bmin = step_min_bytes(N, E, F_, F_, plan_model(plan0)[2], n_par)
        result["step"] = dict(bytes_min=bmin, frac=bmin / (ms * 1e-3) / HBM_PEAK, GBps=bmin / ms / 1e6,
                              model="h, g_out read; out, g_h written; h re-read by the backward; one saved activation; graph and eig columns "
                                    "both ways; parameters read twice, their gradients written")
        table = step_kernel_table(eager_step if args.hipgraph else step, dev)
        result["step_kernels"] = table
        if table and "us_per_step" in table[0]:
            result["step"]["kernel_us_sum"] = sum(r["us_per_step"] for r in table)
        blk_rows = [r for r in (table or []) if "us_per_step" in r and "blk_" in r["kernel"]]
    if torch.distributed.is_initialized() and reducer is not None:
        # per rank: its shard's size and the gradient all-reduce timed on its own (HIP events around 20 back-to-back calls)
        ms_ar = event_ms(reducer, 20, dev, warm=3)
        stats = ddist.gather_rank_stats([rank, N, E, ms_ar], dev)
        result["ranks"] = [dict(rank=int(r), nodes=int(n), edges=int(e), allreduce_ms=a) for r, n, e, a in stats]
        result["allreduce"] = dict(params=int(reducer.flat.numel()), bytes=int(reducer.flat.numel()) * 4,
                                   ms_max=max(a for _, _, _, a in stats), note="one flat fp32 buffer, sum + scale, inside the timed step")
    if rank != 0:
        return result

    if blk_rows:
        # The step ran on the GRAPH-BLOCK route (csrc/dgn_blk_layer.hip: batches up to ops.BLOCK_LAYER_MAX_NODES nodes): the whole layer is
        # five launches out of LDS and the streaming sweep kernels are never launched -- their roofline would describe kernels this
        # step does not run.  Reported instead: the block kernels' own durations (torch.profiler device records
        # of the timed step) and the layer's MANDATORY bytes against their sum -- a launch- / latency-bound regime, marked as such.
        us = sum(r["us_per_step"] for r in blk_rows)
        bmin = result["step"]["bytes_min"]
        result["roofline"] = dict(bound="hbm", regime="launch/latency-bound: graph-block route, whole layer out of LDS",
                                  kernel="blk_forward + blk_tail_fwd + blk_tail_bwd + blk_backward + blk_reduce (csrc/dgn_blk_layer.hip)",
                                  achieved=bmin / (us * 1e-6) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=bmin / (us * 1e-6) / HBM_PEAK,
                                  traffic=None, traffic_source=None,
                                  kernels={_blk_name(r["kernel"]): dict(ms=r["us_per_step"] * 1e-3, launches_per_step=r["calls_per_step"])
                                           for r in blk_rows},
                                  model=dict(N=N, E=E, F=F_, bytes="the layer step's mandatory bytes (roofline.step.bytes_min)", launches_per_step=sum(r["calls_per_step"] for r in blk_rows),
                                             block_kernel_us=us),
                                  step=result.get("step"))
        return result, batch

    # ---- roofline of the aggregation kernels (same shapes the layer launches) ----
    # the plan the layer actually launches: the degree scalers are folded into posttrans, so the sweep
    # runs with S = 1 (dgn_amd/dgn_layer.py: _fold_scalers) and its algorithmic bytes are counted as such
    plan = layer._kplan_x if (wl["type_net"] != "simple" and hasattr(layer, "_kplan_x")) else layer._kplan     # (+ h_in pass-through block)
    T = wl["towers"] if wl["type_net"] == "towers" else 1
    A, S, Ku, x, r = plan_model(plan)
    w = graph.edge_weights(plan)
    hd = h.detach()
    Fk = F_          # width the kernels are launched with
    from dgn_amd import ops as _ops
    f_valid = 0
    if F_ % 2 and wl["type_net"] in ("simple", "complex"):
        # the simple and complex layers run an odd hidden size at F + 1: the complex layer through one zero column in P | Q, the simple
        # layer -- where the list has `Cfg::ODD` kernels (c1, c3) -- on the UN-PADDED rows of h (DgnMsg.f_valid), both directions.  The
        # roofline legs launch exactly what the layer launches (they timed the padded, even-pitch instance before).
        Fk = F_ + 1
        if wl["type_net"] == "simple" and _ops.odd_direct_supported(graph, plan):
            f_valid = F_
        else:
            hd = torch.nn.functional.pad(hd, (0, 1))
    if wl["type_net"] == "simple":
        xs, xd = hd, None
    else:
        pq = torch.randn(N, 2 * Fk, device=dev, generator=gen)
        xs, xd = pq[:, :Fk], pq[:, Fk:]
    if T > 1:    # the towers layer runs the sweep tower-major ([T, N, A*F/T], dgn_layer.py: _fused_towers): time that layout
        out = torch.empty(T, N, plan.out_width(Fk) // T, device=dev)
        g_out = torch.randn(T, N, plan.out_width(Fk) // T, device=dev, generator=gen)
    else:
        out = torch.empty(N, plan.out_width(Fk), device=dev)
        g_out = torch.randn(N, plan.out_width(Fk), device=dev, generator=gen)
    g_src, g_dst, g_in = torch.zeros(N, Fk, device=dev), (torch.zeros(N, Fk, device=dev) if xd is not None else None), torch.zeros(N, Fk, device=dev)
    # with edge features the message has a third, per-edge term R = ef W_e^T [E, F] in slot order (materialised by a streaming
    # Linear): the sweep reads it (+4F per edge, forward and -- with max/min/std -- backward) and the backward writes d R
    # (EdgeTypeFeatures: the term is a [K, F] table + 4 bytes of type per edge; d table = the staged rows summed by type)
    me = torch.randn(n_types if n_types else E, Fk, device=dev, generator=gen) if edge_dim else None
    g_me = torch.empty_like(me) if edge_dim else None
    et = graph.to_slot_order(ef.types).to(torch.int32).contiguous() if n_types else None
    # the forward / backward pair as the layer runs it in training: where the launch has an aux table (dgn_agg_forward_aux: slots of the
    # first max / min and dx signs, one byte per row and feature) the forward writes it and the backward works from it
    n_aux = _ops.agg_aux_bytes(graph, plan, T, Fk, xs, xd, me, hd, et, f_valid=f_valid) if (_ops.AGG_AUX and wl["type_net"] in AUX_LAYERS) else 0
    aux = torch.empty(n_aux, dtype=torch.uint8, device=dev) if n_aux else None
    fwd_call = lambda: launch_forward(graph, plan, T, avg_log, w, xs, xd, me, hd, out, edge_type=et, aux=aux, f_valid=f_valid)
    bwd_call = lambda: launch_backward(graph, plan, T, avg_log, w, xs, xd, me, hd, g_out, g_src, g_dst, g_me, g_in, accumulate=False,
                                       edge_type=et, aux=aux, f_valid=f_valid)
    st_f, st_b = event_stats(fwd_call, dev), event_stats(bwd_call, dev)
    st_w = event_stats(lambda: dgn_amd.compute_edge_weights(graph, plan.channels, eig=graph.ndata["eig"]), dev)
    ms_f, ms_b, ms_w = st_f["median"], st_b["median"], st_w["median"]
    bf, bb = algorithmic_bytes(N, E, F_, A, S, Ku, x, r)
    if n_types:
        bf += 4 * E
        bb += 4 * E + 4 * E + 4 * F_ * E       # types (sweep, reduction), the reduction's read of the staged rows
    elif edge_dim:
        bf += 4 * F_ * E
        bb += 4 * F_ * E * (1 + r)
    bw = E * (4 + 8 * Ku + 4 * Ku) + N * 4        # edge weights: src id, both eig endpoints per channel, weight out; row pointer
    kernels = {"agg_fwd_rows": dict(ms=ms_f, bytes=bf, GBps=bf / ms_f / 1e6, frac=bf / (ms_f * 1e-3) / HBM_PEAK,
                                    timing=st_f),
               "agg_bwd_rows": dict(ms=ms_b, bytes=bb, GBps=bb / ms_b / 1e6, frac=bb / (ms_b * 1e-3) / HBM_PEAK,
                                    timing=st_b),
               "ew_rows": dict(ms=ms_w, bytes=bw, GBps=bw / ms_w / 1e6, frac=bw / (ms_w * 1e-3) / HBM_PEAK, timing=st_w)}
    # simple / complex layers with several scalers: the three posttrans products on the degree-class kernels (MFMA-bound: exact fp32
    # v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense peak), priced on the USEFUL flops 2 N K f_out each
    kernels.update(dc_posttrans_legs(graph, wl, N, F_, plan.out_width(Fk), dev, gen))
    dom = "agg_bwd_rows" if ms_b >= ms_f else "agg_fwd_rows"
    # the timed op is one dgn_agg_forward / dgn_agg_backward call: forward = agg_fwd_short (4 rows per wave, short
    # rows) or agg_fwd_rows; backward = agg_bwd_short (four short rows per wave) or agg_bwd_rows, + seg_sum_rows (second phase of the
    # atomic-free scatter)
    # -- or, on batches of small graphs from 131 072 nodes on, agg_bwd_block ALONE (one wave per run of whole graphs, d x_src in LDS)
    # -- or, on k-NN / SBM batches (more than 3 edges per node, graphs of up to 512 nodes), agg_bwd_graph ALONE (a workgroup per graph)
    launches = {"agg_fwd_rows": ["agg_fwd_rows", "agg_fwd_short"], "agg_bwd_rows": ["agg_bwd_rows", "agg_bwd_short", "seg_sum_rows", "agg_bwd_block", "agg_bwd_graph"]}[dom]
    label = {"agg_fwd_rows": "dgn_agg_forward (agg_fwd_short | agg_fwd_rows)",
             "agg_bwd_rows": "dgn_agg_backward (agg_bwd_block | agg_bwd_graph | agg_bwd_short + seg_sum_rows | agg_bwd_rows + seg_sum_rows)"}[dom]
    triad = hbm_triad_GBps(dev)
    # The launched list carries the h_in pass-through block of the complex / towers layers as one more "aggregator"
    # (A = survey's A + 1: the sweep really writes that block).  The same launch priced with SURVEY 8(d)'s own A:
    A_survey = A - int(any(op == 10 for l in plan.launches for op in l.ops))
    bf_s, bb_s = algorithmic_bytes(N, E, F_, A_survey, S, Ku, x, r)
    dom_ms = kernels[dom]["ms"]
    frac_survey = (bb_s if dom == "agg_bwd_rows" else bf_s) / (dom_ms * 1e-3) / HBM_PEAK
    traffic = pmc_traffic(tag, launches)
    result["roofline"] = dict(bound="hbm", kernel=label, achieved=kernels[dom]["GBps"], peak=HBM_PEAK / 1e9, unit="GB/s",
                              frac=kernels[dom]["frac"], traffic=traffic, traffic_source=TRAFFIC_SOURCE if traffic else None,
                              kernels=kernels, triad_GBps=triad, frac_of_triad=kernels[dom]["GBps"] / triad,
                              model=dict(N=N, E=E, F=F_, A=A, S=S, Ku=Ku, x=x, r=r, aux_bytes=n_aux, f_valid=f_valid),
                              frac_with_survey_A=dict(A=A_survey, frac=frac_survey,
                                                      note="same launch priced without the h_in pass-through block"),
                              step=result.get("step"))
    return result, batch


MFMA_F32_PEAK = 157.3e12  # FLOP/s, dense fp32 MFMA (/opt/skills/guides/MI355X_MICROARCH.md)


def dc_posttrans_legs(graph, wl, N, fo, K, dev, gen):
    """dgn_dc_gemm (forward, input gradient) and dgn_dc_wgrad on the layer's shapes, through the C ABI: {} where the layer does not take
    that route (towers, a single scaler, an in-degree >= 32)."""
    import ctypes as C
    from dgn_amd import _lib, ops as _ops
    S = len(wl["scalers"].split())
    dc = graph.degree_classes() if (wl["type_net"] in ("simple", "complex") and S > 1 and _ops.DC_POSTTRANS and N >= _ops.DC_MIN_NODES) else None
    lib = _lib.load()
    if dc is None or not (lib.dgn_dc_supported(K, fo) and lib.dgn_dc_supported(fo, K) and lib.dgn_dc_wgrad_supported(K, fo)):
        return {}
    scale = torch.rand(32, S, device=dev, generator=gen) + 0.5
    d = _lib.DgnDegreeClasses(n_units=dc["n_units"], vperm=dc["vperm"].data_ptr(), unit_class=dc["unit_class"].data_ptr(),
                              present=dc["present"].data_ptr(), scale=scale.data_ptr())
    agg, g = torch.randn(N, K, device=dev, generator=gen), torch.randn(N, fo, device=dev, generator=gen)
    wc = torch.randn(32, fo, K, device=dev, generator=gen) / K ** 0.5
    wct = wc.transpose(1, 2).contiguous()
    y, g_agg, g_wf = torch.empty(N, fo, device=dev), torch.empty(N, K, device=dev), torch.empty(S * fo, K, device=dev)
    nbytes = lib.dgn_dc_wgrad_workspace_bytes(dc["n_units"], K, fo)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    calls = {
        "dc_gemm_forward": lambda: _lib.check(lib.dgn_dc_gemm(C.byref(d), K, fo, 1, agg.data_ptr(), K, 0, wc.data_ptr(), K, fo * K, 0, None, None,
                                                              y.data_ptr(), fo, 0, 0, st), "dgn_dc_gemm"),
        "dc_gemm_input_grad": lambda: _lib.check(lib.dgn_dc_gemm(C.byref(d), fo, K, 1, g.data_ptr(), fo, 0, wct.data_ptr(), fo, fo * K, 0, None, None,
                                                                 g_agg.data_ptr(), K, 0, 0, st), "dgn_dc_gemm"),
        "dc_wgrad": lambda: _lib.check(lib.dgn_dc_wgrad(C.byref(d), S, K, fo, g.data_ptr(), fo, agg.data_ptr(), K, g_wf.data_ptr(), K, None,
                                                        ws.data_ptr(), nbytes, st), "dgn_dc_wgrad"),
    }
    flops = 2.0 * N * K * fo
    out = {}
    for name, fn in calls.items():
        stt = event_stats(fn, dev)
        out[name] = dict(ms=stt["median"], flops=flops, TFLOPs=flops / stt["median"] / 1e9, bound="mfma", peak_TFLOPs=MFMA_F32_PEAK / 1e12,
                         frac=flops / (stt["median"] * 1e-3) / MFMA_F32_PEAK, bytes=4 * N * (K + fo), timing=stt)
    return out


def run_c5(args, wl, rank, world, dev, steps=None, warmup=None, tag=None):
    """Single forward pass of the fused aggregation on the power-law graph (HBM-roofline run)."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    kw = dict(wl["gen"][1])
    if args.scale != 1.0:
        kw["num_nodes"] = int(kw["num_nodes"] * args.scale)
        kw["num_edges"] = int(kw["num_edges"] * args.scale)
    partition = world > 1 and args.c5_mode == "partition"
    # partition: ALL ranks build the same graph and own a destination range with a balanced edge count (strong
    # scaling of one graph; the node features stay whole on every rank, the sweep needs no exchange).
    # replicas: every rank its own graph (weak scaling).
    indptr, src, eig = synth.powerlaw_csr(device=dev, seed=0 if partition else rank, **kw)
    r0, r1 = 0, indptr.numel() - 1
    if partition:
        r0, r1 = ddist.row_ranges_by_edges(indptr, world)[rank]
        graph = ddist.shard_rows(indptr, src, r0, r1)
        graph.ndata["eig"] = eig
    else:
        graph = dgn_amd.DGNGraph.from_csr(indptr, src, eig=eig)
    N_all = indptr.numel() - 1
    eig_info = None
    if args.c5_eig == "lobpcg" and not partition:
        # real Laplacian eigenvectors instead of random columns: k lowest eigenpairs of D - (A + A^T)/2 by
        # LOBPCG on the sweep's own products; column 0 of eig is the trivial one, dir1..3 read columns 1..3
        from dgn_amd.eig import lobpcg_eigvecs
        torch.cuda.synchronize(dev)
        t_e = time.perf_counter()
        eig, lam, its, res = lobpcg_eigvecs(graph, eig.shape[1], iters=args.c5_eig_iters, tol=1e-3,
                                            generator=torch.Generator(device=dev).manual_seed(0))
        torch.cuda.synchronize(dev)
        eig_info = dict(method="LOBPCG on the sweep's sum-aggregator products", seconds=time.perf_counter() - t_e, iterations=its,
                        eigenvalues=[float(v) for v in lam], residual_norms=[float(v) for v in res])
        graph.ndata["eig"] = eig
    del indptr
    N, E, F_ = graph.num_nodes, graph.num_edges, wl["hidden"]
    plan = dgn_amd.make_plan(wl["aggregators"].split(), wl["scalers"].split())
    avg_log = float(graph.log_deg.mean().item())
    gen = torch.Generator(device=dev).manual_seed(0 if partition else rank)
    X_all = torch.randn(N_all, F_, device=dev, generator=gen)
    X = X_all[r0:r1]                                   # x_in: the shard's own rows
    out = torch.empty(N, plan.out_width(F_), device=dev)
    w = graph.edge_weights(plan)

    def step():
        launch_forward(graph, plan, 1, avg_log, w, X_all, None, None, X, out)

    for _ in range(warmup):
        step()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    ms_local = (time.perf_counter() - t0) * 1e3 / steps           # this rank's own sweep time (before it waits for the others)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    ms = ddist.barrier_max_ms((time.perf_counter() - t0) * 1e3 / steps, dev)
    e_total = torch.tensor([float(E)], dtype=torch.float64, device=dev)
    if torch.distributed.is_initialized():
        torch.distributed.all_reduce(e_total)
    ranks_info = None
    if torch.distributed.is_initialized():
        ms_ag = None
        if partition:
            # the exchange between two sharded layers: the F-wide output rows of every shard all-gathered (dist.all_gather_rows)
            rows = torch.randn(N, F_, device=dev)
            rr = [(int(a), int(b)) for a, b in ddist.gather_rank_stats([r0, r1], dev)]
            ms_ag = event_ms(lambda: ddist.all_gather_rows(rows, rr), 5, dev, warm=2)
        stats = ddist.gather_rank_stats([rank, N, E, ms_local, ms_ag if ms_ag is not None else -1.0], dev)
        ranks_info = [dict(rank=int(r), rows=int(n), edges=int(e), sweep_ms=t, all_gather_ms=(a if a >= 0 else None)) for r, n, e, t, a in stats]
    result = dict(ms_per_step=ms, value=float(e_total.item()) / (ms * 1e-3), edges_per_rank=E, nodes_per_rank=N, ranks=ranks_info,
                  eig=eig_info or "random columns", scaling="strong" if partition else "weak",
                  parallelism=(f"1 graph, {world} destination-range shards (balanced edges), features replicated, no exchange "
                               f"inside the sweep") if partition else (f"{world} independent replicas" if world > 1 else "single GPU"))
    if rank == 0:
        A, S, Ku, x, r = plan_model(plan)
        ms_f = event_ms(step, max(3, steps), dev)
        ms_w = event_ms(lambda: dgn_amd.compute_edge_weights(graph, plan.channels, eig=eig), 3, dev)
        bf, _ = algorithmic_bytes(N, E, F_, A, S, Ku, x, r)
        result["roofline"] = dict(bound="hbm", kernel="agg_fwd_rows+hub", achieved=bf / ms_f / 1e6, peak=HBM_PEAK / 1e9,
                                  unit="GB/s", frac=bf / (ms_f * 1e-3) / HBM_PEAK, triad_GBps=(triad := hbm_triad_GBps(dev)),
                                  frac_of_triad=bf / ms_f / 1e6 / triad,
                                  traffic=pmc_traffic("c5", ["agg_fwd_rows", "agg_hub_slices", "agg_fwd_hub_combine"])
                                  if args.scale == 1.0 and not args.aggregators and not args.scalers else None,
                                  traffic_source=TRAFFIC_SOURCE,
                                  kernels={"agg_fwd(all launches)": dict(ms=ms_f, bytes=bf),
                                           "edge_weights": dict(ms=ms_w, bytes=(bw := E * (4 + 12 * Ku) + N * 4), GBps=bw / ms_w / 1e6,
                                                                frac=bw / (ms_w * 1e-3) / HBM_PEAK)},
                                  model=dict(N=N, E=E, F=F_, A=A, S=S, Ku=Ku, x=x, r=r, n_hub=graph.n_hub,
                                             n_slices=graph.n_chunks, max_degree=int(graph.in_degree.max().item())))
    return result, None


def run_c5_layer(args, wl, rank, world, dev, steps=None, warmup=None, tag=None):
    """The C5 graph through a WHOLE simple layer forward (nets/dgn_layer.py:178-202 without gradients): sweep with the scalers folded
    behind posttrans ([N, 8 x 128] aggregates), the folded posttrans product (1024 -> 3 x 128), scale-combine + BatchNorm (running
    statistics) + ReLU + residual.  Reported with the bytes a FUSED layer would have to move (no aggregate block written or read:
    E (4 + 4F + 4Ku) + N (4 + 4Ku + 4F + 4F)) and with the product's MFMA time, which is what bounds it."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    kw = dict(wl["gen"][1])
    kw["num_nodes"], kw["num_edges"] = int(kw["num_nodes"] * args.scale), int(kw["num_edges"] * args.scale)
    indptr, src, eig = synth.powerlaw_csr(device=dev, seed=0, **kw)
    graph = dgn_amd.DGNGraph.from_csr(indptr, src, eig=eig)
    del indptr
    N, E, F_ = graph.num_nodes, graph.num_edges, wl["hidden"]
    avg_log = float(graph.log_deg.mean().item())
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, False, True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)}, "simple", True,
                             towers=1, edge_features=False, edge_dim=0).model.to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(0)
    h = torch.randn(N, F_, device=dev, generator=gen)
    plan = layer._kplan
    lin = layer.posttrans.fully_connected[0].linear
    A_ = len(layer.aggregators)
    S_ = layer.plan.n_scalers
    from dgn_amd.dgn_layer import node_linear
    w_f = lin.weight.detach().reshape(F_, S_, A_ * F_).permute(1, 0, 2).reshape(S_ * F_, A_ * F_).contiguous()

    def step():
        with torch.no_grad():
            return layer(graph, h, None, None)

    parts = {}
    with torch.no_grad():
        for _ in range(warmup):
            step()
        ms = event_ms(step, steps, dev, warm=0)
        agg = layer.aggregate(graph, h, plan, graph.ndata["eig"])
        parts["sweep (scalers folded: [N, 1024] aggregates written)"] = event_ms(lambda: layer.aggregate(graph, h, plan, graph.ndata["eig"]), steps, dev)
        parts["posttrans product 1024 -> 384 (exact-fp32 MFMA)"] = event_ms(lambda: node_linear(agg, w_f), steps, dev)
        from dgn_amd import ops as _ops
        from dgn_amd.dgn_layer import _scale_table
        split = None
        if _ops.dc_posttrans_split_supported(graph, agg, F_, S_):
            sc = _scale_table(graph, layer.plan.applied_scalers, layer._avg_log)
            n_hub = int(graph.degree_classes_split()["hub_rows"].numel())
            split_ms = event_ms(lambda: _ops.dc_posttrans_split(graph, agg, lin.weight, lin.bias, sc, None, A_, F_), steps, dev)
            split = dict(ms=split_ms, hub_rows=n_hub, flops=2.0 * (N - n_hub) * (A_ * F_) * F_ + 2.0 * n_hub * (A_ * F_) * (S_ * F_))
        del agg
    A, S, Ku, x, r = plan_model(dgn_amd.make_plan(wl["aggregators"].split(), wl["scalers"].split()))
    fused_bytes = E * (4 + 4 * F_ + 4 * Ku) + N * (4 + 4 * Ku + 4 * F_ + 4 * F_)
    unfused_bytes = E * (4 + 4 * F_ + 4 * Ku) + N * (4 + 4 * Ku + 4 * F_) + 2 * N * 4 * A * F_ + 2 * N * 4 * S * F_ + N * 4 * 2 * F_
    flops = 2.0 * N * (A * F_) * (S * F_)
    prod_key = "posttrans product 1024 -> 384 (exact-fp32 MFMA)"
    if split is not None:      # the route the layer takes: the MFMA fraction is priced on ITS flops and time
        parts["(not on the layer's route) folded product 1024 -> 384 on all rows"] = parts[prod_key]
        parts[prod_key], flops = split["ms"], split["flops"]
    result = dict(ms_per_step=ms, value=E / (ms * 1e-3), edges_per_rank=E, nodes_per_rank=N, scaling="weak", parallelism="single GPU",
                  roofline=dict(bound="mfma", kernel="posttrans product of the layer (dc_gemm + tile_gemm on the hub rows, v_mfma_f32_16x16x4_f32)",
                                achieved=flops / (parts["posttrans product 1024 -> 384 (exact-fp32 MFMA)"] * 1e-3) / 1e12, peak=MFMA_F32_PEAK / 1e12,
                                unit="TFLOP/s", frac=flops / (parts["posttrans product 1024 -> 384 (exact-fp32 MFMA)"] * 1e-3) / MFMA_F32_PEAK, traffic=None,
                                kernels={k: dict(ms=v) for k, v in parts.items()},
                                model=dict(N=N, E=E, F=F_, A=A, S=S, Ku=Ku, layer_flops=flops, fused_layer_bytes=fused_bytes,
                                           hub_rows=None if split is None else split["hub_rows"],
                                           posttrans_route="folded 1024 -> 384 product" if split is None else
                                           "one 1024 -> 128 product per in-degree class below 32, folded product on the gathered hub rows",
                                           unfused_layer_bytes=unfused_bytes, sweep_only_bytes_c5=algorithmic_bytes(N, E, F_, A, S, Ku, x, r)[0],
                                           hbm_frac_on_fused_bytes=fused_bytes / (ms * 1e-3) / HBM_PEAK,
                                           mfma_floor_ms=flops / MFMA_F32_PEAK * 1e3)))
    return result, None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks_if_needed(args):
    """``--gpus N`` without a launcher environment: re-execute under torch.distributed.run, one rank per GPU.  Never
    degrades silently: fewer visible devices than requested ranks is an error."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {n_dev} GPU(s) are visible; refusing to run "
                         f"fewer ranks under that label")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def compact(result):
    """A sub-result of the default run: timing, throughput and the roofline of its dominant aggregation call."""
    r = result.get("roofline") or {}
    out = dict(ms_per_step=result["ms_per_step"], value=result["value"], unit="edges/s", edges=result["edges_per_rank"],
               nodes=result["nodes_per_rank"])
    if "eig" in result:
        out["eig"] = result["eig"]
    if result.get("step_kernels"):
        out["step_kernels"] = result["step_kernels"]
    if r:
        out["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                 "frac_of_triad", "model", "frac_with_survey_A", "step") if k in r}
        out["roofline"]["kernels"] = {k: {kk: vv for kk, vv in v.items() if kk in ("ms", "bytes", "GBps", "frac", "timing", "TFLOPs", "bound", "flops")}
                                      for k, v in (r.get("kernels") or {}).items()}
    return out


def run_inference(args, dev, steps=20, warmup=5):
    """Forward-only (no-grad, eval-mode) pass of the headline layer: the path on which sweep + posttrans + scale-combine run as ONE
    kernel (layer_fwd_fused) and the [N, A*F] aggregate rows never reach memory; timed with that kernel and with the separate ones."""
    from dgn_amd import ops
    wl = dict(WORKLOADS["c2"])
    batch, graph = build_batch(wl, 41, dev)
    F_, N, E = wl["hidden"], graph.num_nodes, graph.num_edges
    torch.manual_seed(0)
    avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)}, "towers", True,
                             towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).eval()
    h = torch.randn(N, F_, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    snorm = batch["snorm_n"].to(dev)
    out = {}
    saved = ops.FUSED_FORWARD
    try:
        for tag, flag in (("fused_kernel", True), ("separate_kernels", False)):
            ops.FUSED_FORWARD = flag
            with torch.no_grad():
                def step():
                    graph._wcache.clear()
                    return layer(graph, h, None, snorm)
                for _ in range(warmup):
                    step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) * 1e3 / steps
            out[tag] = dict(ms_per_step=ms, value=E / (ms * 1e-3), unit="edges/s")
    finally:
        ops.FUSED_FORWARD = saved
    out["config"] = "c2 layer, eval mode, torch.no_grad(): edge weights + forward only"
    return out


def eval_forward_ms(wl, dev, iters=300):
    """The layer's evaluation-mode forward (eval() under no_grad: the reference's validation / test loops) on the workload's batch, eager."""
    batch, graph = build_batch(wl, 41, dev)
    F_, N = wl["hidden"], graph.num_nodes
    avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, wl.get("dropout", 0.0), wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)},
                             wl["type_net"], True, towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).eval()
    h = torch.randn(N, F_, device=dev)
    snorm = batch["snorm_n"].to(dev)

    def step():
        graph._wcache.clear()
        with torch.no_grad():
            return layer(graph, h, None, snorm)

    for _ in range(20):
        step()
    best = None
    for _ in range(3):      # (host-bound: the best of three windows -- a long-lived process showed 4x outliers on single windows)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters // 3):
            step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / (iters // 3)
        best = ms if best is None else min(best, ms)
    return best


def eager_stack_ms(wl, dev, layers=4, iters=200, warm=30):
    """EAGER training step of a STACK of `layers` identical layers (the reference's nets hold L = 4: configs/*.json "L": 4) on the
    workload's batch: forward through all, ONE backward.  Autograd's hand-over to its device thread (~65 us) is paid once per
    backward(), not once per layer, so a single-layer eager leg overstates the per-layer host cost.  Returns
    (ms per step, ms per step with autograd on the calling thread)."""
    batch, graph = build_batch(wl, 41, dev)
    F_, N = wl["hidden"], graph.num_nodes
    avg_log = float(torch.log(graph.in_degree.float() + 1).mean().item())
    torch.manual_seed(0)
    stack = [dgn_amd.DGNLayer(F_, F_, wl.get("dropout", 0.0), wl.get("graph_norm", True), True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(avg_log)},
                              wl["type_net"], True, towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).train() for _ in range(layers)]
    params = [p for l in stack for p in l.parameters()]
    gen = torch.Generator(device=dev).manual_seed(0)
    h = torch.randn(N, F_, device=dev, generator=gen).requires_grad_(True)
    ct = torch.randn(N, F_, device=dev, generator=gen)
    snorm = batch["snorm_n"].to(dev)

    def step():
        graph._wcache.clear()
        h.grad = None
        for p in params:
            p.grad = None
        y = h
        for l in stack:
            y = l(graph, y, None, snorm)
        y.backward(ct)

    def timed():
        for _ in range(warm):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3 / iters

    ms = timed()
    torch.autograd.set_multithreading_enabled(False)
    try:
        ms_st = timed()
    finally:
        torch.autograd.set_multithreading_enabled(True)
    return ms, ms_st


def _max_graph_edges(b):
    """Largest number of directed edges of one graph of a synthetic batch (host side: what a data loader knows about its data set)."""
    import numpy as np
    cuts = np.concatenate([[0], np.cumsum(np.asarray(b["sizes"], dtype=np.int64))])
    gid = np.searchsorted(cuts, b["dst"].numpy(), side="right") - 1
    return int(np.bincount(gid, minlength=len(cuts) - 1).max())


def run_bucketed(args, dev, steps=200, warmup=30, n_batches=8, workload="c2_b128"):
    """The reference's operating point -- batches of 128 molecules whose node / edge counts differ from step to step -- on ONE
    captured HIP graph: the batch lives in static buffers at a fixed capacity (hipgraph.PaddedBatch), BatchNorm reads the number of
    real rows from the device.  A timed step = graph preparation of the NEXT batch in place (dgn_graph_build + _csc, no host sync)
    + copies of its features / eig / graph norm / cotangent into the static buffers + one graph launch (edge weights, forward,
    backward).  The batches are on the device beforehand (as after a data loader's H2D copy)."""
    from dgn_amd.hipgraph import PaddedBatch, bucket_capacity, capture
    wl = dict(WORKLOADS[workload])
    F_ = wl["hidden"]
    raw = [synth.molecule_batch(seed=41 + i, **wl["gen"][1]) for i in range(n_batches)]
    n_cap, e_cap = bucket_capacity(max(int(b["num_nodes"]) for b in raw), max(b["src"].numel() for b in raw))
    gen = torch.Generator(device=dev).manual_seed(0)
    data = []
    for b in raw:
        N = int(b["num_nodes"])
        data.append(dict(src=b["src"].to(dev), dst=b["dst"].to(dev), N=N, eig=b["eig"].to(dev), snorm=b["snorm_n"].to(dev),
                         sizes=[int(x) for x in b["sizes"]], max_edges=_max_graph_edges(b),
                         h=torch.randn(N, F_, device=dev, generator=gen), ct=torch.randn(N, F_, device=dev, generator=gen)))
    torch.manual_seed(0)
    layer = dgn_amd.DGNLayer(F_, F_, 0.0, True, True, wl["aggregators"], wl["scalers"], {"log": torch.tensor(1.0)}, wl["type_net"], True,
                             towers=wl["towers"], edge_features=False, edge_dim=0).model.to(dev).train()
    pb = PaddedBatch(n_cap, e_cap, dev, eig_dim=raw[0]["eig"].shape[1])
    # the static block table (one graph per block, capacity = the data set's largest graph): the captured step runs the graph-block route
    pb.graph.set_block_capacity(max(len(d["sizes"]) for d in data), max(max(d["sizes"]) for d in data), max(d["max_edges"] for d in data))
    h_buf, sn_buf, ct_buf = pb.add_node_tensor("h", F_, requires_grad=True), pb.add_node_tensor("snorm", 1), pb.add_node_tensor("ct", F_)
    params = list(layer.parameters())

    def load(i):
        d = data[i % n_batches]
        pb.load(d["src"], d["dst"], d["N"], d["eig"], node=dict(h=d["h"], snorm=d["snorm"], ct=d["ct"]), graph_sizes=d["sizes"])

    def bare_step():
        pb.graph.invalidate_caches()              # edge weights and scaler tables are recomputed inside the captured region
        layer(pb.graph, h_buf, None, sn_buf).backward(ct_buf)

    def reset():
        h_buf.grad = None
        for p in params:
            p.grad = None

    load(0)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            reset()
            bare_step()
    torch.cuda.current_stream(dev).wait_stream(side)
    reset()
    graph = capture(bare_step, warmup=0)
    for s_ in range(warmup):
        load(s_)
        graph.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s_ in range(steps):
        load(s_)
        graph.replay()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / steps
    edges = sum(d["src"].numel() for d in data) / n_batches
    return dict(ms_per_step=ms, value=edges / (ms * 1e-3), unit="edges/s", steps=steps, warmup=warmup,
                capacity=dict(nodes=n_cap, edges=e_cap), batches=[dict(nodes=d["N"], edges=int(d["src"].numel())) for d in data],
                config=f"{workload}: {n_batches} different batches of 128 molecules cycled through ONE captured HIP graph (padded to the capacity); a step = "
                       "in-place graph preparation + input copies + graph launch (edge weights + layer forward + backward)")


def run_net(args, dev, n_graphs=128, steps=60, warmup=15, layers=4, edge_feat=False):
    """The graph-regression NET around the layer (dgn_amd.nets.DGNNet = nets/molecules_graph_regression/dgn_net.py: atom embedding, four
    towers layers, mean readout, MLPReadout, L1 loss; configs/molecules_graph_regression_DGN_ZINC.json: L = 4, batch 128, no bond
    features) at the reference's ZINC batch size.  One training step = graph preparation of the batch + edge weights + forward + loss +
    backward + Adam update; eager (no graph capture)."""
    from dgn_amd.nets import DGNNet
    raw = [synth.molecule_batch(n_graphs=n_graphs, seed=41 + i, extra_bonds=3.9, eig_dim=6) for i in range(4)]
    avg_log = float(torch.log(torch.bincount(raw[0]["dst"], minlength=int(raw[0]["num_nodes"])).float() + 1).mean())
    net = DGNNet(dict(num_atom_type=28, num_bond_type=4, hidden_dim=70, out_dim=70, in_feat_dropout=0.0, dropout=0.0, L=layers, type_net="towers",
                      pos_enc_dim=0, readout="mean", graph_norm=True, batch_norm=True, aggregators="mean max min dir1-av dir1-dx",
                      scalers="identity amplification attenuation", avg_d={"log": torch.tensor(avg_log)}, residual=True, edge_feat=edge_feat, edge_dim=10 if edge_feat else 0,
                      pretrans_layers=1, posttrans_layers=1, device=str(dev))).to(dev).train()
    try:
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)      # one multi-tensor kernel instead of a launch per parameter
    except Exception:
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(0)
    batches = []
    for b in raw:
        N, E = int(b["num_nodes"]), b["src"].numel()
        batches.append(dict(src=b["src"].to(dev), dst=b["dst"].to(dev), N=N, eig=b["eig"].to(dev), sizes=[int(x) for x in b["sizes"]], max_edges=_max_graph_edges(b),
                            atoms=torch.randint(0, 28, (N,), generator=gen).to(dev), bonds=torch.randint(0, 4, (E,), generator=gen).to(dev),
                            snorm=b["snorm_n"].to(dev), y=torch.randn(len(b["sizes"]), 1, generator=gen).to(dev)))

    def step(i):
        b = batches[i % len(batches)]
        g = dgn_amd.DGNGraph(b["src"], b["dst"], b["N"], eig=b["eig"])          # per-batch graph preparation is part of the step
        g.batch_num_nodes = b["sizes"]
        opt.zero_grad(set_to_none=True)
        loss = net.loss(net(g, b["atoms"], b["bonds"] if edge_feat else None, b["snorm"], None), b["y"])
        loss.backward()
        opt.step()

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / max(steps, 1) * 1e3
    edges = sum(b["src"].numel() for b in batches) / len(batches)
    captured = None
    if getattr(args, "net_capture", True) and not edge_feat:
        # the same training step as ONE captured HIP graph over capacity-padded static buffers (hipgraph.CapturedNetStep); a step =
        # load of the next batch (graph rebuilt in place, input copies) + one graph launch
        try:
            from dgn_amd.hipgraph import CapturedNetStep, bucket_capacity
            n_cap, e_cap = bucket_capacity(max(b["N"] for b in batches), max(b["src"].numel() for b in batches))
            cs = CapturedNetStep(net, n_cap, e_cap, g_cap=n_graphs + 1, eig_dim=batches[0]["eig"].shape[1], lr=1e-3,
                                 max_graph_nodes=max(max(b["sizes"]) for b in batches), max_graph_edges=max(b["max_edges"] for b in batches))
            load = lambda b: cs.load(b["src"], b["dst"], b["N"], b["eig"], b["atoms"], b["snorm"], b["sizes"], b["y"])
            load(batches[0])
            cs.capture(warmup=3)
            for i in range(10):
                load(batches[i % len(batches)])
                cs.step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                load(batches[i % len(batches)])
                cs.step()
            torch.cuda.synchronize(dev)
            ms_c = (time.perf_counter() - t0) / steps * 1e3
            t0 = time.perf_counter()
            for i in range(steps):
                cs.step()
            torch.cuda.synchronize(dev)
            ms_r = (time.perf_counter() - t0) / steps * 1e3
            captured = dict(ms_per_step=ms_c, value=layers * edges / (ms_c * 1e-3), replay_only_ms=ms_r, capacity=dict(nodes=n_cap, edges=e_cap),
                            config="the same step as one captured HIP graph (forward, masked L1 loss, backward, capturable Adam) over padded static "
                                   "buffers; a step = in-place graph preparation + input copies + one graph launch; replay_only_ms = the launch alone")
        except Exception as exc:
            captured = dict(error=f"{type(exc).__name__}: {exc}")
    return dict(captured=captured, ms_per_step=ms, value=layers * edges / (ms * 1e-3), unit="layer-edges/s", steps=steps, warmup=warmup, layers=layers, graphs=n_graphs,
                edge_feat=edge_feat,
                config="ZINC-like batch of 128 molecules through dgn_amd.nets.DGNNet (4 towers layers, mean readout, MLP, L1 loss, fused Adam), "
                       "eager, graph preparation included")


COMMITTED_LINE = os.path.join(ROOT, "profiles", "r06_bench_all_extras.json")


def regression_warnings(results, headline=None):
    """Measurement hygiene: every roofline fraction of this run is compared with the committed line of the same
    command (profiles/r06_bench_all_extras.json); a leg that fell below half of its committed value gets a `warning` field instead of
    passing silently (a box with a disturbed clock, or a regression)."""
    try:
        ref = json.load(open(COMMITTED_LINE))
    except Exception:
        return
    def check(name, mine, theirs):
        k_m, k_t = (mine or {}).get("kernels") or {}, (theirs or {}).get("kernels") or {}
        bad = [f"{k}: frac {k_m[k]['frac']:.3f} vs committed {k_t[k]['frac']:.3f}" for k in k_m
               if k in k_t and k_m[k].get("frac") and k_t[k].get("frac") and k_m[k]["frac"] < 0.5 * k_t[k]["frac"]
               and max(k_m[k].get("ms", 0), k_t[k].get("ms", 0)) >= 0.03]        # (10-us kernels double with the box's mood: not a signal)
        return bad
    for name, r in results.items():
        bad = check(name, r.get("roofline"), (ref.get("extra", {}).get(name) or {}).get("roofline"))
        if bad:
            r["warning"] = "roofline leg below half of the committed value: " + "; ".join(bad)
    if headline is not None:
        bad = check("c2", headline.get("roofline"), ref.get("roofline"))
        if bad:
            headline["warning"] = "roofline leg below half of the committed value: " + "; ".join(bad)


def run_extras(args, dev):
    """Short runs of the other BASELINE configs appended to the default single-GPU line (driver-verifiable)."""
    extra = {}
    # default: the BASELINE five (c2 is the headline); everything else behind --all-extras
    plan = [("c2_b128", 200, 30), ("zinc_json_b128", 200, 30), ("hiv_json_b128", 200, 30), ("c1", 50, 20), ("c3", 100, 30), ("c4", 50, 20), ("c5", 3, 1),
            ("c5_layer", 3, 1)]      # (row f1 on the driver line: the C5 graph through a whole simple layer forward)
    if args.all_extras:
        plan = [("c1", 50, 20), ("c3", 100, 30), ("c3_drop", 50, 20), ("c4", 50, 20), ("c4_drop", 50, 20), ("c2c", 50, 20), ("c2e", 50, 20), ("c2et", 50, 20), ("zinc_json", 50, 20),
                ("pattern_json", 50, 20), ("c3_mega", 30, 10), ("c4_mega", 30, 10), ("zinc_json_b128", 200, 30), ("c2_b128", 200, 30), ("c1_b128", 200, 30), ("hiv_json_b128", 200, 30), ("c4_b128", 200, 30),
                ("c5", 3, 1), ("c5_layer", 3, 1)]
    for name, steps, warmup in plan:
        wl = dict(WORKLOADS[name])
        t0 = time.perf_counter()
        try:
            runner = run_c5 if wl["type_net"] == "op" else (run_c5_layer if wl["type_net"] == "layer_fwd" else run_layer_workload)
            res, batch = runner(args, wl, 0, 1, dev, steps=steps, warmup=warmup, tag=name)
            extra[name] = compact(res)
            extra[name]["config"] = wl["desc"]
            extra[name]["steps"], extra[name]["warmup"] = steps, warmup
            if name in ("c3", "c4", "c2_b128", "zinc_json_b128", "c1_b128", "hiv_json_b128", "c4_b128", "pattern_json") and wl["type_net"] not in ("op", "layer_fwd"):
                # batch-128 / batch-2048 legs are host-bound when every kernel is launched from Python (the reference's own regime): the
                # same step captured once into a HIP graph and replayed is what the GPU side costs
                import copy
                cap = copy.copy(args)
                cap.hipgraph = True
                try:
                    rc, _ = run_layer_workload(cap, dict(WORKLOADS[name]), 0, 1, dev, steps=100, warmup=20, tag=name)
                    extra[name]["captured_ms_per_step"] = rc["ms_per_step"]
                except Exception as exc:
                    extra[name]["captured_ms_per_step"] = None
                    extra[name]["captured_error"] = f"{type(exc).__name__}: {exc}"[:200]
                # ... and eager with autograd's backward on the CALLING thread (torch.autograd.set_multithreading_enabled(False), one line
                # of a training script): the hand-over to the engine's device thread and back is 60-110 us of a 0.1 ms step
                torch.autograd.set_multithreading_enabled(False)
                try:
                    rs, _ = run_layer_workload(args, dict(WORKLOADS[name]), 0, 1, dev, steps=steps, warmup=warmup, tag=name)
                    extra[name]["eager_st_ms_per_step"] = rs["ms_per_step"]
                except Exception as exc:
                    extra[name]["eager_st_error"] = f"{type(exc).__name__}: {exc}"[:200]
                finally:
                    torch.autograd.set_multithreading_enabled(True)
                if name.endswith("_b128"):
                    # ... and, opt-in (dgn_amd.ops.DIRECT_PARAM_GRADS): the block route's backward assigns the parameters' .grad itself
                    # instead of handing 33 gradients (towers) to autograd's AccumulateGrad nodes
                    from dgn_amd import ops as _ops
                    torch.autograd.set_multithreading_enabled(False)
                    _ops.DIRECT_PARAM_GRADS = True
                    try:
                        rd, _ = run_layer_workload(args, dict(WORKLOADS[name]), 0, 1, dev, steps=steps, warmup=warmup, tag=name)
                        extra[name]["eager_direct_ms_per_step"] = rd["ms_per_step"]
                    except Exception as exc:
                        extra[name]["eager_direct_error"] = f"{type(exc).__name__}: {exc}"[:200]
                    finally:
                        _ops.DIRECT_PARAM_GRADS = False
                        torch.autograd.set_multithreading_enabled(True)
                if name.endswith("_b128") or name == "c3":
                    # (c3: CIFAR10's 8-NN graphs TRAIN on the streaming kernels, their evaluation forward takes the graph-block route -- round 6)
                    try:
                        extra[name]["eval_fwd_ms"] = eval_forward_ms(dict(WORKLOADS[name]), dev)
                    except Exception as exc:
                        extra[name]["eval_fwd_error"] = f"{type(exc).__name__}: {exc}"[:200]
                if name.endswith("_b128"):
                    try:      # the eager step per layer inside a 4-layer stack (one backward() for four layers)
                        ms4, ms4_st = eager_stack_ms(dict(WORKLOADS[name]), dev)
                        extra[name]["stack4_ms"], extra[name]["stack4_st_ms"] = ms4 / 4, ms4_st / 4
                    except Exception as exc:
                        extra[name]["stack4_error"] = f"{type(exc).__name__}: {exc}"[:200]
            if name == "c1" and not args.no_cpu_baseline:
                # BASELINE configs[0] is quoted on the reference's CPU path: the same bounded CPU sample for it
                extra[name]["cpu_baseline"] = cpu_baseline(wl, batch, min(args.cpu_sample_graphs, len(batch["sizes"])), reps=3)
        except Exception as exc:          # an extra must never take the headline down with it -- but it is reported
            extra[name] = dict(error=f"{type(exc).__name__}: {exc}")
        extra[name]["wall_s"] = time.perf_counter() - t0
        del wl
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()
    regression_warnings(extra)
    if not args.all_extras:
        return extra
    try:
        extra["c2_inference"] = run_inference(args, dev)
    except Exception as exc:
        extra["c2_inference"] = dict(error=f"{type(exc).__name__}: {exc}")
    try:
        extra["zinc_net_b128"] = run_net(args, dev)
    except Exception as exc:
        extra["zinc_net_b128"] = dict(error=f"{type(exc).__name__}: {exc}")
    try:
        extra["c2_b128_bucketed"] = run_bucketed(args, dev)
    except Exception as exc:
        extra["c2_b128_bucketed"] = dict(error=f"{type(exc).__name__}: {exc}")
    try:      # the reference's shipped ZINC layer (complex, hidden 45) at its own batch size, the same way
        extra["zinc_json_b128_bucketed"] = run_bucketed(args, dev, workload="zinc_json_b128")
    except Exception as exc:
        extra["zinc_json_b128_bucketed"] = dict(error=f"{type(exc).__name__}: {exc}")
    return extra


FULL_LINE_PATHS = (os.path.join(ROOT, "gpurun_out", "bench_full.json"), os.path.join("/tmp", "dgn_bench_full.json"))
COMPACT_LIMIT = 4096


def compact_line(line):
    """The ONE line the driver parses (the r03 line grew to 32 KB and BENCH_r03.parsed came back null).
    Everything the contract names, the roofline of the dominant kernel with per-kernel (ms, bytes, frac) only, the CPU baseline,
    and per extra workload just (ms_per_step, value, roofline.frac, kernel).  The full record goes to a file, never to stdout."""
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in line}
    cfg = line.get("config") or {}
    out["config"] = {k: cfg[k] for k in ("workload", "edges_per_gpu", "nodes_per_gpu", "parallelism", "step") if k in cfg}
    r = line.get("roofline")
    if r:
        rr = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "frac_of_triad",
                                "triad_GBps", "model") if k in r}
        if isinstance(r.get("step"), dict):
            rr["step"] = {k: r["step"][k] for k in ("bytes_min", "frac", "kernel_us_sum") if k in r["step"]}
        if isinstance(rr.get("traffic_source"), str) and len(rr["traffic_source"]) > 120:
            rr["traffic_source"] = "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes; not re-measured in this run)"
        rr["kernels"] = {k: {kk: v[kk] for kk in ("ms", "bytes", "frac") if kk in v} for k, v in (r.get("kernels") or {}).items()
                         if isinstance(v, dict)}
        out["roofline"] = rr
    else:
        out["roofline"] = None
    c = line.get("cpu_baseline")
    out["cpu_baseline"] = ({k: c[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "host_cpus") if k in c}
                           if c else None)
    for k in ("ranks", "allreduce", "rccl_world", "warning"):
        if k in line:
            out[k] = line[k]
    if line.get("extra"):
        ex = {}
        for name, e in line["extra"].items():
            if "error" in e:
                ex[name] = dict(error=str(e["error"])[:80])
                continue
            ee = {k: e[k] for k in ("ms_per_step", "value", "scaling", "allreduce_ms_max", "captured_ms_per_step", "eager_st_ms_per_step", "eager_direct_ms_per_step", "eval_fwd_ms",
                                    "stack4_ms", "stack4_st_ms") if k in e}
            if e.get("roofline"):
                ee["frac"] = e["roofline"].get("frac")
                if e["roofline"].get("bound") == "mfma":      # (c5_layer: the fraction is of the fp32 MFMA peak, on the posttrans product)
                    ee["bound"] = "mfma"
                if isinstance(e["roofline"].get("step"), dict):
                    ee["step_frac"] = e["roofline"]["step"].get("frac")
            if e.get("cpu_baseline"):
                ee["cpu_edges_per_s"] = e["cpu_baseline"].get("value")
            ex[name] = ee
        out["extra"] = ex
    out["full_record"] = "gpurun_out/bench_full.json"

    def rnd(o):
        if isinstance(o, float):
            return float(f"{o:.6g}")
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return o
    out = rnd(out)
    # never exceed the limit: shed optional fields in a fixed order
    for victim in ("extra", "ranks", ("roofline", "kernels"), ("roofline", "model"), ("cpu_baseline", "sample")):
        if len(json.dumps(out)) < COMPACT_LIMIT:
            break
        if isinstance(victim, tuple):
            if isinstance(out.get(victim[0]), dict):
                out[victim[0]].pop(victim[1], None)
        else:
            out.pop(victim, None)
    return out


def emit(line, args):
    """Full record -> gpurun_out/bench_full.json (and /tmp); compact record -> the last (only) stdout line."""
    for path in FULL_LINE_PATHS:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(line, f)
        except OSError:
            pass
    sys.stdout.flush()
    print(json.dumps(compact_line(line)), flush=True)


DP_EXTRAS_TIMEOUT_S = 240


def run_dp_extras(args, rank, world, dev, line):
    """N > 1, default workload: the data-parallel config BASELINE.json / SURVEY 8(e) NAME -- configs[3]: ogbg-molhiv, batch 2048 -- rides on
    the same line, weak (one 2048-graph batch per rank) and strong (ONE 2048-graph batch, graphs sharded over the ranks by edge count:
    dist.shard_by_edges), each with the flat gradient all-reduce inside the timed step.  Every rank runs the legs (collectives);
    rank 0 reports them as extra.c4_dp_weak / extra.c4_dp_strong (VERDICT r05 missing #5).  The legs run AFTER rank 0 has assembled the
    headline record (`line`), under a watchdog: should a collective of an extra never return (one rank failed where the others did not),
    rank 0 still prints the headline line -- with the extra marked as timed out -- and every rank leaves."""
    if not ((world > 1 or os.environ.get("DGN_BENCH_DP_EXTRAS") == "1") and args.workload == "c2" and not args.no_extras and not args.hipgraph):
        return {}
    import copy, threading
    dp_extra = {}

    def give_up():
        if rank == 0 and line is not None:
            out = dict(line)
            out.setdefault("extra", {}).update(dp_extra)
            out["extra"]["c4_dp_timeout"] = dict(error=f"the data-parallel extras did not finish within {DP_EXTRAS_TIMEOUT_S} s; headline unaffected")
            emit(out, args)
        os._exit(0 if rank == 0 else 3)

    dog = threading.Timer(DP_EXTRAS_TIMEOUT_S, give_up)
    dog.daemon = True
    dog.start()
    try:
        for mode in ("weak", "strong"):
            a2 = copy.copy(args)
            a2.scaling = mode
            try:
                r2 = run_layer_workload(a2, dict(WORKLOADS["c4"]), rank, world, dev, steps=50, warmup=20, tag="c4")
                r2 = r2[0] if isinstance(r2, tuple) else r2
                dp_extra[f"c4_dp_{mode}"] = dict(ms_per_step=r2["ms_per_step"], value=r2["value"], scaling=r2["scaling"],
                                                 edges_per_rank=r2["edges_per_rank"], nodes_per_rank=r2["nodes_per_rank"],
                                                 allreduce_ms_max=(r2.get("allreduce") or {}).get("ms_max"),
                                                 config=WORKLOADS["c4"]["desc"] + (": one global batch sharded by edge count" if mode == "strong" else ": one batch per rank"))
            except Exception as exc:       # (symmetric across ranks: the same code on the same shapes)
                dp_extra[f"c4_dp_{mode}"] = dict(error=f"{type(exc).__name__}: {exc}"[:160])
            torch.cuda.empty_cache()
    finally:
        dog.cancel()
    return dp_extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the device needs ~30 steps after the set-up's idle time to reach its steady clocks (tools/step_ramp.py, profiles/r06_step_ramp.txt:
    # steps 6-25 average 1.5-2 % above steps 26+), so the default run warms up for 30 steps and times 100 (0.17 s of device time)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="layer workloads with --gpus > 1: 'weak' = every rank its own batch (default), 'strong' = one global "
                         "batch, graphs sharded over the ranks by edge count")
    ap.add_argument("--scale", type=float, default=1.0, help="c5 only: scale N and E")
    ap.add_argument("--c5-mode", default="partition", choices=["partition", "replicas"],
                    help="c5 with --gpus > 1: one graph split by destination ranges (strong scaling) or one graph per rank")
    ap.add_argument("--c5-eig", default="random", choices=["random", "lobpcg"],
                    help="c5: eig columns are random numbers (the sweep's cost does not depend on the values) or the graph's lowest "
                         "Laplacian eigenvectors computed by dgn_amd.eig.lobpcg_eigvecs")
    ap.add_argument("--c5-eig-iters", type=int, default=40)
    ap.add_argument("--aggregators", default=None, help="override the workload's aggregator string (experiments)")
    ap.add_argument("--scalers", default=None, help="override the workload's scaler string (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-results of the other configs in the default run")
    ap.add_argument("--all-extras", action="store_true",
                    help="default run: also the non-BASELINE legs (c2c, c2e, json configs, mega batches, batch-128, inference, nets)")
    ap.add_argument("--hipgraph", action="store_true",
                    help="layer workloads, 1 GPU: capture the step (edge weights + forward + backward) in a HIP graph and replay it")
    ap.add_argument("--gemm-tuning", default="off", choices=["off", "file", "tune"],
                    help="PyTorch TunableOp for whatever library GEMMs remain (small batches below the kernels' row thresholds): 'off' "
                         "(default: every Linear of the benched layers runs on this library's own kernels), 'file' replays "
                         "dgn_amd/tunableop_gfx950.csv, 'tune' tunes and rewrites it")
    ap.add_argument("--cpu-sample-graphs", type=int, default=1024)
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "weak"

    spawn_ranks_if_needed(args)
    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', 1)} ranks")
    configure_gemm_tuning(args.gemm_tuning)
    rank, world, local = ddist.init_from_env("nccl")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wl = dict(WORKLOADS[args.workload])
    if args.aggregators:
        wl["aggregators"] = args.aggregators
    if args.scalers:
        wl["scalers"] = args.scalers
    runner = run_c5 if wl["type_net"] == "op" else (run_c5_layer if wl["type_net"] == "layer_fwd" else run_layer_workload)
    res = runner(args, wl, rank, world, dev)
    if rank != 0:
        run_dp_extras(args, rank, world, dev, None)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    result, batch = res
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    line = dict(metric={"op": "dgn_aggregation_fwd_edges_per_sec", "layer_fwd": "dgn_layer_fwd_edges_per_sec"}.get(wl["type_net"], "dgn_layer_fwd_bwd_edges_per_sec"),
                value=result["value"], unit="edges/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=result["ms_per_step"], higher_is_better=True, scaling=result.get("scaling", "weak"), vs_baseline=None,
                dtype="f32",
                data="synthetic",
                config=dict(workload=args.workload + ": " + wl["desc"], edges_per_gpu=result["edges_per_rank"],
                            nodes_per_gpu=result["nodes_per_rank"], type_net=wl["type_net"], hidden=wl["hidden"],
                            aggregators=wl["aggregators"], scalers=wl["scalers"], towers=wl["towers"],
                            parallelism=result.get("parallelism") or (
                                f"dp{world}: {'one global batch, graphs sharded by edge count' if result.get('scaling') == 'strong' else 'one batch per rank'}"
                                f", flat-gradient all-reduce ({backend} = RCCL, {world} ranks)" if world > 1 else "single GPU"),
                            step=("edge weights + layer forward + backward" + (" + gradient all-reduce" if world > 1 else "")
                                  + (" (HIP graph replay)" if args.hipgraph else ""))
                            if wl["type_net"] not in ("op", "layer_fwd") else ("aggregation forward" if wl["type_net"] == "op" else "layer forward, no gradients")),
                roofline=result.get("roofline"))
    if result.get("step_kernels"):
        line["step_kernels"] = result["step_kernels"]      # (full record only: compact_line does not carry it)
    if torch.distributed.is_initialized():
        line["rccl_world"] = dict(backend=backend, world_size=torch.distributed.get_world_size())
    if result.get("ranks"):
        line["ranks"] = result["ranks"]
        if result.get("allreduce"):
            line["allreduce"] = result["allreduce"]
    if world == 1 and not args.no_cpu_baseline and batch is not None:
        line["cpu_baseline"] = cpu_baseline(wl, batch, min(args.cpu_sample_graphs, len(batch["sizes"])))
    else:
        line["cpu_baseline"] = None
    if world == 1 and args.workload == "c2" and not args.hipgraph and not args.aggregators and not args.scalers:
        regression_warnings({}, headline=line)
    if world == 1 and args.workload == "c2" and not args.no_extras and not args.hipgraph and not args.aggregators and not args.scalers:
        del res, result, batch
        torch.cuda.empty_cache()
        line["extra"] = run_extras(args, dev)
    dp_extra = run_dp_extras(args, rank, world, dev, line)
    if dp_extra:
        line.setdefault("extra", {}).update(dp_extra)
    emit(line, args)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
