"""Multi-GPU path on CPU: world_size-2 gloo processes.  Graphs are sharded by edge count, every rank runs
the layer on its shard, ONE flat all-reduce averages the gradients.  Checked against a single process
that runs the same shards sequentially and averages the gradients (BatchNorm sees per-shard statistics in
both).  The aggregation itself is served by the oracle here (tests/oracle_backend.py);
the product kernels are GPU-only and are covered by the -m gpu tests."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    from dgn_amd import synth
    b = synth.molecule_batch(12, seed=5, laplacian_eig=False)
    sizes = b["sizes"].tolist()
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    gid_of_node = torch.repeat_interleave(torch.arange(len(sizes)), b["sizes"])
    edge_gid = gid_of_node[b["dst"]]
    edges_per_graph = torch.bincount(edge_gid, minlength=len(sizes)).tolist()
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(b["num_nodes"], 10, generator=gen)
    return b, offs, edge_gid, edges_per_graph, h


def _shard_batch(b, offs, edge_gid, h, graph_ids):
    """Sub-batch holding the given graphs (relabelled consecutively)."""
    import dgn_amd
    node_chunks, src, dst, new_off = [], [], [], 0
    for gi in graph_ids:
        lo, hi = offs[gi], offs[gi + 1]
        m = edge_gid == gi
        src.append(b["src"][m] - lo + new_off)
        dst.append(b["dst"][m] - lo + new_off)
        node_chunks.append(torch.arange(lo, hi))
        new_off += hi - lo
    nodes = torch.cat(node_chunks)
    g = dgn_amd.DGNGraph(torch.cat(src), torch.cat(dst), new_off, eig=b["eig"][nodes])
    return g, h[nodes], b["snorm_n"][nodes]


def _build_layer(type_net="towers"):
    import dgn_amd
    torch.manual_seed(3)
    layer = dgn_amd.DGNLayer(10, 10, 0.0, True, True, "mean max dir1-dx dir1-av", "identity amplification attenuation",
                             {"log": torch.tensor(1.0)}, type_net, True, towers=5 if type_net == "towers" else 1, edge_features=False,
                             edge_dim=0).model
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
    return layer


def _install_oracle_backend():
    import dgn_amd.dgn_layer as dl
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import (oracle_bn_tail, oracle_bn_tail_fused, oracle_combine_bn_tail, oracle_directional_aggregate,
                                oracle_scale_combine)
    dl.directional_aggregate = oracle_directional_aggregate
    dl.scale_combine = oracle_scale_combine
    dl.bn_tail = oracle_bn_tail
    dl.bn_tail_fused = oracle_bn_tail_fused
    dl.combine_bn_tail = oracle_combine_bn_tail


def _loss_backward(layer, g, h, snorm):
    for p in layer.parameters():
        p.grad = None
    y = layer(g, h, None, snorm)
    (y * y).mean().backward()


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from dgn_amd import dist as ddist
    r, w, _ = ddist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    _install_oracle_backend()
    b, offs, edge_gid, epg, h = _make_problem()
    shards = ddist.shard_by_edges(epg, world)
    layer = _build_layer()
    g, hs, sn = _shard_batch(b, offs, edge_gid, h, shards[rank])
    _loss_backward(layer, g, hs, sn)
    ddist.FlatGradAllReduce(layer.parameters())()
    if rank == 0:
        torch.save({n: p.grad.clone() for n, p in layer.named_parameters()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def _sync_worker(rank, world, port, out_path, type_net):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from dgn_amd import dist as ddist
    from dgn_amd import ops
    import dgn_amd.dgn_layer as dl
    ddist.init_from_env("gloo")
    _install_oracle_backend()
    dl.bn_tail = ops.bn_tail                      # (the product's own tail: with SyncBatchNorm1d modules it must leave the fused kernels)
    b, offs, edge_gid, epg, h = _make_problem()
    shards = ddist.shard_by_edges(epg, world)
    layer = ddist.convert_sync_batchnorm(_build_layer(type_net))
    assert all(type(m).__name__ != "BatchNorm1d" for m in layer.modules())
    g, hs, sn = _shard_batch(b, offs, edge_gid, h, shards[rank])
    _loss_backward(layer, g, hs, sn)
    ddist.FlatGradAllReduce(layer.parameters())()
    if rank == 0:
        torch.save(dict(grads={n: p.grad.clone() for n, p in layer.named_parameters()}, buffers={n: v.clone() for n, v in layer.named_buffers()},
                        keys=list(layer.state_dict().keys())), out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,type_net", [(2, "towers"), (3, "simple")])
def test_sync_batchnorm_equals_one_process_over_the_whole_batch(tmp_path, world, type_net):
    """SURVEY 8(e), optional SyncBN: with ``dist.convert_sync_batchnorm`` a data-parallel step over (uneven) shards is the step of one
    process over the union batch -- parameter gradients after the flat all-reduce, BatchNorm running statistics, state_dict keys."""
    out_path = str(tmp_path / "sync.pt")
    mp.spawn(_sync_worker, args=(world, _free_port(), out_path, type_net), nprocs=world, join=True)
    got = torch.load(out_path)

    from dgn_amd import dist as ddist
    import dgn_amd.dgn_layer as dl
    saved = (dl.directional_aggregate, dl.scale_combine, dl.bn_tail, dl.bn_tail_fused, dl.combine_bn_tail)
    try:
        _install_oracle_backend()
        b, offs, edge_gid, epg, h = _make_problem()
        shards = ddist.shard_by_edges(epg, world)
        layer = _build_layer(type_net)
        order = [gi for sh in shards for gi in sh]
        g, hs, sn = _shard_batch(b, offs, edge_gid, h, order)                 # the union batch, graphs in shard order
        y = layer(g, hs, None, sn)
        rows = [sum(offs[gi + 1] - offs[gi] for gi in sh) for sh in shards]
        loss = sum((part * part).mean() for part in torch.split(y, rows)) / world
        loss.backward()
    finally:
        dl.directional_aggregate, dl.scale_combine, dl.bn_tail, dl.bn_tail_fused, dl.combine_bn_tail = saved
    assert got["keys"] == list(layer.state_dict().keys())
    for n, p in layer.named_parameters():
        torch.testing.assert_close(got["grads"][n], p.grad, rtol=2e-5, atol=2e-6, msg=n)
    for n, v in layer.named_buffers():
        torch.testing.assert_close(got["buffers"][n], v, rtol=1e-5, atol=1e-6, msg=n)


def test_shard_by_edges_balances():
    from dgn_amd.dist import shard_by_edges
    counts = [50, 10, 40, 30, 20, 60, 5, 45]
    shards = shard_by_edges(counts, 3)
    assert sorted(i for s in shards for i in s) == list(range(len(counts)))
    loads = [sum(counts[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(counts)
    assert shard_by_edges(counts, 3) == shards          # deterministic
    assert shard_by_edges(counts, 1) == [list(range(len(counts)))]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_dp_gradients_match_sequential_shards(tmp_path, world):
    """world 4: the shards are UNEVEN (graph counts and node counts differ per rank): FlatGradAllReduce's 1 / world scaling and the
    per-rank BatchNorm statistics against the same shards run one after the other."""
    out_path = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = torch.load(out_path)

    # single process: the same shards one after the other, gradients averaged
    from dgn_amd import dist as ddist
    import dgn_amd.dgn_layer as dl
    saved = (dl.directional_aggregate, dl.scale_combine, dl.bn_tail)
    try:
        _install_oracle_backend()
        b, offs, edge_gid, epg, h = _make_problem()
        shards = ddist.shard_by_edges(epg, world)
        acc = None
        for rnk in range(world):
            layer = _build_layer()
            g, hs, sn = _shard_batch(b, offs, edge_gid, h, shards[rnk])
            _loss_backward(layer, g, hs, sn)
            grads = {n: p.grad.clone() for n, p in layer.named_parameters()}
            acc = grads if acc is None else {n: acc[n] + grads[n] for n in acc}
        want = {n: v / world for n, v in acc.items()}
    finally:
        dl.directional_aggregate, dl.scale_combine, dl.bn_tail = saved
    assert set(got) == set(want)
    for n in want:
        torch.testing.assert_close(got[n], want[n], rtol=1e-5, atol=1e-6, msg=n)


def test_layer_algebra_matches_reference_on_cpu(golden):
    """Host-side layer algebra (P/Q pretrans decomposition, all towers in one sweep, scaler folding, split
    posttrans) against the reference's layer outputs, with the aggregation served by the oracle."""
    import numpy as np
    import dgn_amd
    import dgn_amd.dgn_layer as dl
    saved = (dl.directional_aggregate, dl.scale_combine, dl.bn_tail)
    T = torch.from_numpy
    try:
        _install_oracle_backend()
        g = golden("g4_layers")
        src, dst, N = T(g["src"]), T(g["dst"]), int(g["N"])
        for name in g["cases"].tolist():
            meta = g[f"{name}/meta"].tolist()
            layer = dgn_amd.DGNLayer(in_dim=int(meta[1]), out_dim=int(meta[2]), dropout=0.0, graph_norm=True, batch_norm=True,
                                     aggregators=meta[3], scalers=meta[4], avg_d={"log": torch.tensor(float(meta[5]))},
                                     type_net=meta[0], residual=True, towers=int(meta[6]), divide_input=bool(int(meta[7])),
                                     edge_features=bool(int(meta[8])), edge_dim=int(meta[9]),
                                     pretrans_layers=int(meta[10]), posttrans_layers=int(meta[11])).model
            layer.load_state_dict({k[len(name) + 5:]: T(g[k]) for k in g.files if k.startswith(f"{name}/sd::")})
            layer.train(bool(int(meta[12])))
            graph = dgn_amd.DGNGraph(src, dst, N, eig=T(g[f"{name}/eig"]))
            h = T(g[f"{name}/h"]).clone().requires_grad_(True)
            e = T(g[f"{name}/e"]).clone().requires_grad_(True)
            y = layer(graph, h, e, T(g["snorm_n"]))
            np.testing.assert_allclose(y.detach().numpy(), g[f"{name}/y"], rtol=2e-5, atol=2e-5, err_msg=name)
    finally:
        dl.directional_aggregate, dl.scale_combine, dl.bn_tail = saved


# ---- one giant graph split by destination ranges ---------------------------------------

def _sharded_worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        from dgn_amd import dist as ddist
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_backend import oracle_directional_aggregate
        import dgn_amd
        ddist.init_from_env("gloo")
        gen = torch.Generator().manual_seed(5)
        N, F_ = 57, 6
        deg = torch.randint(0, 9, (N,), generator=gen)
        deg[7] = 60                                           # one long row: the cut must stay monotone around it
        indptr = torch.zeros(N + 1, dtype=torch.long)
        indptr[1:] = torch.cumsum(deg, 0)
        src = torch.randint(0, N, (int(indptr[-1]),), generator=gen)
        X, eig = torch.randn(N, F_, generator=gen), torch.randn(N, 3, generator=gen)
        plan = dgn_amd.make_plan(["mean", "max", "dir1-dx", "dir2-av"], ["identity", "amplification"])
        full = dgn_amd.DGNGraph.from_csr(indptr, src)
        ref = oracle_directional_aggregate(full, plan, 1.1, x_src=X, x_in=X, eig=eig)
        ranges = ddist.row_ranges_by_edges(indptr, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == N and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        r0, r1 = ranges[rank]
        shard = ddist.shard_rows(indptr, src, r0, r1)
        assert shard.num_nodes == r1 - r0 and shard.num_src == N and shard.row_base == r0
        assert shard.num_edges == int(indptr[r1] - indptr[r0])
        # rows of the shard through the oracle: same sources, the shard's destinations shifted back to global ids
        # (the scalers read the degree per row, so the padding rows around the shard do not matter)
        emb = torch.zeros(N + 1, dtype=torch.long)
        emb[r0 + 1:r1 + 1] = shard.indptr.long()[1:]
        emb[r1 + 1:] = emb[r1]
        local = oracle_directional_aggregate(dgn_amd.DGNGraph.from_csr(emb, shard.src.long()), plan, 1.1, x_src=X, x_in=X, eig=eig)[r0:r1]
        local = local.detach().requires_grad_(True)
        gathered = ddist.all_gather_rows(local, ranges)
        ok = gathered.shape == ref.shape and torch.allclose(gathered, ref, rtol=1e-6, atol=1e-6)
        # the gather is differentiable: rank r weighs the gathered rows with (r + 1) * Cw, so the gradient of a rank's own
        # rows is the SUM over ranks = (1 + 2 + ... + world) * Cw[r0:r1]
        Cw = torch.randn(ref.shape, generator=torch.Generator().manual_seed(9))
        (gathered * ((rank + 1) * Cw)).sum().backward()
        ok = ok and torch.allclose(local.grad, (world * (world + 1) / 2) * Cw[r0:r1], rtol=1e-6, atol=1e-6)
        q.put((rank, bool(ok), [tuple(r) for r in ranges]))
        dist.destroy_process_group()
    except Exception as exc:   # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_single_graph(world):
    """world 4: row 7 holds 60 of the ~290 edges, i.e. most of one rank's share -- the cut behind it swallows the next one, the shards
    are uneven in rows (all_gather_rows pads to the longest shard) and the ranges must stay a monotone partition."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"
    assert all(r[2] == res[0][2] for r in res)
    ranges = res[0][2]
    rows = [b - a for a, b in ranges]
    assert sum(rows) == 57 and (world == 2 or len(set(rows)) > 1), ranges          # uneven shards at world 4
