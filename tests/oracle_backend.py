"""Test-only stand-in for dgn_amd.ops.directional_aggregate backed by the CPU oracle, so that the host
logic around the kernels (layer algebra, data-parallel harness) can be exercised without a GPU.
It is installed by monkeypatching inside tests; the product package never imports it."""
import torch

from oracle import dgn_oracle as orc


def oracle_directional_aggregate(graph, plan, avg_log, x_src=None, x_dst=None, m_edge=None, x_in=None, eig=None,
                                 n_towers=1, weights=None, tower_major=False, x_pair=None, edge_type=None):
    if x_pair is not None:
        half = x_pair.shape[1] // 2
        x_src, x_dst = x_pair[:, :half], x_pair[:, half:]
    src = graph.src.long()
    dst = torch.repeat_interleave(torch.arange(graph.num_nodes), graph.in_degree)
    msg = 0
    if x_src is not None:
        msg = msg + x_src[src]
    if x_dst is not None:
        msg = msg + x_dst[dst]
    if m_edge is not None:
        msg = msg + (m_edge if edge_type is None else m_edge[edge_type.long()])
    eig = graph.ndata["eig"] if eig is None else eig
    F_ = msg.shape[1]
    if x_in is None:
        x_in = torch.zeros(graph.num_nodes, F_)
    avg = torch.tensor(float(avg_log))
    names = [a for a in plan.aggregators if a != "__x_in__"]
    out = orc.aggregate_graph(src, dst, graph.num_nodes, msg, eig, x_in, names, list(plan.scalers), avg)
    if len(names) != len(plan.aggregators):      # h_in pass-through block (single identity scaler by contract)
        assert len(plan.scalers) == 1 and list(plan.aggregators)[-1] == "__x_in__"
        out = torch.cat([out, x_in], dim=1)
    if n_towers > 1:      # [S][A][T][Ft] -> [T][S][A][Ft]
        N = out.shape[0]
        SA = out.shape[1] // F_
        out = out.view(N, SA, n_towers, F_ // n_towers).permute(0, 2, 1, 3).reshape(N, -1)
    if tower_major:       # [N, T*K] -> [T, N, K]
        out = out.view(out.shape[0], n_towers, -1).transpose(0, 1).contiguous()
    return out


def oracle_scale_combine(z, scale, bias, row_scale):
    """torch restatement of dgn_scale_combine_forward (autograd does the backward)."""
    T, N, W = z.shape
    S = 1 if scale is None else scale.shape[1]
    fo = W // S
    y = z.view(T, N, S, fo)
    y = (y * scale.view(1, N, S, 1)).sum(2) if scale is not None else y[:, :, 0]
    y = y.transpose(0, 1).reshape(N, T * fo)
    if bias is not None:
        y = y + bias.reshape(1, -1)
    if row_scale is not None:
        y = y * row_scale.reshape(-1, 1)
    return y


def oracle_bn_tail(x, bns, training, relu=False, residual=None):
    """plain-torch restatement of dgn_amd.ops.bn_tail (the nn.BatchNorm1d modules themselves)."""
    if not isinstance(bns, (list, tuple)):
        bns = [bns]
    w = x.shape[1] // len(bns)
    y = torch.cat([b(x[:, i * w:(i + 1) * w]) for i, b in enumerate(bns)], dim=1) if len(bns) > 1 else bns[0](x)
    y = torch.relu(y) if relu else y
    return y + residual if residual is not None else y


def oracle_bn_tail_fused(x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, relu=False,
                         residual=None):
    """plain-torch restatement of dgn_amd.ops.bn_tail_fused (F.batch_norm on the concatenated channels)."""
    y = torch.nn.functional.batch_norm(x, running_mean, running_var, gamma, beta, training, momentum, eps)
    if training and num_batches_tracked is not None:
        with torch.no_grad():
            num_batches_tracked.add_(1)
    y = torch.relu(y) if relu else y
    return y + residual if residual is not None else y


def oracle_combine_bn_tail(z, scale, bias, row_scale, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                           relu=False, residual=None):
    """plain-torch restatement of dgn_amd.ops.combine_bn_tail (training mode)."""
    y = oracle_scale_combine(z, scale, bias, row_scale)
    return oracle_bn_tail_fused(y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, True, relu, residual)
