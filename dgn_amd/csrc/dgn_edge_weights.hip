// Per-edge directional weights for gfx950, computed once per (graph, eig) and shared by every
// layer and tower of a forward/backward pass (the reference recomputes them inside every
// aggregator of every layer: realworld_benchmark/nets/aggregators.py:35-71).
//
//   delta_j = eig[src_j, k] - eig[i, k]                                   (aggregators.py:36/:49)
//   ABSNORM  : w_j = delta_j / (sum_j |delta_j| + eps)                     (:49-50, :56-57, :36-38)
//   BALANCED : w_j = (relu(d_j)/(sum relu(d)+eps) + relu(-d_j)/(sum relu(-d)+eps)) / 2   (:63-69)
//   SOFTMAX  : w_j = softmax_j(alpha |delta_j|)                            (:43-44)
//
// One wavefront per destination row, lanes across the row's CSR slots; the row normalisers are a
// wave reduction.  Hub rows are cut into hub_chunk-slot slices (slice statistics -> combine ->
// slice write) so that a 10^6-edge row is not one wave's serial loop.
#include <hip/hip_runtime.h>

#include <cmath>

#include "dgn_common.hpp"

namespace dgn {
namespace {

struct EwParams {
    const int32_t* indptr;
    const int32_t* src;
    int64_t n_nodes;
    int64_t row_base;      // global id of row 0 (destination-range shard), 0 otherwise
    int32_t hub_threshold;
    int32_t hub_chunk;
    const int32_t* hub_rows;
    const int32_t* hub_chunk_ptr;
    const int32_t* chunk_hub;
    int64_t n_hub;
    int64_t n_chunks;
    const float* eig;      // [N, ld]  (node mode)
    const float* eig_s;    // [E, ld]  (slot mode)
    const float* eig_d;    // [E, ld]
    int64_t ld_eig;
    int32_t n_ch;
    DgnChannel ch[DGN_MAX_CH];
    float* w;
    int64_t ld_w;
    float* slice_stats;    // [n_chunks][DGN_MAX_CH][5]
    float* hub_stats;      // [n_hub][DGN_MAX_CH][5]
};

struct Stats {  // per channel
    float sabs, spos, sneg, mx, se;
};

__device__ __forceinline__ float edge_delta(const EwParams& p, int row, int e, int col) {
    if (p.eig_s) return p.eig_s[(int64_t)e * p.ld_eig + col] - p.eig_d[(int64_t)e * p.ld_eig + col];
    return p.eig[(int64_t)p.src[e] * p.ld_eig + col] - p.eig[(p.row_base + row) * p.ld_eig + col];
}

// statistics of slots [beg, end) of row `row` for every channel (wave-wide results)
__device__ __forceinline__ void range_stats(Stats (&st)[DGN_MAX_CH], const EwParams& p, int row, int beg, int end) {
    const int lane = lane_id();
    bool any_softmax = false;
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        st[c] = Stats{0.f, 0.f, 0.f, -INFINITY, 0.f};
        if (c < p.n_ch && p.ch[c].kind == DGN_W_SOFTMAX) any_softmax = true;
    }
    for (int e = beg + lane; e < end; e += kWave) {
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            if (c < p.n_ch) {
                const float d = edge_delta(p, row, e, p.ch[c].eig_col);
                st[c].sabs += fabsf(d);
                st[c].spos += fmaxf(d, 0.f);
                st[c].sneg += fmaxf(-d, 0.f);
                st[c].mx = fmaxf(st[c].mx, p.ch[c].alpha * fabsf(d));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        if (c < p.n_ch) {
            st[c].sabs = wave_sum(st[c].sabs);
            st[c].spos = wave_sum(st[c].spos);
            st[c].sneg = wave_sum(st[c].sneg);
            st[c].mx = wave_max(st[c].mx);
        }
    }
    if (any_softmax) {
        for (int e = beg + lane; e < end; e += kWave) {
#pragma unroll
            for (int c = 0; c < DGN_MAX_CH; ++c) {
                if (c < p.n_ch && p.ch[c].kind == DGN_W_SOFTMAX) {
                    const float d = edge_delta(p, row, e, p.ch[c].eig_col);
                    st[c].se += expf(p.ch[c].alpha * fabsf(d) - st[c].mx);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c)
            if (c < p.n_ch && p.ch[c].kind == DGN_W_SOFTMAX) st[c].se = wave_sum(st[c].se);
    }
}

__device__ __forceinline__ void range_write(const Stats (&st)[DGN_MAX_CH], const EwParams& p, int row, int beg, int end) {
    const int lane = lane_id();
    for (int e = beg + lane; e < end; e += kWave) {
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            if (c < p.n_ch) {
                const float d = edge_delta(p, row, e, p.ch[c].eig_col);
                const float eps = p.ch[c].eps;
                float v;
                if (p.ch[c].kind == DGN_W_ABSNORM) {
                    v = d / (st[c].sabs + eps);
                } else if (p.ch[c].kind == DGN_W_BALANCED) {
                    v = (fmaxf(d, 0.f) / (st[c].spos + eps) + fmaxf(-d, 0.f) / (st[c].sneg + eps)) / 2.f;
                } else {
                    v = expf(p.ch[c].alpha * fabsf(d) - st[c].mx) / st[c].se;
                }
                p.w[(int64_t)c * p.ld_w + e] = v;
            }
        }
    }
}

// Rows of at most kFlatMax slots (every row of a molecule batch) are handled one per THREAD by ew_rows_flat:
// per-edge weights are a few scalars per slot, so a wave per 2-slot row would be all launch overhead.
constexpr int kFlatMax = 16;

__global__ __launch_bounds__(256) void ew_rows_flat(const EwParams p) {
    const int64_t row64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row64 >= p.n_nodes) return;
    const int row = (int)row64;
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    const int deg = end - beg;
    if (deg == 0 || deg > kFlatMax) return;
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        if (c >= p.n_ch) break;
        const int col = p.ch[c].eig_col;
        const float eps = p.ch[c].eps, alpha = p.ch[c].alpha;
        const int kind = p.ch[c].kind;
        float sabs = 0.f, spos = 0.f, sneg = 0.f, mx = -INFINITY;
        for (int e = beg; e < end; ++e) {
            const float d = edge_delta(p, row, e, col);
            sabs += fabsf(d); spos += fmaxf(d, 0.f); sneg += fmaxf(-d, 0.f);
            mx = fmaxf(mx, alpha * fabsf(d));
        }
        float se = 0.f;
        if (kind == DGN_W_SOFTMAX)
            for (int e = beg; e < end; ++e) se += expf(alpha * fabsf(edge_delta(p, row, e, col)) - mx);
        for (int e = beg; e < end; ++e) {
            const float d = edge_delta(p, row, e, col);
            float v;
            if (kind == DGN_W_ABSNORM) v = d / (sabs + eps);
            else if (kind == DGN_W_BALANCED) v = (fmaxf(d, 0.f) / (spos + eps) + fmaxf(-d, 0.f) / (sneg + eps)) / 2.f;
            else v = expf(alpha * fabsf(d) - mx) / se;
            p.w[(int64_t)c * p.ld_w + e] = v;
        }
    }
}

__global__ __launch_bounds__(kBlock) void ew_rows(const EwParams p) {
    const int64_t row64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row64 >= p.n_nodes) return;
    const int row = uniform_i((int)row64);
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    if (end - beg <= kFlatMax || end - beg > p.hub_threshold) return;    // short rows: ew_rows_flat; hubs: slices
    Stats st[DGN_MAX_CH];
    range_stats(st, p, row, beg, end);
    range_write(st, p, row, beg, end);
}

__device__ __forceinline__ void slice_bounds(const EwParams& p, int chunk, int& hub, int& row, int& beg, int& end) {
    hub = p.chunk_hub[chunk];
    row = p.hub_rows[hub];
    const int rbeg = p.indptr[row], rend = p.indptr[row + 1];
    beg = rbeg + (chunk - p.hub_chunk_ptr[hub]) * p.hub_chunk;
    end = min(beg + p.hub_chunk, rend);
}

__global__ __launch_bounds__(kBlock) void ew_hub_slice_stats(const EwParams p) {
    const int64_t chunk64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk64 >= p.n_chunks) return;
    const int chunk = uniform_i((int)chunk64);
    int hub, row, beg, end;
    slice_bounds(p, chunk, hub, row, beg, end);
    Stats st[DGN_MAX_CH];
    range_stats(st, p, row, beg, end);
    if (lane_id() == 0) {
        float* o = p.slice_stats + (int64_t)chunk * DGN_MAX_CH * 5;
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            o[c * 5 + 0] = st[c].sabs; o[c * 5 + 1] = st[c].spos; o[c * 5 + 2] = st[c].sneg;
            o[c * 5 + 3] = st[c].mx; o[c * 5 + 4] = st[c].se;
        }
    }
}

// one thread per (hub row, channel): merge the slice statistics in slot order
__global__ void ew_hub_combine(const EwParams p) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_hub * DGN_MAX_CH) return;
    const int hub = (int)(t / DGN_MAX_CH), c = (int)(t % DGN_MAX_CH);
    float sabs = 0.f, spos = 0.f, sneg = 0.f, mx = -INFINITY, se = 0.f;
    for (int k = p.hub_chunk_ptr[hub]; k < p.hub_chunk_ptr[hub + 1]; ++k) {
        const float* s = p.slice_stats + ((int64_t)k * DGN_MAX_CH + c) * 5;
        sabs += s[0]; spos += s[1]; sneg += s[2];
        const float m2 = s[3], e2 = s[4];
        const float M = fmaxf(mx, m2);
        if (M > -INFINITY) se = se * expf(mx - M) + e2 * expf(m2 - M);
        mx = M;
    }
    float* o = p.hub_stats + ((int64_t)hub * DGN_MAX_CH + c) * 5;
    o[0] = sabs; o[1] = spos; o[2] = sneg; o[3] = mx; o[4] = se;
}

__global__ __launch_bounds__(kBlock) void ew_hub_slice_write(const EwParams p) {
    const int64_t chunk64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk64 >= p.n_chunks) return;
    const int chunk = uniform_i((int)chunk64);
    int hub, row, beg, end;
    slice_bounds(p, chunk, hub, row, beg, end);
    Stats st[DGN_MAX_CH];
    const float* s = p.hub_stats + (int64_t)hub * DGN_MAX_CH * 5;
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) st[c] = Stats{s[c * 5 + 0], s[c * 5 + 1], s[c * 5 + 2], s[c * 5 + 3], s[c * 5 + 4]};
    range_write(st, p, row, beg, end);
}

size_t ws_bytes_for(const DgnGraph* g) {
    if (!g || g->n_hub <= 0) return 0;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return up((size_t)g->n_chunks * DGN_MAX_CH * 5 * sizeof(float)) + up((size_t)g->n_hub * DGN_MAX_CH * 5 * sizeof(float));
}

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" size_t dgn_edge_weights_workspace_bytes(const DgnGraph* g, int32_t n_ch) {
    (void)n_ch;
    return ws_bytes_for(g);
}

extern "C" int dgn_edge_weights(const DgnGraph* g, const float* eig, const float* eig_s_edge, const float* eig_d_edge,
                                int64_t ld_eig, int32_t n_ch, const DgnChannel* ch, float* w, int64_t ld_w, void* ws,
                                size_t ws_bytes, void* stream_) {
    if (!g || !ch || !w) { set_error("null graph/channels/output"); return DGN_ERR_INVALID; }
    if (n_ch < 1 || n_ch > DGN_MAX_CH) { set_error("n_ch=%d outside 1..%d", n_ch, DGN_MAX_CH); return DGN_ERR_INVALID; }
    if (!eig && !(eig_s_edge && eig_d_edge)) { set_error("need eig, or eig_s_edge and eig_d_edge"); return DGN_ERR_INVALID; }
    if (g->n_nodes > INT32_MAX - 1 || g->n_edges > INT32_MAX - 1) { set_error("graph outside the int32 CSR range"); return DGN_ERR_INVALID; }
    if (ld_w < g->n_edges) { set_error("ld_w smaller than n_edges"); return DGN_ERR_INVALID; }
    for (int c = 0; c < n_ch; ++c) {
        if (ch[c].kind < DGN_W_ABSNORM || ch[c].kind > DGN_W_SOFTMAX) { set_error("unknown channel kind %d", ch[c].kind); return DGN_ERR_INVALID; }
        if (ch[c].eig_col < 0 || ch[c].eig_col >= ld_eig) { set_error("channel %d: eig column %d outside 0..%lld", c, ch[c].eig_col, (long long)ld_eig - 1); return DGN_ERR_INVALID; }
    }
    if (g->n_nodes == 0 || g->n_edges == 0) return DGN_OK;
    if (g->n_hub > 0 && (!ws || ws_bytes < ws_bytes_for(g))) { set_error("workspace too small: need %zu bytes", ws_bytes_for(g)); return DGN_ERR_WORKSPACE; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    EwParams p{};
    p.indptr = g->indptr; p.src = g->src; p.n_nodes = g->n_nodes; p.row_base = g->row_base;
    p.n_hub = g->n_hub; p.n_chunks = g->n_hub > 0 ? g->n_chunks : 0;
    p.hub_threshold = g->n_hub > 0 ? g->hub_threshold : INT32_MAX;
    p.hub_chunk = g->hub_chunk; p.hub_rows = g->hub_rows; p.hub_chunk_ptr = g->hub_chunk_ptr; p.chunk_hub = g->chunk_hub;
    if (eig_s_edge && eig_d_edge) { p.eig_s = eig_s_edge; p.eig_d = eig_d_edge; } else { p.eig = eig; }
    p.ld_eig = ld_eig; p.n_ch = n_ch;
    for (int c = 0; c < n_ch; ++c) p.ch[c] = ch[c];
    p.w = w; p.ld_w = ld_w;
    if (g->n_hub > 0) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        p.slice_stats = static_cast<float*>(ws);
        p.hub_stats = reinterpret_cast<float*>(static_cast<char*>(ws) + up((size_t)g->n_chunks * DGN_MAX_CH * 5 * sizeof(float)));
    }
    hipLaunchKernelGGL(ew_rows_flat, dim3((unsigned)((p.n_nodes + 255) / 256)), dim3(256), 0, stream, p);
    if (g->max_in_degree == 0 || g->max_in_degree > kFlatMax) {     // skipped when every row is known to be short
        const unsigned nb = (unsigned)((p.n_nodes + kWavesPerBlock - 1) / kWavesPerBlock);
        hipLaunchKernelGGL(ew_rows, dim3(nb), dim3(kBlock), 0, stream, p);
    }
    if (p.n_hub > 0) {
        const unsigned ns = (unsigned)((p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
        hipLaunchKernelGGL(ew_hub_slice_stats, dim3(ns), dim3(kBlock), 0, stream, p);
        const unsigned nc = (unsigned)((p.n_hub * DGN_MAX_CH + 255) / 256);
        hipLaunchKernelGGL(ew_hub_combine, dim3(nc), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(ew_hub_slice_write, dim3(ns), dim3(kBlock), 0, stream, p);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
