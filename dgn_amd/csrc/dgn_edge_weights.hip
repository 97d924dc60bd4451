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
#include <cstdlib>

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
    int32_t vw;            // vector width (floats) of the eig row loads: 4, 2 or 1
    int32_t n_need;        // floats of a row prefix the channels read (largest column + 1, capped at kEigRegs)
    DgnChannel ch[DGN_MAX_CH];
    float* w;
    int64_t ld_w;
    float* slice_stats;    // [n_chunks][DGN_MAX_CH][5]
    float* hub_stats;      // [n_hub][DGN_MAX_CH][5]
};

struct Stats {  // per channel
    float sabs, spos, sneg, mx, se;
};

// ---- eig access ---------------------------------------------------------------------------------------------------
// An edge needs eig[src, col_c] for every channel c.  The columns of one node are adjacent (eig is [N, K] row major, K = 4..7),
// so the row prefix up to the largest column used is fetched with ONE (K = 4: 16-byte) vector load per endpoint and the
// channels pick their column from registers -- instead of one dependent 4-byte gather per channel and pass (round 1: 6 gathers
// per edge on C5's three channels, 0.41 TB/s).  `vw` = widest vector the row stride / base alignment allow (host-chosen);
// columns >= kEigRegs fall back to scalar loads.
constexpr int kEigRegs = 8;

__device__ __forceinline__ void load_eig_prefix(float (&v)[kEigRegs], const float* rowp, int n_need, int vw) {
#pragma unroll
    for (int i = 0; i < kEigRegs; ++i) v[i] = 0.f;
    if (vw == 4) {
        const float4 a = *reinterpret_cast<const float4*>(rowp);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        if (n_need > 4) {
            const float4 b = *reinterpret_cast<const float4*>(rowp + 4);
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
    } else if (vw == 2) {
#pragma unroll
        for (int i = 0; i < kEigRegs / 2; ++i) {
            if (2 * i < n_need) {
                const float2 a = *reinterpret_cast<const float2*>(rowp + 2 * i);
                v[2 * i] = a.x; v[2 * i + 1] = a.y;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kEigRegs; ++i)
            if (i < n_need) v[i] = rowp[i];
    }
}
__device__ __forceinline__ float pick_col(const float (&v)[kEigRegs], int col) {
    float r = 0.f;                      // compare chain: a dynamic register index would go to scratch
#pragma unroll
    for (int i = 0; i < kEigRegs; ++i) r = (i == col) ? v[i] : r;
    return r;
}

// destination side of a row (node mode): wave-uniform scalars
struct RowEig {
    float d[DGN_MAX_CH];
    __device__ __forceinline__ void load(const EwParams& p, int row) {
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) d[c] = (c < p.n_ch && p.eig) ? p.eig[(p.row_base + row) * p.ld_eig + p.ch[c].eig_col] : 0.f;
    }
};

// delta_c of slot e for every channel
__device__ __forceinline__ void edge_deltas(float (&dl)[DGN_MAX_CH], const EwParams& p, const RowEig& re, int e) {
    float vs[kEigRegs];
    if (p.eig_s) {      // slot mode: both endpoints per slot
        float vd[kEigRegs];
        load_eig_prefix(vs, p.eig_s + (int64_t)e * p.ld_eig, p.n_need, p.vw);
        load_eig_prefix(vd, p.eig_d + (int64_t)e * p.ld_eig, p.n_need, p.vw);
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            dl[c] = 0.f;
            if (c < p.n_ch) {
                const int col = p.ch[c].eig_col;
                dl[c] = col < kEigRegs ? pick_col(vs, col) - pick_col(vd, col)
                                       : p.eig_s[(int64_t)e * p.ld_eig + col] - p.eig_d[(int64_t)e * p.ld_eig + col];
            }
        }
        return;
    }
    const float* rowp = p.eig + (int64_t)p.src[e] * p.ld_eig;
    load_eig_prefix(vs, rowp, p.n_need, p.vw);
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        dl[c] = 0.f;
        if (c < p.n_ch) {
            const int col = p.ch[c].eig_col;
            dl[c] = (col < kEigRegs ? pick_col(vs, col) : rowp[col]) - re.d[c];
        }
    }
}

__device__ __forceinline__ float weight_of(const DgnChannel& ch, const Stats& st, float d) {
    if (ch.kind == DGN_W_ABSNORM) return d / (st.sabs + ch.eps);
    if (ch.kind == DGN_W_BALANCED) return (fmaxf(d, 0.f) / (st.spos + ch.eps) + fmaxf(-d, 0.f) / (st.sneg + ch.eps)) / 2.f;
    return expf(ch.alpha * fabsf(d) - st.mx) / st.se;
}

// statistics of slots [beg, end) of row `row` for every channel (wave-wide results).  The gathered deltas are PARKED in the
// output planes (w[c][e] := delta): the softmax pass and range_write read them back -- sequential, coalesced 4-byte reads of
// what the same lane wrote -- instead of gathering eig a second time (rows longer than one slot batch and hub slices: 39 % of
// C5's edges).
__device__ __forceinline__ void range_stats(Stats (&st)[DGN_MAX_CH], const EwParams& p, int row, int beg, int end) {
    const int lane = lane_id();
    RowEig re;
    re.load(p, row);
    bool any_softmax = false;
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        st[c] = Stats{0.f, 0.f, 0.f, -INFINITY, 0.f};
        if (c < p.n_ch && p.ch[c].kind == DGN_W_SOFTMAX) any_softmax = true;
    }
    for (int e = beg + lane; e < end; e += kWave) {
        float dl[DGN_MAX_CH];
        edge_deltas(dl, p, re, e);
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            if (c < p.n_ch) {
                const float d = dl[c];
                p.w[(int64_t)c * p.ld_w + e] = d;
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
            for (int c = 0; c < DGN_MAX_CH; ++c)
                if (c < p.n_ch && p.ch[c].kind == DGN_W_SOFTMAX)
                    st[c].se += expf(p.ch[c].alpha * fabsf(p.w[(int64_t)c * p.ld_w + e]) - st[c].mx);
        }
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c)
            if (c < p.n_ch && p.ch[c].kind == DGN_W_SOFTMAX) st[c].se = wave_sum(st[c].se);
    }
}

// weights from the parked deltas, in place (same lane -> slot mapping as range_stats)
__device__ __forceinline__ void range_write(const Stats (&st)[DGN_MAX_CH], const EwParams& p, int row, int beg, int end) {
    const int lane = lane_id();
    for (int e = beg + lane; e < end; e += kWave) {
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            if (c < p.n_ch) {
                float* q = p.w + (int64_t)c * p.ld_w + e;
                *q = weight_of(p.ch[c], st[c], *q);
            }
        }
    }
}

// Three row classes, one kernel each (a kernel skips the rows of the other classes; the host skips a kernel when the graph's
// largest in-degree says its class is empty):
//   deg <= kFlatMax (4: every row of a molecule batch)   one THREAD per row, all deltas in registers (ew_rows_flat)
//   deg <= kGroup  (16)                                   16 lanes per row, four rows per wave (ew_rows_g16)
//   longer                                                one wave per row (ew_rows); hub rows: slices
// Every class gathers eig ONCE per edge (the deltas wait in registers between the statistics and the write) except wave rows
// longer than 64 slots.
constexpr int kFlatMax = 4;
constexpr int kGroup = 16;

template <int FM>
__device__ __forceinline__ void ew_flat_body(const EwParams& p, int64_t row64);
__global__ __launch_bounds__(256) void ew_rows_flat(const EwParams p) { ew_flat_body<kFlatMax>(p, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }
// Round 6: batches whose LARGEST in-degree is 5 .. 8 (k-NN graphs: CIFAR10 / MNIST superpixels, 8 neighbours) -- a thread per row there too,
// eight gathers in flight per lane, one launch; the 16-lanes-per-row class walked four rows per one-wave workgroup, three dependent round
// trips each (c3_mega, 960 k rows: 0.382 ms = 0.07 of the HBM roofline).
__global__ __launch_bounds__(256) void ew_rows_flat8(const EwParams p) { ew_flat_body<8>(p, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }
template <int FM>
__device__ __forceinline__ void ew_flat_body(const EwParams& p, int64_t row64) {
    if (row64 >= p.n_nodes) return;
    const int row = (int)row64;
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    const int deg = end - beg;
    if (deg == 0 || deg > FM) return;
    RowEig re;
    re.load(p, row);
    float dl[FM][DGN_MAX_CH];
#pragma unroll
    for (int j = 0; j < FM; ++j) {           // all gathers issued before the first use
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) dl[j][c] = 0.f;
        if (j < deg) edge_deltas(dl[j], p, re, beg + j);
    }
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        if (c >= p.n_ch) break;
        Stats st{0.f, 0.f, 0.f, -INFINITY, 0.f};
#pragma unroll
        for (int j = 0; j < FM; ++j) {       // slot order, as the reference's sum over the mailbox dimension
            if (j < deg) {
                const float d = dl[j][c];
                st.sabs += fabsf(d); st.spos += fmaxf(d, 0.f); st.sneg += fmaxf(-d, 0.f);
                st.mx = fmaxf(st.mx, p.ch[c].alpha * fabsf(d));
            }
        }
        if (p.ch[c].kind == DGN_W_SOFTMAX) {
#pragma unroll
            for (int j = 0; j < FM; ++j)
                if (j < deg) st.se += expf(p.ch[c].alpha * fabsf(dl[j][c]) - st.mx);
        }
#pragma unroll
        for (int j = 0; j < FM; ++j)
            if (j < deg) p.w[(int64_t)c * p.ld_w + beg + j] = weight_of(p.ch[c], st, dl[j][c]);
    }
}

__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = kGroup / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kGroup);
    return v;
}
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = kGroup / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kGroup));
    return v;
}

// Row classes by ballot: a one-wave workgroup looks at 64 consecutive rows (one coalesced read of their degrees), lists the ones
// of its class in LDS and works through that list -- four rows at a time here (16 lanes each), one row at a time in ew_rows.
// On a power-law graph most rows belong to another class: launching a wave (or 16 lanes) per ROW made the dispatcher the cost of
// these kernels (10 M one-wave workgroups = 2.2 ms at ~4.6 workgroups/ns), and several waves per workgroup tie a finished wave's
// slot to the longest row of the group (measured: 2.4 -> 4.0 ms).
// (R = candidate rows per wave: 64 on large graphs; 4 / 1 -- the plain 16-lanes-per-row / wave-per-row mapping -- on small
// batches, where a wave walking through 20 rows one after the other would serialise their gather latencies: CIFAR10-like batch
// 0.018 ms with R = 1 / 4, 0.037 ms with R = 64)
template <int R>
__device__ __forceinline__ int class_rows(const EwParams& p, int64_t base, int lo, int hi, int* list, int& my_beg, int& my_deg) {
    const int lane = lane_id();
    const int64_t r = base + lane;
    my_beg = 0;
    my_deg = 0;
    if (lane < R && r < p.n_nodes) {
        my_beg = p.indptr[r];
        my_deg = p.indptr[r + 1] - my_beg;
    }
    const bool mine = my_deg > lo && my_deg <= hi;
    const uint64_t mask = __ballot(mine);
    if (mine) list[__popcll(mask & ((1ull << lane) - 1))] = lane;
    return __popcll(mask);
}

template <int R>
__device__ __forceinline__ void ew_g16_body(const EwParams& p, int64_t blk, int* list);
template <int R>
__global__ __launch_bounds__(kWave) void ew_rows_g16(const EwParams p) {
    __shared__ int list[kWave];
    ew_g16_body<R>(p, blockIdx.x, list);
}
template <int R>
__device__ __forceinline__ void ew_g16_body(const EwParams& p, int64_t blk, int* list) {
    const int64_t base = blk * R;
    int my_beg, my_deg;
    const int count = class_rows<R>(p, base, kFlatMax, kGroup, list, my_beg, my_deg);
    const int lane = lane_id(), grp = lane / kGroup, l = lane % kGroup;
    for (int it = 0; it * (kWave / kGroup) < count; ++it) {
        const int idx = it * (kWave / kGroup) + grp;
        const bool have = idx < count;
        const int rl = have ? list[idx] : 0;                      // (LDS written by this wave: program order suffices)
        const int row = (int)(base + rl);
        const int beg = __shfl(my_beg, rl, kWave), deg = __shfl(my_deg, rl, kWave);
        const bool in = have && l < deg;
        RowEig re;
        re.load(p, have ? row : 0);
        float dl[DGN_MAX_CH];
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) dl[c] = 0.f;
        if (in) edge_deltas(dl, p, re, beg + l);
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            if (c >= p.n_ch) break;
            const float d = dl[c];
            Stats st{0.f, 0.f, 0.f, -INFINITY, 0.f};
            st.sabs = group_sum(fabsf(d));
            st.spos = group_sum(fmaxf(d, 0.f));
            st.sneg = group_sum(fmaxf(-d, 0.f));
            st.mx = group_max(in ? p.ch[c].alpha * fabsf(d) : -INFINITY);
            if (p.ch[c].kind == DGN_W_SOFTMAX) st.se = group_sum(in ? expf(p.ch[c].alpha * fabsf(d) - st.mx) : 0.f);
            if (in) p.w[(int64_t)c * p.ld_w + beg + l] = weight_of(p.ch[c], st, d);
        }
    }
}

template <int R>
__device__ __forceinline__ void ew_rows_body(const EwParams& p, int64_t blk, int* list);
template <int R>
__global__ __launch_bounds__(kWave) void ew_rows(const EwParams p) {
    __shared__ int list[kWave];
    ew_rows_body<R>(p, blockIdx.x, list);
}
// Small batches (fewer than 2^20 rows) whose largest in-degree is unknown or above 4: the three row classes in ONE launch of one-wave
// workgroups -- block ranges [0, nf) thread-per-row, [nf, nf + ng) 16 lanes per row, the rest a wave per row.  Three launches of
// 6-10 us each were as long as the forward sweep itself on a CIFAR10 batch of 128 graphs (VERDICT r03, weak list).
__global__ __launch_bounds__(kWave) void ew_rows_small(const EwParams p, int nf, int ng) {
    __shared__ int list[kWave];
    const int b = blockIdx.x;
    if (b < nf) ew_flat_body<kFlatMax>(p, (int64_t)b * kWave + threadIdx.x);
    else if (b < nf + ng) ew_g16_body<kWave / kGroup>(p, b - nf, list);
    else ew_rows_body<1>(p, b - nf - ng, list);
}
template <int R>
__device__ __forceinline__ void ew_rows_body(const EwParams& p, int64_t blk, int* list) {
    const int64_t base = blk * R;
    int my_beg, my_deg;
    const int count = class_rows<R>(p, base, kGroup, p.hub_threshold, list, my_beg, my_deg);
    for (int it = 0; it < count; ++it) {
    const int rl = list[it];
    const int row = uniform_i((int)(base + rl));
    const int beg = bcast_i(my_beg, rl), deg = bcast_i(my_deg, rl), end = beg + deg;
    if (deg <= kWave) {
        // the row is one slot per lane: ONE gather per edge, the deltas stay in registers between statistics and write
        const int lane = lane_id();
        const bool in = lane < deg;
        RowEig re;
        re.load(p, row);
        float dl[DGN_MAX_CH];
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) dl[c] = 0.f;
        if (in) edge_deltas(dl, p, re, beg + lane);
        Stats st[DGN_MAX_CH];
#pragma unroll
        for (int c = 0; c < DGN_MAX_CH; ++c) {
            st[c] = Stats{0.f, 0.f, 0.f, -INFINITY, 0.f};
            if (c < p.n_ch) {
                const float d = dl[c];            // (0 on idle lanes: adds nothing; alpha * 0 must not win the max)
                st[c].sabs = wave_sum(fabsf(d));
                st[c].spos = wave_sum(fmaxf(d, 0.f));
                st[c].sneg = wave_sum(fmaxf(-d, 0.f));
                st[c].mx = wave_max(in ? p.ch[c].alpha * fabsf(d) : -INFINITY);
                if (p.ch[c].kind == DGN_W_SOFTMAX) st[c].se = wave_sum(in ? expf(p.ch[c].alpha * fabsf(d) - st[c].mx) : 0.f);
            }
        }
        if (in) {
#pragma unroll
            for (int c = 0; c < DGN_MAX_CH; ++c)
                if (c < p.n_ch) p.w[(int64_t)c * p.ld_w + beg + lane] = weight_of(p.ch[c], st[c], dl[c]);
        }
        continue;
    }
    Stats st[DGN_MAX_CH];
    range_stats(st, p, row, beg, end);
    range_write(st, p, row, beg, end);
    }
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

// one wave per hub row: lanes stride over the row's slices (a 2.4 M-edge row has 2 400 of them), wave reductions join them;
// the softmax partial sums are rescaled to the row's maximum
__global__ __launch_bounds__(kWave) void ew_hub_combine(const EwParams p) {
    const int hub = blockIdx.x;
    if (hub >= p.n_hub) return;
    const int lane = lane_id();
    const int k0 = p.hub_chunk_ptr[hub], k1 = p.hub_chunk_ptr[hub + 1];
#pragma unroll
    for (int c = 0; c < DGN_MAX_CH; ++c) {
        float sabs = 0.f, spos = 0.f, sneg = 0.f, mx = -INFINITY;
        for (int k = k0 + lane; k < k1; k += kWave) {
            const float* s = p.slice_stats + ((int64_t)k * DGN_MAX_CH + c) * 5;
            sabs += s[0]; spos += s[1]; sneg += s[2];
            mx = fmaxf(mx, s[3]);
        }
        sabs = wave_sum(sabs); spos = wave_sum(spos); sneg = wave_sum(sneg); mx = wave_max(mx);
        float se = 0.f;
        for (int k = k0 + lane; k < k1; k += kWave) {
            const float* s = p.slice_stats + ((int64_t)k * DGN_MAX_CH + c) * 5;
            if (s[3] > -INFINITY) se += s[4] * expf(s[3] - mx);
        }
        se = wave_sum(se);
        if (lane == 0) {
            float* o = p.hub_stats + ((int64_t)hub * DGN_MAX_CH + c) * 5;
            o[0] = sabs; o[1] = spos; o[2] = sneg; o[3] = mx; o[4] = se;
        }
    }
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
    int max_col = 0;
    for (int c = 0; c < n_ch; ++c) { p.ch[c] = ch[c]; max_col = ch[c].eig_col > max_col ? ch[c].eig_col : max_col; }
    p.n_need = max_col + 1 < 8 ? max_col + 1 : 8;
    {   // widest vector load every row start and the needed prefix allow (a vector never reaches past its row's last column
        // rounded up to the vector width, so ld_eig must be a multiple of it)
        auto ok = [&](int v) {
            auto al = [&](const float* q) { return !q || (reinterpret_cast<uintptr_t>(q) % (4 * v)) == 0; };
            return ld_eig % v == 0 && al(eig) && al(eig_s_edge) && al(eig_d_edge);
        };
        p.vw = ok(4) ? 4 : (ok(2) ? 2 : 1);
    }
    p.w = w; p.ld_w = ld_w;
    if (g->n_hub > 0) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        p.slice_stats = static_cast<float*>(ws);
        p.hub_stats = reinterpret_cast<float*>(static_cast<char*>(ws) + up((size_t)g->n_chunks * DGN_MAX_CH * 5 * sizeof(float)));
    }
    // (round 6: 2^20 -> 2^19 rows.  Below it the three classes are one launch of N / 64 + N / 4 + N one-wave workgroups, most of which
    //  leave at once: at 960 k k-NN rows (c3_mega) the dispatcher alone is 0.26 ms of a 0.383-ms launch; the ballot path takes 0.139)
    static const int64_t big_min = getenv("DGN_EW_BIG_MIN") ? atoll(getenv("DGN_EW_BIG_MIN")) : (1 << 19);
    const bool big = p.n_nodes >= big_min;         // row classes by ballot (64 candidate rows per wave) vs a wave / 16 lanes per row
    static const bool no_merge = getenv("DGN_EW_SEPARATE") != nullptr;
    static const bool no_flat8 = getenv("DGN_EW_NO_FLAT8") != nullptr;
    if (!no_flat8 && g->max_in_degree > kFlatMax && g->max_in_degree <= 8) {      // every row at most 8 slots: a thread per row, one launch
        hipLaunchKernelGGL(ew_rows_flat8, dim3((unsigned)((p.n_nodes + 255) / 256)), dim3(256), 0, stream, p);
    } else if (!big && !no_merge && (g->max_in_degree == 0 || g->max_in_degree > kFlatMax)) {
        const int nf = (int)((p.n_nodes + kWave - 1) / kWave), ng = (int)((p.n_nodes + 3) / 4);
        const bool rows = g->max_in_degree == 0 || g->max_in_degree > kGroup;
        hipLaunchKernelGGL(ew_rows_small, dim3((unsigned)(nf + ng + (rows ? p.n_nodes : 0))), dim3(kWave), 0, stream, p, nf, ng);
    } else {
    hipLaunchKernelGGL(ew_rows_flat, dim3((unsigned)((p.n_nodes + 255) / 256)), dim3(256), 0, stream, p);
    if (g->max_in_degree == 0 || g->max_in_degree > kFlatMax)        // (skipped when every row is known to be shorter)
    {
        if (big) hipLaunchKernelGGL(ew_rows_g16<kWave>, dim3((unsigned)((p.n_nodes + kWave - 1) / kWave)), dim3(kWave), 0, stream, p);
        else hipLaunchKernelGGL(ew_rows_g16<kWave / kGroup>, dim3((unsigned)((p.n_nodes + 3) / 4)), dim3(kWave), 0, stream, p);
    }
    if (g->max_in_degree == 0 || g->max_in_degree > kGroup) {
        if (big) hipLaunchKernelGGL(ew_rows<kWave>, dim3((unsigned)((p.n_nodes + kWave - 1) / kWave)), dim3(kWave), 0, stream, p);
        else hipLaunchKernelGGL(ew_rows<1>, dim3((unsigned)p.n_nodes), dim3(kWave), 0, stream, p);
    }
    }
    if (p.n_hub > 0) {
        const unsigned ns = (unsigned)((p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
        hipLaunchKernelGGL(ew_hub_slice_stats, dim3(ns), dim3(kBlock), 0, stream, p);
        hipLaunchKernelGGL(ew_hub_combine, dim3((unsigned)p.n_hub), dim3(kWave), 0, stream, p);
        hipLaunchKernelGGL(ew_hub_slice_write, dim3(ns), dim3(kBlock), 0, stream, p);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
