// Fused DGN aggregation sweep for gfx950 (MI355X): forward and backward.
//
// One wavefront owns one destination row (x one 64*VEC-wide feature tile).  Lane l holds VEC
// consecutive features, so every gathered source row is one fully coalesced load instruction.
// The row's CSR slots are fetched 64 at a time (source id + per-edge weights, coalesced) and
// broadcast lane -> SGPR with v_readlane, so the gather address is scalar-base + lane offset.
// All aggregators requested for the layer share the single read of each message; the degree
// scalers and the reference's concat order are applied in the epilogue.
//
// Reference semantics restated here: realworld_benchmark/nets/aggregators.py:8-71,
// scalers.py:7-18, dgn_layer.py:161-173 (paths relative to the reference tree).
//
// Rows longer than hub_threshold ("hub rows" of power-law graphs) are cut into hub_chunk-edge
// slices: a slice kernel writes partial accumulators to the workspace and a combine kernel
// merges them in slot order and runs the epilogue (all accumulators are associative).
//
// Backward = (optional) recompute of the row's accumulators with first-occurrence arg tracking
// for max/min, per-row coefficient vectors, then one emit pass over the row's slots:
//   dm_j = c0 + cv*m_j + sum_c (w_jc*cs_c + |w_jc|*ca_c) + [j==argmax]*gmax + [j==argmin]*gmin
// scattered with hardware fp32 atomics into d x_src[src_j]; d x_dst / d x_in are per-row.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "dgn_common.hpp"

namespace dgn {
namespace {

enum : uint32_t {
    NEED_SUM = 1u, NEED_SQ = 2u, NEED_MAX = 4u, NEED_MIN = 8u,
    NEED_XIN = 16u,      // some dx aggregator reads x_in
    NEED_RECOMP = 32u,   // backward must recompute the accumulators
    NEED_M_EMIT = 64u    // backward emit pass needs the message value (var/std)
};

struct AggParams {
    // graph
    const int32_t* indptr;
    const int32_t* src;
    int64_t n_nodes;
    int32_t hub_threshold;
    int32_t hub_chunk;
    const int32_t* hub_rows;
    const int32_t* hub_chunk_ptr;
    const int32_t* chunk_hub;
    int64_t n_hub;
    int64_t n_chunks;
    // message
    int32_t F;
    int32_t Ft;  // F / n_towers
    const float* x_src;  int64_t ld_src;
    const float* x_dst;  int64_t ld_dst;
    const float* m_edge; int64_t ld_edge;
    const float* x_in;   int64_t ld_in;
    const float* w;      int64_t ld_w;
    const float* log_deg;
    // spec
    int32_t n_agg;
    int32_t agg_total;
    int32_t agg_offset;
    int32_t n_ch;
    int32_t n_scalers;
    int32_t n_towers;
    int8_t op[DGN_MAX_AGG];
    int8_t ch[DGN_MAX_AGG];
    int8_t scaler[DGN_MAX_SCALERS];
    uint8_t ch_signed;   // bit c: sum_j w_jc m_j needed
    uint8_t ch_abs;      // bit c: sum_j |w_jc| m_j needed
    float avg_log;
    float eps;
    uint32_t need;
    // forward output / backward input
    float* out;           int64_t ld_out;
    const float* g_out;   int64_t ld_gout;
    // backward sinks
    float* g_src;  int64_t ldg_src;
    float* g_dst;  int64_t ldg_dst;
    float* g_edge; int64_t ldg_edge;
    float* g_in;   int64_t ldg_in;
    // workspace (hub rows)
    float* part;          // [n_chunks][n_slots][F]
    float* part_sw;       // [n_chunks][DGN_MAX_CH]
    float* coef;          // [n_hub][n_coef][F]  (backward)
    int32_t n_slots;
    int32_t n_coef;
};

// accumulator slot ids in the hub workspace
constexpr int SLOT_SUM = 0, SLOT_SQ = 1, SLOT_MAX = 2, SLOT_MIN = 3, SLOT_AMAX = 4, SLOT_AMIN = 5, SLOT_W0 = 6;
// coefficient slot ids (backward hub path)
constexpr int COEF_C0 = 0, COEF_CV = 1, COEF_GMAX = 2, COEF_GMIN = 3, COEF_AMAX = 4, COEF_AMIN = 5, COEF_W0 = 6;

template <int VEC, int NCH, bool STATS, bool TRACK>
struct Acc {
    float sum[VEC];
    float sq[STATS ? VEC : 1];
    float mx[STATS ? VEC : 1];
    float mn[STATS ? VEC : 1];
    int amax[(STATS && TRACK) ? VEC : 1];
    int amin[(STATS && TRACK) ? VEC : 1];
    float ws[NCH][VEC];
    float wa[NCH][VEC];
    float sw[NCH];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum[i] = 0.f;
        if constexpr (STATS) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sq[i] = 0.f;
                mx[i] = -INFINITY;
                mn[i] = INFINITY;
                if constexpr (TRACK) { amax[i] = -1; amin[i] = -1; }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            sw[c] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { ws[c][i] = 0.f; wa[c][i] = 0.f; }
        }
    }

    // one message; pos = CSR slot (for first-occurrence arg tracking)
    __device__ __forceinline__ void add(const float (&m)[VEC], const float (&wk)[NCH], int pos, const AggParams& p) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum[i] += m[i];
        if constexpr (STATS) {
            if (p.need & NEED_SQ) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) sq[i] = fmaf(m[i], m[i], sq[i]);
            }
            if (p.need & NEED_MAX) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    if constexpr (TRACK) {
                        if (m[i] > mx[i]) { mx[i] = m[i]; amax[i] = pos; }
                    } else {
                        mx[i] = fmaxf(mx[i], m[i]);
                    }
                }
            }
            if (p.need & NEED_MIN) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    if constexpr (TRACK) {
                        if (m[i] < mn[i]) { mn[i] = m[i]; amin[i] = pos; }
                    } else {
                        mn[i] = fminf(mn[i], m[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < p.n_ch) {
                sw[c] += wk[c];
                if (p.ch_signed & (1u << c)) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) ws[c][i] = fmaf(wk[c], m[i], ws[c][i]);
                }
                if (p.ch_abs & (1u << c)) {
                    float a = fabsf(wk[c]);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) wa[c][i] = fmaf(a, m[i], wa[c][i]);
                }
            }
        }
    }

    // merge a later partial (slot order preserved: strict compare keeps the first occurrence)
    __device__ __forceinline__ void merge(const Acc& o, const AggParams& p) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum[i] += o.sum[i];
        if constexpr (STATS) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sq[i] += o.sq[i];
                if constexpr (TRACK) {
                    if (o.mx[i] > mx[i]) { mx[i] = o.mx[i]; amax[i] = o.amax[i]; }
                    if (o.mn[i] < mn[i]) { mn[i] = o.mn[i]; amin[i] = o.amin[i]; }
                } else {
                    mx[i] = fmaxf(mx[i], o.mx[i]);
                    mn[i] = fminf(mn[i], o.mn[i]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            sw[c] += o.sw[c];
#pragma unroll
            for (int i = 0; i < VEC; ++i) { ws[c][i] += o.ws[c][i]; wa[c][i] += o.wa[c][i]; }
        }
    }
};

// message of slot e coming from node s:  x_src[s] + x_dst[row] + m_edge[e]
template <int VEC>
__device__ __forceinline__ void load_msg(float (&m)[VEC], const AggParams& p, int s, int e, int f0, const float (&xd)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) m[i] = xd[i];
    if (p.x_src) {
        float t[VEC];
        ldv<VEC>(t, p.x_src + (int64_t)s * p.ld_src + f0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] += t[i];
    }
    if (p.m_edge) {
        float t[VEC];
        ldv<VEC>(t, p.m_edge + (int64_t)e * p.ld_edge + f0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] += t[i];
    }
}

// accumulate CSR slots [beg, end) of one destination row
template <int VEC, int NCH, bool STATS, bool TRACK>
__device__ __forceinline__ void accumulate_range(Acc<VEC, NCH, STATS, TRACK>& acc, const AggParams& p, int beg, int end,
                                                 int f0, bool active, const float (&xd)[VEC]) {
    const int lane = lane_id();
    for (int base = beg; base < end; base += kWave) {
        const int e_l = base + lane;
        const bool in = e_l < end;
        const int my_src = in ? p.src[e_l] : 0;
        float my_w[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) my_w[c] = (in && c < p.n_ch) ? p.w[(int64_t)c * p.ld_w + e_l] : 0.f;
        const int cnt = min(kWave, end - base);
        int k = 0;
        for (; k + 4 <= cnt; k += 4) {
            float m[4][VEC];
            float wk[4][NCH];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = bcast_i(my_src, k + u);
#pragma unroll
                for (int c = 0; c < NCH; ++c) wk[u][c] = bcast_f(my_w[c], k + u);
                if (active) load_msg<VEC>(m[u], p, s, base + k + u, f0, xd);
            }
            if (active) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc.add(m[u], wk[u], base + k + u, p);
            }
        }
        for (; k < cnt; ++k) {
            float m[VEC];
            float wk[NCH];
            const int s = bcast_i(my_src, k);
#pragma unroll
            for (int c = 0; c < NCH; ++c) wk[c] = bcast_f(my_w[c], k);
            if (active) {
                load_msg<VEC>(m, p, s, base + k, f0, xd);
                acc.add(m, wk, base + k, p);
            }
        }
    }
}

__device__ __forceinline__ float scaler_factor(int kind, float logd, float avg) {
    if (kind == DGN_SCALE_AMPLIFICATION) return logd / avg;
    if (kind == DGN_SCALE_ATTENUATION) return avg / logd;
    return 1.f;
}

// output column of (scaler s, aggregator a, feature f):  [T][S][A][Ft]
__device__ __forceinline__ int64_t out_col(const AggParams& p, int s, int a, int f) {
    const int t = f / p.Ft;
    const int ft = f - t * p.Ft;
    return (int64_t)t * ((int64_t)p.n_scalers * p.agg_total * p.Ft) + ((int64_t)s * p.agg_total + p.agg_offset + a) * p.Ft + ft;
}

// raw variance exactly as aggregators.py:25-27 (two rounded products, one subtraction)
__device__ __forceinline__ float raw_var(float sq, float sum, float d) {
    const float ms = __fdiv_rn(sq, d);
    const float mean = __fdiv_rn(sum, d);
    return __fsub_rn(ms, __fmul_rn(mean, mean));
}

// r = sum_j w_j m_j - (sum_j w_j) x  with the reference's two roundings (aggregators.py:52/:59)
__device__ __forceinline__ float dx_residual(float ws, float sw, float x) {
    const float t = sw * x;
    return ws - t;
}

template <int VEC, int NCH, bool STATS, bool TRACK>
__device__ __forceinline__ void agg_value(float (&val)[VEC], int op, int c, const Acc<VEC, NCH, STATS, TRACK>& acc,
                                          const float (&xin)[VEC], float d, const AggParams& p) {
    // select the channel's accumulators with compares (keeps them in registers)
    float wsv[VEC], wav[VEC], swv = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { wsv[i] = 0.f; wav[i] = 0.f; }
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
        if (cc == c) {
            swv = acc.sw[cc];
#pragma unroll
            for (int i = 0; i < VEC; ++i) { wsv[i] = acc.ws[cc][i]; wav[i] = acc.wa[cc][i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float v = 0.f;
        switch (op) {
            case DGN_AGG_MEAN: v = __fdiv_rn(acc.sum[i], d); break;
            case DGN_AGG_SUM: v = acc.sum[i]; break;
            case DGN_AGG_MAX: if constexpr (STATS) v = acc.mx[i]; break;
            case DGN_AGG_MIN: if constexpr (STATS) v = acc.mn[i]; break;
            case DGN_AGG_VAR: if constexpr (STATS) v = fmaxf(raw_var(acc.sq[i], acc.sum[i], d), 0.f); break;
            case DGN_AGG_STD:
                if constexpr (STATS) v = __fsqrt_rn(__fadd_rn(fmaxf(raw_var(acc.sq[i], acc.sum[i], d), 0.f), p.eps));
                break;
            case DGN_AGG_DIR_AV: v = wav[i]; break;
            case DGN_AGG_DIR_WSUM: v = wsv[i]; break;
            case DGN_AGG_DIR_DX_NO_ABS: v = __fsub_rn(wsv[i], __fmul_rn(swv, xin[i])); break;
            case DGN_AGG_DIR_DX: v = fabsf(__fsub_rn(wsv[i], __fmul_rn(swv, xin[i]))); break;
        }
        val[i] = v;
    }
}

// write one finished row: aggregator values x scalers in the reference concat order
template <int VEC, int NCH, bool STATS>
__device__ __forceinline__ void write_row(const Acc<VEC, NCH, STATS, false>& acc, const AggParams& p, int row, int deg,
                                          int f0) {
    float* orow = p.out + (int64_t)row * p.ld_out;
    if (deg == 0) {
        float z[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) z[i] = 0.f;
        for (int s = 0; s < p.n_scalers; ++s)
            for (int a = 0; a < p.n_agg; ++a) stv<VEC>(orow + out_col(p, s, a, f0), z);
        return;
    }
    float xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xin[i] = 0.f;
    if (p.need & NEED_XIN) ldv<VEC>(xin, p.x_in + (int64_t)row * p.ld_in + f0);
    const float d = (float)deg;
    float fac[DGN_MAX_SCALERS];
    const float logd = p.log_deg ? p.log_deg[row] : 0.f;
#pragma unroll
    for (int s = 0; s < DGN_MAX_SCALERS; ++s) fac[s] = s < p.n_scalers ? scaler_factor(p.scaler[s], logd, p.avg_log) : 1.f;
    for (int a = 0; a < p.n_agg; ++a) {
        float val[VEC];
        agg_value<VEC, NCH, STATS, false>(val, p.op[a], p.ch[a], acc, xin, d, p);
#pragma unroll
        for (int s = 0; s < DGN_MAX_SCALERS; ++s) {
            if (s < p.n_scalers) {
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = p.scaler[s] == DGN_SCALE_IDENTITY ? val[i] : val[i] * fac[s];
                stv<VEC>(orow + out_col(p, s, a, f0), o);
            }
        }
    }
}

// ---- hub workspace I/O ----------------------------------------------------------------------

template <int VEC, int NCH, bool STATS, bool TRACK>
__device__ __forceinline__ void store_partial(const Acc<VEC, NCH, STATS, TRACK>& acc, const AggParams& p, int64_t chunk,
                                              int f0, bool active) {
    float* base = p.part + chunk * (int64_t)p.n_slots * p.F;
    if (active) {
        stv<VEC>(base + (int64_t)SLOT_SUM * p.F + f0, acc.sum);
        if constexpr (STATS) {
            stv<VEC>(base + (int64_t)SLOT_SQ * p.F + f0, acc.sq);
            stv<VEC>(base + (int64_t)SLOT_MAX * p.F + f0, acc.mx);
            stv<VEC>(base + (int64_t)SLOT_MIN * p.F + f0, acc.mn);
            if constexpr (TRACK) {
                stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)SLOT_AMAX * p.F + f0), acc.amax);
                stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)SLOT_AMIN * p.F + f0), acc.amin);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < p.n_ch) {
                stv<VEC>(base + (int64_t)(SLOT_W0 + 2 * c) * p.F + f0, acc.ws[c]);
                stv<VEC>(base + (int64_t)(SLOT_W0 + 2 * c + 1) * p.F + f0, acc.wa[c]);
            }
        }
    }
    if (lane_id() == 0 && blockIdx.y == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) p.part_sw[chunk * DGN_MAX_CH + c] = acc.sw[c];
    }
}

template <int VEC, int NCH, bool STATS, bool TRACK>
__device__ __forceinline__ void load_partial(Acc<VEC, NCH, STATS, TRACK>& acc, const AggParams& p, int64_t chunk, int f0,
                                             bool active) {
    const float* base = p.part + chunk * (int64_t)p.n_slots * p.F;
    acc.init();
    if (active) {
        ldv<VEC>(acc.sum, base + (int64_t)SLOT_SUM * p.F + f0);
        if constexpr (STATS) {
            ldv<VEC>(acc.sq, base + (int64_t)SLOT_SQ * p.F + f0);
            ldv<VEC>(acc.mx, base + (int64_t)SLOT_MAX * p.F + f0);
            ldv<VEC>(acc.mn, base + (int64_t)SLOT_MIN * p.F + f0);
            if constexpr (TRACK) {
                ldvi<VEC>(acc.amax, reinterpret_cast<const int*>(base + (int64_t)SLOT_AMAX * p.F + f0));
                ldvi<VEC>(acc.amin, reinterpret_cast<const int*>(base + (int64_t)SLOT_AMIN * p.F + f0));
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < p.n_ch) {
                ldv<VEC>(acc.ws[c], base + (int64_t)(SLOT_W0 + 2 * c) * p.F + f0);
                ldv<VEC>(acc.wa[c], base + (int64_t)(SLOT_W0 + 2 * c + 1) * p.F + f0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc.sw[c] = p.part_sw[chunk * DGN_MAX_CH + c];
}

// ---- forward kernels --------------------------------------------------------------------------

template <int VEC, int NCH, bool STATS>
__global__ __launch_bounds__(kBlock) void agg_fwd_rows(const AggParams p) {
    const int64_t n_blocks = (p.n_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t lb = xcd_remap(blockIdx.x, n_blocks);
    if (lb < 0) return;
    const int64_t row64 = lb * kWavesPerBlock + (threadIdx.x >> 6);
    if (row64 >= p.n_nodes) return;
    const int row = uniform_i((int)row64);
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    const int deg = end - beg;
    if (deg > p.hub_threshold) return;  // hub row: slice + combine kernels own it
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<VEC, NCH, STATS, false> acc;
    acc.init();
    float xd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xd[i] = 0.f;
    if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
    accumulate_range<VEC, NCH, STATS, false>(acc, p, beg, end, f0, active, xd);
    if (active) write_row<VEC, NCH, STATS>(acc, p, row, deg, f0);
}

// one wave per hub slice: partial accumulators -> workspace
template <int VEC, int NCH, bool STATS, bool TRACK>
__global__ __launch_bounds__(kBlock) void agg_hub_slices(const AggParams p) {
    const int64_t chunk64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk64 >= p.n_chunks) return;
    const int chunk = uniform_i((int)chunk64);
    const int hub = p.chunk_hub[chunk];
    const int row = p.hub_rows[hub];
    const int rbeg = p.indptr[row], rend = p.indptr[row + 1];
    const int beg = rbeg + (chunk - p.hub_chunk_ptr[hub]) * p.hub_chunk;
    const int end = min(beg + p.hub_chunk, rend);
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<VEC, NCH, STATS, TRACK> acc;
    acc.init();
    float xd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xd[i] = 0.f;
    if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
    accumulate_range<VEC, NCH, STATS, TRACK>(acc, p, beg, end, f0, active, xd);
    store_partial<VEC, NCH, STATS, TRACK>(acc, p, chunk, f0, active);
}

// one wave per hub row: merge its slices in slot order, then the normal epilogue
template <int VEC, int NCH, bool STATS>
__global__ __launch_bounds__(kBlock) void agg_fwd_hub_combine(const AggParams p) {
    const int64_t hub64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (hub64 >= p.n_hub) return;
    const int hub = uniform_i((int)hub64);
    const int row = p.hub_rows[hub];
    const int deg = p.indptr[row + 1] - p.indptr[row];
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<VEC, NCH, STATS, false> acc, part;
    acc.init();
    for (int c = p.hub_chunk_ptr[hub]; c < p.hub_chunk_ptr[hub + 1]; ++c) {
        load_partial<VEC, NCH, STATS, false>(part, p, c, f0, active);
        acc.merge(part, p);
    }
    if (active) write_row<VEC, NCH, STATS>(acc, p, row, deg, f0);
}

// ---- backward -----------------------------------------------------------------------------------

template <int VEC, int NCH>
struct Coef {
    float c0[VEC], cv[VEC], gmax[VEC], gmin[VEC];
    int amax[VEC], amin[VEC];
    float cs[NCH][VEC], ca[NCH][VEC];
};

// per-row coefficient vectors from the upstream gradient and the (recomputed) accumulators;
// also returns d x_in for this row.
template <int VEC, int NCH, bool STATS>
__device__ __forceinline__ void make_coef(Coef<VEC, NCH>& k, float (&gxin)[VEC], const Acc<VEC, NCH, STATS, true>& acc,
                                          const AggParams& p, int row, int deg, int f0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        k.c0[i] = 0.f; k.cv[i] = 0.f; k.gmax[i] = 0.f; k.gmin[i] = 0.f; gxin[i] = 0.f;
        k.amax[i] = -1; k.amin[i] = -1;
    }
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { k.amax[i] = acc.amax[i]; k.amin[i] = acc.amin[i]; }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { k.cs[c][i] = 0.f; k.ca[c][i] = 0.f; }
    }
    const float d = (float)deg;
    const float logd = p.log_deg ? p.log_deg[row] : 0.f;
    float fac[DGN_MAX_SCALERS];
#pragma unroll
    for (int s = 0; s < DGN_MAX_SCALERS; ++s) fac[s] = s < p.n_scalers ? scaler_factor(p.scaler[s], logd, p.avg_log) : 0.f;
    float xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xin[i] = 0.f;
    if (p.need & NEED_XIN) ldv<VEC>(xin, p.x_in + (int64_t)row * p.ld_in + f0);
    const float* grow = p.g_out + (int64_t)row * p.ld_gout;
    for (int a = 0; a < p.n_agg; ++a) {
        float g[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] = 0.f;
#pragma unroll
        for (int s = 0; s < DGN_MAX_SCALERS; ++s) {
            if (s < p.n_scalers) {
                float t[VEC];
                ldv<VEC>(t, grow + out_col(p, s, a, f0));
#pragma unroll
                for (int i = 0; i < VEC; ++i) g[i] += p.scaler[s] == DGN_SCALE_IDENTITY ? t[i] : t[i] * fac[s];
            }
        }
        const int op = p.op[a];
        const int c = p.ch[a];
        float wsv[VEC], swv = 0.f, dcs[VEC], dca[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { wsv[i] = 0.f; dcs[i] = 0.f; dca[i] = 0.f; }
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            if (cc == c) {
                swv = acc.sw[cc];
#pragma unroll
                for (int i = 0; i < VEC; ++i) wsv[i] = acc.ws[cc][i];
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            switch (op) {
                case DGN_AGG_MEAN: k.c0[i] += g[i] / d; break;
                case DGN_AGG_SUM: k.c0[i] += g[i]; break;
                case DGN_AGG_MAX: k.gmax[i] += g[i]; break;
                case DGN_AGG_MIN: k.gmin[i] += g[i]; break;
                case DGN_AGG_VAR:
                case DGN_AGG_STD:
                    if constexpr (STATS) {
                        const float rv = raw_var(acc.sq[i], acc.sum[i], d);
                        if (rv > 0.f) {  // relu'(x) = [x > 0]
                            float gg = g[i];
                            if (op == DGN_AGG_STD) gg = gg / (2.f * __fsqrt_rn(rv + p.eps));
                            const float t = gg * 2.f / d;
                            k.cv[i] += t;
                            k.c0[i] -= t * (acc.sum[i] / d);
                        }
                    }
                    break;
                case DGN_AGG_DIR_AV: dca[i] = g[i]; break;
                case DGN_AGG_DIR_WSUM: dcs[i] = g[i]; break;
                case DGN_AGG_DIR_DX_NO_ABS:
                    dcs[i] = g[i];
                    gxin[i] -= swv * g[i];
                    break;
                case DGN_AGG_DIR_DX: {
                    const float r = dx_residual(wsv[i], swv, xin[i]);
                    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);  // d|r|/dr with sign(0) = 0
                    dcs[i] = sg * g[i];
                    gxin[i] -= sg * swv * g[i];
                    break;
                }
            }
        }
        if (op >= DGN_AGG_DIR_AV) {
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                if (cc == c) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { k.cs[cc][i] += dcs[i]; k.ca[cc][i] += dca[i]; }
                }
            }
        }
    }
}

// emit dm_j for slots [beg, end) of a row; returns the row-sum of dm_j in rsum
template <int VEC, int NCH>
__device__ __forceinline__ void emit_range(const Coef<VEC, NCH>& k, float (&rsum)[VEC], const AggParams& p, int beg,
                                           int end, int f0, bool active, const float (&xd)[VEC]) {
    const int lane = lane_id();
    const bool need_m = (p.need & NEED_M_EMIT) != 0;
    for (int base = beg; base < end; base += kWave) {
        const int e_l = base + lane;
        const bool in = e_l < end;
        const int my_src = in ? p.src[e_l] : 0;
        float my_w[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) my_w[c] = (in && c < p.n_ch) ? p.w[(int64_t)c * p.ld_w + e_l] : 0.f;
        const int cnt = min(kWave, end - base);
        for (int kk = 0; kk < cnt; ++kk) {
            const int s = bcast_i(my_src, kk);
            float wk[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) wk[c] = bcast_f(my_w[c], kk);
            if (!active) continue;
            const int pos = base + kk;
            float gm[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) gm[i] = k.c0[i];
            if (need_m) {
                float m[VEC];
                load_msg<VEC>(m, p, s, pos, f0, xd);
#pragma unroll
                for (int i = 0; i < VEC; ++i) gm[i] = fmaf(k.cv[i], m[i], gm[i]);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c < p.n_ch) {
                    const float a = fabsf(wk[c]);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gm[i] = fmaf(a, k.ca[c][i], fmaf(wk[c], k.cs[c][i], gm[i]));
                }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if (k.amax[i] == pos) gm[i] += k.gmax[i];
                if (k.amin[i] == pos) gm[i] += k.gmin[i];
                rsum[i] += gm[i];
            }
            if (p.g_src) {
                float* dst = p.g_src + (int64_t)s * p.ldg_src + f0;
#pragma unroll
                for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, gm[i]);
            }
            if (p.g_edge) stv<VEC>(p.g_edge + (int64_t)pos * p.ldg_edge + f0, gm);
        }
    }
}

template <int VEC>
__device__ __forceinline__ void add_row_grads(const AggParams& p, int row, int f0, const float (&rsum)[VEC],
                                              const float (&gxin)[VEC], bool with_xin) {
    if (p.g_dst) {
        float* dst = p.g_dst + (int64_t)row * p.ldg_dst + f0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, rsum[i]);
    }
    if (with_xin && p.g_in && (p.need & NEED_XIN)) {
        float* dst = p.g_in + (int64_t)row * p.ldg_in + f0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, gxin[i]);
    }
}

template <int VEC, int NCH, bool STATS>
__global__ __launch_bounds__(kBlock) void agg_bwd_rows(const AggParams p) {
    const int64_t n_blocks = (p.n_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t lb = xcd_remap(blockIdx.x, n_blocks);
    if (lb < 0) return;
    const int64_t row64 = lb * kWavesPerBlock + (threadIdx.x >> 6);
    if (row64 >= p.n_nodes) return;
    const int row = uniform_i((int)row64);
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    const int deg = end - beg;
    if (deg == 0 || deg > p.hub_threshold) return;
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    float xd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xd[i] = 0.f;
    if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
    Acc<VEC, NCH, STATS, true> acc;
    acc.init();
    if (p.need & NEED_RECOMP) {
        accumulate_range<VEC, NCH, STATS, true>(acc, p, beg, end, f0, active, xd);
    } else if (p.n_ch > 0) {
        // only sum_j w_jc is needed (d x_in of dx-no-abs): weights alone, no gathers
        const int lane = lane_id();
        float part[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) part[c] = 0.f;
        for (int e = beg + lane; e < end; e += kWave) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (c < p.n_ch) part[c] += p.w[(int64_t)c * p.ld_w + e];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc.sw[c] = wave_sum(part[c]);
    }
    Coef<VEC, NCH> k;
    float gxin[VEC], rsum[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) rsum[i] = 0.f;
    if (active) make_coef<VEC, NCH, STATS>(k, gxin, acc, p, row, deg, f0);
    emit_range<VEC, NCH>(k, rsum, p, beg, end, f0, active, xd);
    if (active) add_row_grads<VEC>(p, row, f0, rsum, gxin, true);
}

// hub backward, phase 2: merge slice partials, build the row's coefficient vectors, park them
template <int VEC, int NCH, bool STATS>
__global__ __launch_bounds__(kBlock) void agg_bwd_hub_coef(const AggParams p) {
    const int64_t hub64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (hub64 >= p.n_hub) return;
    const int hub = uniform_i((int)hub64);
    const int row = p.hub_rows[hub];
    const int deg = p.indptr[row + 1] - p.indptr[row];
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<VEC, NCH, STATS, true> acc, part;
    acc.init();
    for (int c = p.hub_chunk_ptr[hub]; c < p.hub_chunk_ptr[hub + 1]; ++c) {
        load_partial<VEC, NCH, STATS, true>(part, p, c, f0, active);
        acc.merge(part, p);
    }
    if (!active) return;
    Coef<VEC, NCH> k;
    float gxin[VEC], zero[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) zero[i] = 0.f;
    make_coef<VEC, NCH, STATS>(k, gxin, acc, p, row, deg, f0);
    float* base = p.coef + (int64_t)hub * p.n_coef * p.F;
    stv<VEC>(base + (int64_t)COEF_C0 * p.F + f0, k.c0);
    stv<VEC>(base + (int64_t)COEF_CV * p.F + f0, k.cv);
    stv<VEC>(base + (int64_t)COEF_GMAX * p.F + f0, k.gmax);
    stv<VEC>(base + (int64_t)COEF_GMIN * p.F + f0, k.gmin);
    stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)COEF_AMAX * p.F + f0), k.amax);
    stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)COEF_AMIN * p.F + f0), k.amin);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c < p.n_ch) {
            stv<VEC>(base + (int64_t)(COEF_W0 + 2 * c) * p.F + f0, k.cs[c]);
            stv<VEC>(base + (int64_t)(COEF_W0 + 2 * c + 1) * p.F + f0, k.ca[c]);
        }
    }
    add_row_grads<VEC>(p, row, f0, zero, gxin, true);  // d x_in only (rsum = 0)
}

// hub backward, phase 3: one wave per slice emits with the parked coefficients
template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void agg_bwd_hub_emit(const AggParams p) {
    const int64_t chunk64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk64 >= p.n_chunks) return;
    const int chunk = uniform_i((int)chunk64);
    const int hub = p.chunk_hub[chunk];
    const int row = p.hub_rows[hub];
    const int rbeg = p.indptr[row], rend = p.indptr[row + 1];
    const int beg = rbeg + (chunk - p.hub_chunk_ptr[hub]) * p.hub_chunk;
    const int end = min(beg + p.hub_chunk, rend);
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    float xd[VEC], rsum[VEC], gxin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xd[i] = 0.f; rsum[i] = 0.f; gxin[i] = 0.f; }
    Coef<VEC, NCH> k;
    if (active) {
        if (p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
        const float* base = p.coef + (int64_t)hub * p.n_coef * p.F;
        ldv<VEC>(k.c0, base + (int64_t)COEF_C0 * p.F + f0);
        ldv<VEC>(k.cv, base + (int64_t)COEF_CV * p.F + f0);
        ldv<VEC>(k.gmax, base + (int64_t)COEF_GMAX * p.F + f0);
        ldv<VEC>(k.gmin, base + (int64_t)COEF_GMIN * p.F + f0);
        ldvi<VEC>(k.amax, reinterpret_cast<const int*>(base + (int64_t)COEF_AMAX * p.F + f0));
        ldvi<VEC>(k.amin, reinterpret_cast<const int*>(base + (int64_t)COEF_AMIN * p.F + f0));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { k.cs[c][i] = 0.f; k.ca[c][i] = 0.f; }
            if (c < p.n_ch) {
                ldv<VEC>(k.cs[c], base + (int64_t)(COEF_W0 + 2 * c) * p.F + f0);
                ldv<VEC>(k.ca[c], base + (int64_t)(COEF_W0 + 2 * c + 1) * p.F + f0);
            }
        }
    }
    emit_range<VEC, NCH>(k, rsum, p, beg, end, f0, active, xd);
    if (active) add_row_grads<VEC>(p, row, f0, rsum, gxin, false);
}

// ---- host side --------------------------------------------------------------------------------

bool aligned(const void* ptr, int bytes) { return (reinterpret_cast<uintptr_t>(ptr) % bytes) == 0; }

int pick_vec(const DgnAggSpec* spec, const DgnMsg* msg, const float* out, int64_t ld_out, const DgnMsgGrad* gr) {
    const int64_t Ft = msg->F / spec->n_towers;
    for (int vec : {4, 2}) {
        bool ok = (msg->F % vec == 0) && (Ft % vec == 0) && (ld_out % vec == 0) && aligned(out, 4 * vec);
        // keep more than half of the 64 lanes busy, unless the row is too narrow anyway
        if (msg->F / vec <= 32 && vec > 2) ok = false;
        auto chk = [&](const float* ptr, int64_t ld) {
            if (ptr && (!aligned(ptr, 4 * vec) || ld % vec != 0)) ok = false;
        };
        chk(msg->x_src, msg->ld_src);
        chk(msg->x_dst, msg->ld_dst);
        chk(msg->m_edge, msg->ld_edge);
        chk(msg->x_in, msg->ld_in);
        if (gr) {
            chk(gr->g_src, gr->ld_src);
            chk(gr->g_dst, gr->ld_dst);
            chk(gr->g_edge, gr->ld_edge);
            chk(gr->g_in, gr->ld_in);
        }
        if (ok) return vec;
    }
    return 1;
}

int n_slots_for(const DgnAggSpec* spec) { return SLOT_W0 + 2 * spec->n_ch; }
int n_coef_for(const DgnAggSpec* spec) { return COEF_W0 + 2 * spec->n_ch; }

int validate(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, const float* log_deg) {
    if (!g || !spec || !msg) { set_error("null graph/spec/msg"); return DGN_ERR_INVALID; }
    if (g->n_nodes < 0 || g->n_edges < 0 || g->n_nodes > INT32_MAX - 1 || g->n_edges > INT32_MAX - 1) {
        set_error("n_nodes/n_edges out of the int32 CSR range"); return DGN_ERR_INVALID;
    }
    if (g->n_nodes > 0 && (!g->indptr || (g->n_edges > 0 && !g->src))) { set_error("null CSR arrays"); return DGN_ERR_INVALID; }
    if (spec->n_agg < 1 || spec->n_agg > DGN_MAX_AGG) { set_error("n_agg=%d outside 1..%d", spec->n_agg, DGN_MAX_AGG); return DGN_ERR_INVALID; }
    if (spec->agg_total != 0 && (spec->agg_offset < 0 || spec->agg_offset + spec->n_agg > spec->agg_total)) { set_error("aggregator slice [%d, %d) outside agg_total=%d", spec->agg_offset, spec->agg_offset + spec->n_agg, spec->agg_total); return DGN_ERR_INVALID; }
    if (spec->n_ch < 0 || spec->n_ch > DGN_MAX_CH) { set_error("n_ch=%d outside 0..%d", spec->n_ch, DGN_MAX_CH); return DGN_ERR_INVALID; }
    if (spec->n_scalers < 1 || spec->n_scalers > DGN_MAX_SCALERS) { set_error("n_scalers=%d outside 1..%d", spec->n_scalers, DGN_MAX_SCALERS); return DGN_ERR_INVALID; }
    if (spec->n_towers < 1 || msg->F < 1 || msg->F % spec->n_towers != 0) { set_error("F=%lld not divisible by n_towers=%d", (long long)msg->F, spec->n_towers); return DGN_ERR_INVALID; }
    if (!msg->x_src && !msg->x_dst && !msg->m_edge) { set_error("message has no term"); return DGN_ERR_INVALID; }
    bool need_scale = false;
    for (int s = 0; s < spec->n_scalers; ++s) {
        if (spec->scaler[s] < DGN_SCALE_IDENTITY || spec->scaler[s] > DGN_SCALE_ATTENUATION) { set_error("unknown scaler %d", spec->scaler[s]); return DGN_ERR_INVALID; }
        need_scale |= spec->scaler[s] != DGN_SCALE_IDENTITY;
    }
    if (need_scale && !log_deg) { set_error("scalers need log_deg"); return DGN_ERR_INVALID; }
    for (int a = 0; a < spec->n_agg; ++a) {
        const int op = spec->agg_op[a];
        if (op < DGN_AGG_MEAN || op > DGN_AGG_DIR_DX_NO_ABS) { set_error("unknown aggregator op %d", op); return DGN_ERR_INVALID; }
        if (op >= DGN_AGG_DIR_AV) {
            if (spec->agg_ch[a] < 0 || spec->agg_ch[a] >= spec->n_ch) { set_error("aggregator %d: channel %d outside 0..%d", a, spec->agg_ch[a], spec->n_ch - 1); return DGN_ERR_INVALID; }
            if (!w) { set_error("directional aggregators need edge weights"); return DGN_ERR_INVALID; }
            if ((op == DGN_AGG_DIR_DX || op == DGN_AGG_DIR_DX_NO_ABS) && !msg->x_in) { set_error("dx aggregators need x_in"); return DGN_ERR_INVALID; }
        }
    }
    if (g->n_hub > 0 && (!g->hub_rows || !g->hub_chunk_ptr || !g->chunk_hub || g->hub_chunk < 1 || g->n_chunks < g->n_hub)) {
        set_error("inconsistent hub description"); return DGN_ERR_INVALID;
    }
    return DGN_OK;
}

void fill_params(AggParams& p, const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                 const float* log_deg) {
    p = AggParams{};
    p.indptr = g->indptr; p.src = g->src; p.n_nodes = g->n_nodes;
    p.n_hub = g->n_hub; p.n_chunks = g->n_hub > 0 ? g->n_chunks : 0;
    p.hub_threshold = g->n_hub > 0 ? g->hub_threshold : INT32_MAX;
    p.hub_chunk = g->hub_chunk; p.hub_rows = g->hub_rows; p.hub_chunk_ptr = g->hub_chunk_ptr; p.chunk_hub = g->chunk_hub;
    p.F = (int32_t)msg->F; p.Ft = (int32_t)(msg->F / spec->n_towers);
    p.x_src = msg->x_src; p.ld_src = msg->ld_src;
    p.x_dst = msg->x_dst; p.ld_dst = msg->ld_dst;
    p.m_edge = msg->m_edge; p.ld_edge = msg->ld_edge;
    p.x_in = msg->x_in; p.ld_in = msg->ld_in;
    p.w = w; p.ld_w = ld_w; p.log_deg = log_deg;
    p.n_agg = spec->n_agg; p.agg_total = spec->agg_total > 0 ? spec->agg_total : spec->n_agg;
    p.agg_offset = spec->agg_total > 0 ? spec->agg_offset : 0;
    p.n_ch = spec->n_ch; p.n_scalers = spec->n_scalers; p.n_towers = spec->n_towers;
    p.avg_log = spec->avg_log; p.eps = spec->eps;
    uint32_t need = NEED_SUM;
    for (int a = 0; a < spec->n_agg; ++a) {
        const int op = spec->agg_op[a], c = spec->agg_ch[a];
        p.op[a] = (int8_t)op; p.ch[a] = (int8_t)(op >= DGN_AGG_DIR_AV ? c : 0);
        switch (op) {
            case DGN_AGG_MAX: need |= NEED_MAX | NEED_RECOMP; break;
            case DGN_AGG_MIN: need |= NEED_MIN | NEED_RECOMP; break;
            case DGN_AGG_STD: case DGN_AGG_VAR: need |= NEED_SQ | NEED_RECOMP | NEED_M_EMIT; break;
            case DGN_AGG_DIR_AV: p.ch_abs |= (uint8_t)(1u << c); break;
            case DGN_AGG_DIR_WSUM: p.ch_signed |= (uint8_t)(1u << c); break;
            case DGN_AGG_DIR_DX: p.ch_signed |= (uint8_t)(1u << c); need |= NEED_XIN | NEED_RECOMP; break;
            case DGN_AGG_DIR_DX_NO_ABS: p.ch_signed |= (uint8_t)(1u << c); need |= NEED_XIN; break;
        }
    }
    for (int s = 0; s < spec->n_scalers; ++s) p.scaler[s] = (int8_t)spec->scaler[s];
    p.need = need;
    p.n_slots = n_slots_for(spec);
    p.n_coef = n_coef_for(spec);
}

bool wants_stats(const AggParams& p) { return (p.need & (NEED_SQ | NEED_MAX | NEED_MIN)) != 0; }

size_t hub_ws_bytes(const DgnGraph* g, const DgnAggSpec* spec, int64_t F) {
    if (!g || g->n_hub <= 0) return 0;
    size_t part = (size_t)g->n_chunks * n_slots_for(spec) * F * sizeof(float);
    size_t sw = (size_t)g->n_chunks * DGN_MAX_CH * sizeof(float);
    size_t coef = (size_t)g->n_hub * n_coef_for(spec) * F * sizeof(float);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return up(part) + up(sw) + up(coef);
}

void carve_ws(AggParams& p, const DgnGraph* g, const DgnAggSpec* spec, void* ws) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char* base = static_cast<char*>(ws);
    size_t part = (size_t)g->n_chunks * n_slots_for(spec) * p.F * sizeof(float);
    size_t sw = (size_t)g->n_chunks * DGN_MAX_CH * sizeof(float);
    p.part = reinterpret_cast<float*>(base);
    p.part_sw = reinterpret_cast<float*>(base + up(part));
    p.coef = reinterpret_cast<float*>(base + up(part) + up(sw));
}

template <typename Fn>
int dispatch(int vec, int n_ch, bool stats, Fn&& fn) {
    // NCH template buckets: 1, 2, 4
    const int nb = n_ch <= 1 ? 1 : (n_ch <= 2 ? 2 : 4);
#define DGN_CASE(V, N, S) if (vec == V && nb == N && stats == S) return fn(std::integral_constant<int, V>{}, std::integral_constant<int, N>{}, std::integral_constant<bool, S>{});
    DGN_CASE(1, 1, false) DGN_CASE(1, 1, true) DGN_CASE(1, 2, false) DGN_CASE(1, 2, true) DGN_CASE(1, 4, false) DGN_CASE(1, 4, true)
    DGN_CASE(2, 1, false) DGN_CASE(2, 1, true) DGN_CASE(2, 2, false) DGN_CASE(2, 2, true) DGN_CASE(2, 4, false) DGN_CASE(2, 4, true)
    DGN_CASE(4, 1, false) DGN_CASE(4, 1, true) DGN_CASE(4, 2, false) DGN_CASE(4, 2, true) DGN_CASE(4, 4, false) DGN_CASE(4, 4, true)
#undef DGN_CASE
    set_error("no kernel for vec=%d n_ch=%d", vec, n_ch);
    return DGN_ERR_INVALID;
}

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" size_t dgn_agg_workspace_bytes(const DgnGraph* g, const DgnAggSpec* spec, int64_t F) {
    if (!g || !spec) return 0;
    return hub_ws_bytes(g, spec, F);
}

extern "C" int dgn_agg_forward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                               const float* log_deg, float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream_) {
    int rc = validate(g, spec, msg, w, log_deg);
    if (rc) return rc;
    if (g->n_nodes == 0) return DGN_OK;
    if (!out || ld_out < (int64_t)spec->n_scalers * (spec->agg_total > 0 ? spec->agg_total : spec->n_agg) * msg->F) { set_error("out is null or ld_out too small"); return DGN_ERR_INVALID; }
    if (g->n_hub > 0 && (!ws || ws_bytes < hub_ws_bytes(g, spec, msg->F))) { set_error("workspace too small: need %zu bytes", hub_ws_bytes(g, spec, msg->F)); return DGN_ERR_WORKSPACE; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AggParams p;
    fill_params(p, g, spec, msg, w, ld_w, log_deg);
    p.out = out; p.ld_out = ld_out;
    if (g->n_hub > 0) carve_ws(p, g, spec, ws);
    const int vec = pick_vec(spec, msg, out, ld_out, nullptr);
    const unsigned tiles = (unsigned)((msg->F + kWave * vec - 1) / (kWave * vec));
    return dispatch(vec, spec->n_ch, wants_stats(p), [&](auto V, auto N, auto S) -> int {
        constexpr int VEC = decltype(V)::value; constexpr int NCH = decltype(N)::value; constexpr bool STATS = decltype(S)::value;
        const int64_t n_blocks = (p.n_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
        dim3 grid((unsigned)xcd_grid(n_blocks), tiles);
        hipLaunchKernelGGL((agg_fwd_rows<VEC, NCH, STATS>), grid, dim3(kBlock), 0, stream, p);
        if (p.n_hub > 0) {
            dim3 gs((unsigned)((p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
            hipLaunchKernelGGL((agg_hub_slices<VEC, NCH, STATS, false>), gs, dim3(kBlock), 0, stream, p);
            dim3 gc((unsigned)((p.n_hub + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
            hipLaunchKernelGGL((agg_fwd_hub_combine<VEC, NCH, STATS>), gc, dim3(kBlock), 0, stream, p);
        }
        DGN_HIP_CHECK(hipGetLastError());
        return DGN_OK;
    });
}

extern "C" int dgn_agg_backward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                                const float* log_deg, const float* g_out, int64_t ld_gout, const DgnMsgGrad* grads,
                                void* ws, size_t ws_bytes, void* stream_) {
    int rc = validate(g, spec, msg, w, log_deg);
    if (rc) return rc;
    if (!grads) { set_error("null grads"); return DGN_ERR_INVALID; }
    if (g->n_nodes == 0) return DGN_OK;
    if (!g_out || ld_gout < (int64_t)spec->n_scalers * (spec->agg_total > 0 ? spec->agg_total : spec->n_agg) * msg->F) { set_error("g_out is null or ld_gout too small"); return DGN_ERR_INVALID; }
    if (g->n_hub > 0 && (!ws || ws_bytes < hub_ws_bytes(g, spec, msg->F))) { set_error("workspace too small: need %zu bytes", hub_ws_bytes(g, spec, msg->F)); return DGN_ERR_WORKSPACE; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AggParams p;
    fill_params(p, g, spec, msg, w, ld_w, log_deg);
    p.g_out = g_out; p.ld_gout = ld_gout;
    p.g_src = msg->x_src ? grads->g_src : nullptr; p.ldg_src = grads->ld_src;
    p.g_dst = msg->x_dst ? grads->g_dst : nullptr; p.ldg_dst = grads->ld_dst;
    p.g_edge = msg->m_edge ? grads->g_edge : nullptr; p.ldg_edge = grads->ld_edge;
    p.g_in = msg->x_in ? grads->g_in : nullptr; p.ldg_in = grads->ld_in;
    if (g->n_hub > 0) carve_ws(p, g, spec, ws);
    const int vec = pick_vec(spec, msg, g_out, ld_gout, grads);
    const unsigned tiles = (unsigned)((msg->F + kWave * vec - 1) / (kWave * vec));
    return dispatch(vec, spec->n_ch, wants_stats(p), [&](auto V, auto N, auto S) -> int {
        constexpr int VEC = decltype(V)::value; constexpr int NCH = decltype(N)::value; constexpr bool STATS = decltype(S)::value;
        const int64_t n_blocks = (p.n_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
        dim3 grid((unsigned)xcd_grid(n_blocks), tiles);
        hipLaunchKernelGGL((agg_bwd_rows<VEC, NCH, STATS>), grid, dim3(kBlock), 0, stream, p);
        if (p.n_hub > 0) {
            dim3 gs((unsigned)((p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
            dim3 gc((unsigned)((p.n_hub + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
            hipLaunchKernelGGL((agg_hub_slices<VEC, NCH, STATS, true>), gs, dim3(kBlock), 0, stream, p);
            hipLaunchKernelGGL((agg_bwd_hub_coef<VEC, NCH, STATS>), gc, dim3(kBlock), 0, stream, p);
            hipLaunchKernelGGL((agg_bwd_hub_emit<VEC, NCH>), gs, dim3(kBlock), 0, stream, p);
        }
        DGN_HIP_CHECK(hipGetLastError());
        return DGN_OK;
    });
}
