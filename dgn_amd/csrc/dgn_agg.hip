// Host side of the fused DGN aggregation (C ABI entry points dgn_agg_*): argument validation, the
// mapping of a DgnAggSpec onto a compile-time accumulator configuration, workspace carving, launches.
// Device code: dgn_agg_kernels.hpp (instantiated per vector width in dgn_agg_v{1,2,4}.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dgn_agg_kernels.hpp"

namespace dgn {
namespace {

bool aligned(const void* ptr, int bytes) { return (reinterpret_cast<uintptr_t>(ptr) % bytes) == 0; }

int pick_vec(const DgnAggSpec* spec, const DgnMsg* msg, const float* out, int64_t ld_out, const DgnMsgGrad* gr) {
    const int64_t Ft = msg->F / spec->n_towers;
    for (int vec : {4, 2}) {
        if (msg->f_valid && vec != 2) continue;      // (rows of an odd width: the 8-byte-lane kernels, whose last lane shifts its pair)
        bool ok = (msg->F % vec == 0) && (Ft % vec == 0) && (ld_out % vec == 0) && aligned(out, 4 * vec) && (spec->tower_stride % vec == 0);
        // keep more than half of the 64 lanes busy, unless the row is too narrow anyway
        if (msg->F / vec <= 32 && vec > 2) ok = false;
        auto chk = [&](const float* ptr, int64_t ld) {
            if (ptr && (!aligned(ptr, 4 * vec) || ld % vec != 0)) ok = false;
        };
        if (!msg->f_valid) chk(msg->x_src, msg->ld_src);      // (f_valid: 4-byte aligned rows at an odd stride)
        chk(msg->x_dst, msg->ld_dst);
        chk(msg->m_edge, msg->ld_edge);
        if (!msg->f_valid) chk(msg->x_in, msg->ld_in);
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
    if (g->n_src < 0 || g->n_src > INT32_MAX - 1) { set_error("n_src out of the int32 range"); return DGN_ERR_INVALID; }
    if (spec->n_agg < 1 || spec->n_agg > DGN_MAX_AGG) { set_error("n_agg=%d outside 1..%d", spec->n_agg, DGN_MAX_AGG); return DGN_ERR_INVALID; }
    if (spec->agg_total != 0 && (spec->agg_offset < 0 || spec->agg_offset + spec->n_agg > spec->agg_total)) { set_error("aggregator slice [%d, %d) outside agg_total=%d", spec->agg_offset, spec->agg_offset + spec->n_agg, spec->agg_total); return DGN_ERR_INVALID; }
    if (spec->n_ch < 0 || spec->n_ch > DGN_MAX_CH) { set_error("n_ch=%d outside 0..%d", spec->n_ch, DGN_MAX_CH); return DGN_ERR_INVALID; }
    if (spec->n_scalers < 1 || spec->n_scalers > DGN_MAX_SCALERS) { set_error("n_scalers=%d outside 1..%d", spec->n_scalers, DGN_MAX_SCALERS); return DGN_ERR_INVALID; }
    if (spec->n_towers < 1 || msg->F < 1 || msg->F % spec->n_towers != 0) { set_error("F=%lld not divisible by n_towers=%d", (long long)msg->F, spec->n_towers); return DGN_ERR_INVALID; }
    if (!msg->x_src && !msg->x_dst && !msg->m_edge) { set_error("message has no term"); return DGN_ERR_INVALID; }
    if (msg->f_valid != 0 && (msg->f_valid != msg->F - 1 || (msg->F & 1) || msg->F < 4 || !msg->x_src || msg->x_dst || msg->m_edge || spec->n_towers != 1 ||
                              msg->ld_src < msg->f_valid || (msg->x_in && msg->ld_in < msg->f_valid) || g->n_hub > 0 || !dgn_agg_f_valid_supported(spec))) {
        set_error("f_valid = %d: rows of an odd width need F = f_valid + 1 even, x_src (and x_in) alone, one tower, no hub rows, a list of dgn_agg_f_valid_supported()", msg->f_valid); return DGN_ERR_INVALID;
    }
    if (msg->edge_type) {
        if (!msg->m_edge || !msg->x_src) { set_error("edge_type needs the table (m_edge) and x_src"); return DGN_ERR_INVALID; }
        if (msg->n_edge_types < 1 || (int64_t)msg->n_edge_types * msg->F > DGN_MAX_EDGE_TABLE) { set_error("edge-type table of %d x %lld floats outside 1..%d", msg->n_edge_types, (long long)msg->F, DGN_MAX_EDGE_TABLE); return DGN_ERR_INVALID; }
    }
    if ((int64_t)spec->n_scalers * (spec->agg_total > 0 ? spec->agg_total : spec->n_agg) * msg->F > INT32_MAX) { set_error("output row wider than 2^31 columns"); return DGN_ERR_INVALID; }
    if (msg->ld_src > INT32_MAX || msg->ld_dst > INT32_MAX || msg->ld_edge > INT32_MAX || msg->ld_in > INT32_MAX) { set_error("row strides must fit in int32"); return DGN_ERR_INVALID; }
    bool need_scale = false;
    for (int s = 0; s < spec->n_scalers; ++s) {
        if (spec->scaler[s] < DGN_SCALE_IDENTITY || spec->scaler[s] > DGN_SCALE_ATTENUATION) { set_error("unknown scaler %d", spec->scaler[s]); return DGN_ERR_INVALID; }
        need_scale |= spec->scaler[s] != DGN_SCALE_IDENTITY;
    }
    if (need_scale && !log_deg) { set_error("scalers need log_deg"); return DGN_ERR_INVALID; }
    for (int a = 0; a < spec->n_agg; ++a) {
        const int op = spec->agg_op[a];
        if (op < DGN_AGG_MEAN || op > DGN_AGG_X_IN) { set_error("unknown aggregator op %d", op); return DGN_ERR_INVALID; }
        if (op == DGN_AGG_X_IN && !msg->x_in) { set_error("the x_in pass-through needs x_in"); return DGN_ERR_INVALID; }
        if (op == DGN_AGG_X_IN && !(spec->n_scalers == 1 && spec->scaler[0] == DGN_SCALE_IDENTITY)) { set_error("the x_in pass-through requires the single identity scaler (fold the scalers into posttrans)"); return DGN_ERR_INVALID; }
        if (op >= DGN_AGG_DIR_AV && op <= DGN_AGG_DIR_DX_NO_ABS) {
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
    p.indptr = g->indptr; p.src = g->src; p.n_nodes = g->n_nodes; p.n_edges = g->n_edges;
    p.n_src = g->n_src > 0 ? g->n_src : g->n_nodes;
    p.n_hub = g->n_hub; p.n_chunks = g->n_hub > 0 ? g->n_chunks : 0;
    p.hub_threshold = g->n_hub > 0 ? g->hub_threshold : INT32_MAX;
    p.hub_chunk = g->hub_chunk; p.hub_rows = g->hub_rows; p.hub_chunk_ptr = g->hub_chunk_ptr; p.chunk_hub = g->chunk_hub;
    p.F = (int32_t)msg->F; p.Ft = (int32_t)(msg->F / spec->n_towers); p.Fv = msg->f_valid ? msg->f_valid : (int32_t)msg->F;
    p.x_src = msg->x_src; p.ld_src = (int32_t)msg->ld_src;
    p.x_dst = msg->x_dst; p.ld_dst = (int32_t)msg->ld_dst;
    p.m_edge = msg->m_edge; p.ld_edge = (int32_t)msg->ld_edge;
    p.edge_type = msg->edge_type; p.n_edge_types = msg->edge_type ? msg->n_edge_types : 0;
    p.x_in = msg->x_in; p.ld_in = (int32_t)msg->ld_in;
    p.w = w; p.ld_w = (int32_t)ld_w; p.log_deg = log_deg;
    p.n_agg = spec->n_agg; p.agg_total = spec->agg_total > 0 ? spec->agg_total : spec->n_agg;
    p.agg_offset = spec->agg_total > 0 ? spec->agg_offset : 0;
    p.n_ch = spec->n_ch; p.n_scalers = spec->n_scalers; p.n_towers = spec->n_towers;
    p.tower_stride = spec->tower_stride > 0 ? spec->tower_stride : (int64_t)spec->n_scalers * p.agg_total * p.Ft;
    p.avg_log = spec->avg_log; p.eps = spec->eps;
    uint32_t need = 0;
    for (int a = 0; a < spec->n_agg; ++a) {
        const int op = spec->agg_op[a], c = spec->agg_ch[a];
        p.op_pack |= (uint64_t)op << (4 * a);
        p.ch_pack |= (uint64_t)((op >= DGN_AGG_DIR_AV && op <= DGN_AGG_DIR_DX_NO_ABS) ? c : 0) << (3 * a);
        switch (op) {
            case DGN_AGG_MAX: need |= NEED_MAX | NEED_RECOMP; break;
            case DGN_AGG_MIN: need |= NEED_MIN | NEED_RECOMP; break;
            case DGN_AGG_STD: case DGN_AGG_VAR: need |= NEED_SQ | NEED_RECOMP | NEED_M_EMIT; break;
            case DGN_AGG_DIR_AV: p.any_av = true; break;
            case DGN_AGG_DIR_WSUM: break;
            case DGN_AGG_DIR_DX: need |= NEED_XIN | NEED_RECOMP; break;
            case DGN_AGG_DIR_DX_NO_ABS: need |= NEED_XIN; break;
            case DGN_AGG_X_IN: need |= NEED_XIN | NEED_XPASS; break;
        }
    }
    for (int s = 0; s < spec->n_scalers; ++s) p.scaler_pack |= (uint32_t)spec->scaler[s] << (2 * s);
    p.need = need;
    p.n_slots = n_slots_for(spec);
    p.n_coef = n_coef_for(spec);
}

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

// ---- gradient of the edge-type table: rows of the staged per-edge gradients summed by type -------------------------------
// (the sweep parked d m_j of slot j at stage[csc_pos[j]]; d table[k] = sum of the rows whose slot has type k).  Two passes, fixed
// slot ranges per wave and a fixed summation order: deterministic.
constexpr int kTabBlocks = 512;
constexpr int kTabGroup = 8;         // staged rows in flight per wave

__global__ __launch_bounds__(1024) void edge_table_grad_partial(const float* __restrict__ stage, const int32_t* __restrict__ csc_pos,
                                                               const int32_t* __restrict__ edge_type, int64_t n_edges, int F, int K,
                                                               float* __restrict__ part) {
    extern __shared__ float tab_acc[];                        // [waves][K][F]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6, KF = K * F;
    float* mine = tab_acc + wave * KF;
    for (int i = lane; i < KF; i += kWave) mine[i] = 0.f;
    const int64_t nw = (int64_t)gridDim.x * waves, gw = (int64_t)blockIdx.x * waves + wave;
    const int64_t per = (n_edges + nw - 1) / nw, j0 = gw * per, j1 = min(n_edges, j0 + per);
    for (int64_t base = j0; base < j1; base += kWave) {
        const int cnt = (int)min((int64_t)kWave, j1 - base);
        const int my_pos = lane < cnt ? csc_pos[base + lane] : 0, my_k = lane < cnt ? edge_type[base + lane] : 0;
        for (int q = 0; q < cnt; q += kTabGroup) {
            for (int f = 2 * lane; f < F; f += 2 * kWave) {
                float2 v[kTabGroup];
#pragma unroll
                for (int u = 0; u < kTabGroup; ++u)
                    if (u == 0 || q + u < cnt) v[u] = *reinterpret_cast<const float2*>(stage + (int64_t)bcast_i(my_pos, q + u) * F + f);
#pragma unroll
                for (int u = 0; u < kTabGroup; ++u) {
                    if (u == 0 || q + u < cnt) {
                        float* a = mine + bcast_i(my_k, q + u) * F + f;      // (one wave, LDS operations in program order)
                        a[0] += v[u].x;
                        a[1] += v[u].y;
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KF; i += blockDim.x) {
        float a = 0.f;
        for (int w = 0; w < waves; ++w) a += tab_acc[w * KF + i];
        part[(int64_t)blockIdx.x * KF + i] = a;
    }
}

// 16 table entries per workgroup, 16 threads per entry: each adds every 16th partial (fixed order), then the 16 are added in order
__global__ __launch_bounds__(256) void edge_table_grad_final(const float* __restrict__ part, int blocks, int F, int K, float* __restrict__ g_tab, int64_t ld) {
    __shared__ float red[16][17];
    const int li = threadIdx.x & 15, lb = threadIdx.x >> 4, i = blockIdx.x * 16 + li, KF = K * F;
    float a = 0.f;
    if (i < KF)
        for (int b = lb; b < blocks; b += 16) a += part[(int64_t)b * KF + i];
    red[lb][li] = a;
    __syncthreads();
    if (lb == 0 && i < KF) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][li];
        const int k = i / F;
        g_tab[(int64_t)k * ld + (i - k * F)] = t;
    }
}

size_t edge_table_ws_bytes(int64_t F, int32_t K) { return (((size_t)kTabBlocks * K * F * sizeof(float)) + 255) & ~(size_t)255; }

int launch(int vec, const AggParams& p, unsigned tiles, hipStream_t stream, bool backward) {
    switch (vec) {
        case 4: return launch_agg_v4(p, tiles, stream, backward);
        case 2: return launch_agg_v2(p, tiles, stream, backward);
        default: return launch_agg_v1(p, tiles, stream, backward);
    }
}

}  // namespace

// shared with dgn_fused.hip
int agg_validate_and_fill(AggParams& p, const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                          const float* log_deg) {
    const int rc = validate(g, spec, msg, w, log_deg);
    if (rc) return rc;
    fill_params(p, g, spec, msg, w, ld_w, log_deg);
    return DGN_OK;
}
}  // namespace dgn

using namespace dgn;

extern "C" size_t dgn_agg_workspace_bytes(const DgnGraph* g, const DgnAggSpec* spec, int64_t F) {
    if (!g || !spec) return 0;
    return hub_ws_bytes(g, spec, F);
}

size_t stage_bytes(const DgnGraph* g, int64_t F) { return ((size_t)g->n_edges * F * sizeof(float) + 255) & ~(size_t)255; }

// The aux byte table (AggParams.aux) applies when the 4-rows-per-wave kernels run a baked-in list that recomputes (max / min / dx)
// but needs no message value in the emit pass (no std / var), with at most two weight channels: short-row graph, even width.
static bool agg_aux_supported(const AggParams& p, const DgnMsg* msg) {
    static const bool off = getenv("DGN_NO_AUX") != nullptr;
    if (off || !is_hot_list(p) || p.n_ch > 2 || !(p.need & NEED_RECOMP) || (p.need & (NEED_M_EMIT | NEED_SQ)) || p.n_nodes <= 0) return false;
    if (short_rows(p)) return msg->x_src && (msg->F % 2) == 0;
    // longer rows (row-per-wave kernels): the dx signs only -- lists without max / min, no hub rows
    static const bool no_rows = getenv("DGN_NO_AUX_ROWS") != nullptr;
    return !no_rows && p.n_ch >= 1 && !(p.need & (NEED_MAX | NEED_MIN)) && p.n_hub == 0;
}
// (row-major sign table of the row-per-wave kernels, see AggParams.aux_rows)
static bool agg_aux_is_rows(const AggParams& p) { return !short_rows(p); }

extern "C" int dgn_agg_f_valid_supported(const DgnAggSpec* spec) {
    // (the lists with Cfg::ODD kernels, dgn_agg_kernels.hpp: odd_width_list; one identity scaler, the whole list in one launch, one tower)
    static const bool no_hot = getenv("DGN_NO_HOT") != nullptr;
    if (!spec || no_hot || spec->n_towers != 1 || spec->n_scalers != 1 || spec->scaler[0] != DGN_SCALE_IDENTITY ||
        (spec->agg_total > 0 && (spec->agg_offset != 0 || spec->agg_total != spec->n_agg))) return 0;
    uint64_t ops = 0, chs = 0;
    for (int a = 0; a < spec->n_agg; ++a) {
        const int op = spec->agg_op[a];
        ops |= (uint64_t)op << (4 * a);
        chs |= (uint64_t)((op >= DGN_AGG_DIR_AV && op <= DGN_AGG_DIR_DX_NO_ABS) ? spec->agg_ch[a] : 0) << (3 * a);
    }
    return odd_width_list(spec->n_agg, ops, chs) && spec->n_ch == (spec->n_agg == 2 ? 1 : 2) ? 1 : 0;
}

extern "C" size_t dgn_agg_aux_bytes(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg) {
    if (!g || !spec || !msg || msg->F <= 0 || spec->n_agg < 1 || spec->n_agg > DGN_MAX_AGG || spec->n_towers < 1) return 0;
    AggParams p;
    fill_params(p, g, spec, msg, nullptr, 0, nullptr);
    return agg_aux_supported(p, msg) ? (((size_t)((g->n_nodes + 3) / 4) * 4 * msg->F + 255) & ~(size_t)255) : 0;     // (groups of four rows)
}

extern "C" size_t dgn_agg_edge_table_workspace_bytes(int64_t F, int32_t n_edge_types) {
    return (F > 0 && n_edge_types > 0) ? edge_table_ws_bytes(F, n_edge_types) : 0;
}

extern "C" size_t dgn_agg_backward_workspace_bytes(const DgnGraph* g, const DgnAggSpec* spec, int64_t F, int32_t deterministic) {
    if (!g || !spec) return 0;
    size_t n = hub_ws_bytes(g, spec, F);
    if (deterministic && g->csc_ptr && g->csc_pos && g->n_edges > 0) n += stage_bytes(g, F);
    return n;
}

extern "C" int dgn_agg_forward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                               const float* log_deg, float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream_) {
    return dgn_agg_forward_aux(g, spec, msg, w, ld_w, log_deg, out, ld_out, nullptr, ws, ws_bytes, stream_);
}

extern "C" int dgn_agg_forward_aux(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                                   const float* log_deg, float* out, int64_t ld_out, unsigned char* aux, void* ws, size_t ws_bytes,
                                   void* stream_) {
    int rc = validate(g, spec, msg, w, log_deg);
    if (rc) return rc;
    if (g->n_nodes == 0) return DGN_OK;
    const int64_t width = (int64_t)spec->n_scalers * (spec->agg_total > 0 ? spec->agg_total : spec->n_agg) *
                          (spec->tower_stride > 0 ? msg->F / spec->n_towers : msg->F);
    if (!out || ld_out < width) { set_error("out is null or ld_out too small"); return DGN_ERR_INVALID; }
    if (g->n_hub > 0 && (!ws || ws_bytes < hub_ws_bytes(g, spec, msg->F))) { set_error("workspace too small: need %zu bytes", hub_ws_bytes(g, spec, msg->F)); return DGN_ERR_WORKSPACE; }
    AggParams p;
    fill_params(p, g, spec, msg, w, ld_w, log_deg);
    if (ld_out > INT32_MAX) { set_error("ld_out must fit in int32"); return DGN_ERR_INVALID; }
    p.out = out; p.ld_out = (int32_t)ld_out;
    if (aux) {
        if (!agg_aux_supported(p, msg)) { set_error("dgn_agg_forward_aux: this launch has no aux table (dgn_agg_aux_bytes() == 0)"); return DGN_ERR_INVALID; }
        p.aux = aux;
        p.aux_rows = agg_aux_is_rows(p);
    }
    if (g->n_hub > 0) carve_ws(p, g, spec, ws);
    const int vec = pick_vec(spec, msg, out, ld_out, nullptr);
    const unsigned tiles = (unsigned)((msg->F + kWave * vec - 1) / (kWave * vec));
    return launch(vec, p, tiles, static_cast<hipStream_t>(stream_), false);
}

namespace dgn {
// Everything dgn_agg_backward does before its launches: validation, the parameter block, the staging buffer of the two-phase scatter,
// define / accumulate mode of the sinks (zero fills where needed).  Shared with the fused backward (dgn_fused.hip), whose upstream
// gradient is formed in LDS: `g_out` may then be NULL (`lds_gout`).  *tab_part_out: workspace of the edge-type table's gradient.
int agg_backward_prepare(AggParams& p, const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                         const float* log_deg, const float* g_out, int64_t ld_gout, bool lds_gout, const DgnMsgGrad* grads, void* ws,
                         size_t ws_bytes, void* stream_, float** tab_part_out, const unsigned char* aux) {
    int rc = validate(g, spec, msg, w, log_deg);
    if (rc) return rc;
    if (!grads) { set_error("null grads"); return DGN_ERR_INVALID; }
    const int64_t width = (int64_t)spec->n_scalers * (spec->agg_total > 0 ? spec->agg_total : spec->n_agg) *
                          (spec->tower_stride > 0 ? msg->F / spec->n_towers : msg->F);
    if (!lds_gout && (!g_out || ld_gout < width)) { set_error("g_out is null or ld_gout too small"); return DGN_ERR_INVALID; }
    if (g->n_hub > 0 && (!ws || ws_bytes < hub_ws_bytes(g, spec, msg->F))) { set_error("workspace too small: need %zu bytes", hub_ws_bytes(g, spec, msg->F)); return DGN_ERR_WORKSPACE; }
    fill_params(p, g, spec, msg, w, ld_w, log_deg);
    if (ld_gout > INT32_MAX || grads->ld_src > INT32_MAX || grads->ld_dst > INT32_MAX || grads->ld_edge > INT32_MAX || grads->ld_in > INT32_MAX) { set_error("strides must fit in int32"); return DGN_ERR_INVALID; }
    p.g_out = g_out; p.ld_gout = (int32_t)ld_gout;
    p.g_src = msg->x_src ? grads->g_src : nullptr; p.ldg_src = (int32_t)grads->ld_src;
    p.g_dst = msg->x_dst ? grads->g_dst : nullptr; p.ldg_dst = (int32_t)grads->ld_dst;
    p.g_edge = msg->m_edge ? grads->g_edge : nullptr; p.ldg_edge = (int32_t)grads->ld_edge;
    p.g_in = msg->x_in ? grads->g_in : nullptr; p.ldg_in = (int32_t)grads->ld_in;
    if (aux) {
        if (lds_gout || !agg_aux_supported(p, msg)) { set_error("dgn_agg_backward_aux: this launch has no aux table (dgn_agg_aux_bytes() == 0)"); return DGN_ERR_INVALID; }
        p.aux = const_cast<unsigned char*>(aux);
        p.aux_rows = agg_aux_is_rows(p);
    }
    if (g->n_hub > 0) carve_ws(p, g, spec, ws);
    // atomic-free scatter when the transposed view and the [E, F] staging buffer are available
    if (p.g_src && g->csc_ptr && g->csc_pos && g->n_edges > 0 && ws &&
        ws_bytes >= hub_ws_bytes(g, spec, msg->F) + stage_bytes(g, msg->F)) {
        p.stage = reinterpret_cast<float*>(static_cast<char*>(ws) + hub_ws_bytes(g, spec, msg->F));
        p.csc_ptr = g->csc_ptr; p.csc_pos = g->csc_pos;
    }
    float* tab_part = nullptr;
    if (msg->edge_type) {
        // the table's gradient is a reduction of the staged rows by type (after the sweep): the sweep itself writes no g_edge
        const size_t need = hub_ws_bytes(g, spec, msg->F) + stage_bytes(g, msg->F) + edge_table_ws_bytes(msg->F, msg->n_edge_types);
        if (!p.stage || ws_bytes < need) { set_error("edge-type table: the backward needs the two-phase scatter (g_src, g->csc_*) and a workspace of %zu bytes", need); return DGN_ERR_WORKSPACE; }
        if (msg->F % 2 != 0) { set_error("edge-type table: F must be even"); return DGN_ERR_INVALID; }
        tab_part = reinterpret_cast<float*>(static_cast<char*>(ws) + hub_ws_bytes(g, spec, msg->F) + stage_bytes(g, msg->F));
        p.g_edge = nullptr;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // accumulate == 0: the sinks arrive uninitialised.  With the two-phase scatter every row of every sink is
    // written by exactly one thread (no zero-fill, no read-modify-write); the atomic scatter needs zeroed sinks.
    p.fresh = grads->accumulate == 0 && p.stage != nullptr;
    p.seg_add = !p.fresh || p.g_in == p.g_src;
    if (grads->accumulate == 0 && !p.fresh) {
        auto zero = [&](float* ptr, int32_t ld, int64_t rows) -> int { return zero_rows_async(ptr, rows, msg->F, ld, stream); };
        if (p.g_src && zero(p.g_src, p.ldg_src, p.n_src)) return DGN_ERR_HIP;
        if (p.g_dst && zero(p.g_dst, p.ldg_dst, p.n_nodes)) return DGN_ERR_HIP;
        if (p.g_in && p.g_in != p.g_src && zero(p.g_in, p.ldg_in, p.n_nodes)) return DGN_ERR_HIP;
    }
    if (tab_part_out) *tab_part_out = tab_part;
    return DGN_OK;
}
}  // namespace dgn

extern "C" int dgn_agg_backward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                                const float* log_deg, const float* g_out, int64_t ld_gout, const DgnMsgGrad* grads,
                                void* ws, size_t ws_bytes, void* stream_) {
    return dgn_agg_backward_aux(g, spec, msg, w, ld_w, log_deg, g_out, ld_gout, nullptr, grads, ws, ws_bytes, stream_);
}

extern "C" int dgn_agg_backward_aux(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                                    const float* log_deg, const float* g_out, int64_t ld_gout, const unsigned char* aux,
                                    const DgnMsgGrad* grads, void* ws, size_t ws_bytes, void* stream_) {
    if (g && g->n_nodes == 0 && spec && msg && grads) return validate(g, spec, msg, w, log_deg);
    AggParams p;
    float* tab_part = nullptr;
    int rc = agg_backward_prepare(p, g, spec, msg, w, ld_w, log_deg, g_out, ld_gout, false, grads, ws, ws_bytes, stream_, &tab_part, aux);
    if (rc) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int vec = pick_vec(spec, msg, g_out, ld_gout, grads);
    const unsigned tiles = (unsigned)((msg->F + kWave * vec - 1) / (kWave * vec));
    // Block backward (dgn_agg_block.hpp): batches of small graphs, define mode, one feature tile, 8-byte lanes, messages without
    // an edge term, the same node set on both sides; lists without a block kernel (anything but the baked-in ones) come back with 1
    if (g->blk_cut && g->blk_gap > 0 && g->n_edges > 0 && grads->accumulate == 0 && g->n_hub == 0 && tiles == 1 && vec == 2 && !msg->edge_type &&
        !msg->m_edge && p.g_src && p.x_src && p.n_src == p.n_nodes && !p.g_edge) {
        p.blk_cut = g->blk_cut;
        const int brc = launch_agg_block_v2(p, g->blk_gap, stream);
        if (brc != 1) return brc;
        p.blk_cut = nullptr;
    }
    // Graph backward (dgn_agg_graph.hpp): graphs beyond a wave's LDS block (k-NN, SBM) -- a workgroup per graph, the destination rows'
    // coefficient vectors in LDS, every source row gathering its out-edges: lists without max / min / std / var, same conditions otherwise
    if (g->gblk_desc && g->n_gblk > 0 && g->csc_order && g->dst_csr && g->csc_ptr && g->n_edges > 0 && grads->accumulate == 0 && g->n_hub == 0 &&
        tiles == 1 && vec == 2 && !msg->edge_type && !msg->m_edge && p.g_src && p.x_src && p.n_src == p.n_nodes && !p.g_edge && !short_rows(p)) {
        p.gblk_desc = g->gblk_desc; p.n_gblk = (int32_t)g->n_gblk; p.gblk_rows = g->gblk_rows; p.csc_order = g->csc_order; p.dst_csr = g->dst_csr;
        p.csc_ptr = g->csc_ptr;
        const int grc = launch_agg_graph_v2(p, stream);
        if (grc != 1) return grc;
        p.gblk_desc = nullptr;
    }
    rc = launch(vec, p, tiles, stream, true);
    if (rc || !msg->edge_type || !grads->g_edge) return rc;
    const int K = msg->n_edge_types, KF = K * (int)msg->F;
    const int waves = std::max(1, std::min(16, 16384 / KF));                 // per-wave accumulator tables: at most 64 KB of LDS
    hipLaunchKernelGGL(edge_table_grad_partial, dim3(kTabBlocks), dim3(kWave * waves), (size_t)waves * KF * sizeof(float), stream,
                       p.stage, p.csc_pos, msg->edge_type, p.n_edges, (int)msg->F, K, tab_part);
    hipLaunchKernelGGL(edge_table_grad_final, dim3((KF + 15) / 16), dim3(256), 0, stream, tab_part, kTabBlocks, (int)msg->F, K,
                       grads->g_edge, grads->ld_edge);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
