// Whole-layer entry points of the towers DGN layer (include/dgn_hip.h: dgn_towers_layer_*): every kernel of one
// DGNLayerTower.forward -- P|Q pretrans Linear, the aggregation sweep, posttrans + scale-combine, BatchNorm tail, mixing
// Linear, bias + LeakyReLU + residual -- and of its backward is enqueued by ONE call, on the caller's stream, with no host
// synchronisation.  At the reference's batch size (128 molecules, configs/molecules_graph_regression_DGN_ZINC.json) the layer is
// host-bound when every kernel is its own Python / autograd node (~48 launches, 1.2-1.4 ms around 0.2 ms of GPU work); here the
// host side of a step is two calls.  No new device code: the calls below are the library's own entry points.
// Reference: realworld_benchmark/nets/dgn_layer.py:254-325 (DGNTower.forward x towers, mixing network, residual).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include <cstdint>

#include "dgn_common.hpp"

namespace dgn {
namespace {

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// g_h = [g_res] + g_a + g_b   (the three contributions to d h: residual, h_in of the sweep, the P|Q input gradient)
__global__ __launch_bounds__(256) void add3_rows(int64_t n4, const float4* __restrict__ res, const float4* __restrict__ a,
                                                 const float4* __restrict__ b, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = a[i];
    const float4 w = b[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    if (res) {
        const float4 r = res[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    out[i] = v;
}
__global__ __launch_bounds__(256) void add3_tail(int64_t first, int64_t n, const float* __restrict__ res, const float* __restrict__ a,
                                                 const float* __restrict__ b, float* __restrict__ out) {
    const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a[i] + b[i] + (res ? res[i] : 0.f);
}

// fused operand buffer of the layer from its per-tower parameters: element i comes from parameter map_param[i] at offset map_off[i]
// (or is zero: the off-diagonal blocks of the block-diagonal P|Q weights, the h columns of the non-identity scaler blocks)
__global__ __launch_bounds__(256) void assemble_params(int64_t n, const int64_t* __restrict__ ptrs, const int32_t* __restrict__ map_param,
                                                       const int32_t* __restrict__ map_off, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int q = map_param[i];
    out[i] = q < 0 ? 0.f : reinterpret_cast<const float*>(ptrs[q])[map_off[i]];
}

// Round 6: BatchNorm's backward column sums without a pass over the rows.  With z = y1 W_mix^T + b, y1 = gamma * xhat + beta:
//     g_y1 = g_z W_mix   =>   sum_m g_y1[m][c]            = sum_o (sum_m g_z[m][o]) W[o][c]             = sum_o db[o] W[o][c]
//                              sum_m g_y1[m][c] xhat[m][c] = sum_o (sum_m g_z[m][o] xhat[m][c]) W[o][c] = sum_o dWx[o][c] W[o][c]
// where dWx = g_z^T xhat and db = g_z^T 1 are what the mixing network's weight-gradient pass accumulates anyway when its operand is
// normalised WITHOUT the affine part (dgn_linear_wgrad_bn with gamma = beta = NULL); the weight gradient proper follows as
//     d W_mix[o][c] = sum_m g_z[m][o] y1[m][c] = gamma[c] dWx[o][c] + beta[c] db[o].
// A wave per column c, lane o (+ 64 q) a row of the weight, doubles for the two 70-term sums (added across the lanes in a fixed order):
// replaces bn_bwd_stats (37 us on ZINC-12k: a read of g_y1 and y0) + bn_bwd_finalize.  (A first version -- one thread per column walking
// the rows -- took 28 us: 70 dependent round trips.)  Reference: autograd through nn.BatchNorm1d at nets/dgn_layer.py:272-273.
__global__ __launch_bounds__(64) void mix_bn_finalize(int Fo, const float* __restrict__ dwx, const float* __restrict__ db, const float* __restrict__ w,
                                                      int64_t ldw, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ g_w, int64_t ldgw, float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                      float* __restrict__ sums) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    double s0 = 0.0, s1 = 0.0;
    for (int o = lane; o < Fo; o += 64) {
        const float x = dwx[(int64_t)o * Fo + c], b = db[o], wv = w[(int64_t)o * ldw + c];
        s0 += (double)b * (double)wv;
        s1 += (double)x * (double)wv;
        g_w[(int64_t)o * ldgw + c] = ga * x + be * b;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
    }
    if (lane == 0) {
        sums[c] = (float)s0; sums[Fo + c] = (float)s1;
        g_beta[c] = (float)s0; g_gamma[c] = (float)s1;
    }
}

struct Dims {
    int64_t N;
    int T, fi, fo, S, Fm, Fo, K;
};

bool dims_of(const DgnTowersLayer* L, Dims& d, const char* fn) {
    if (!L || !L->graph || !L->spec) { set_error("%s: null layer / graph / spec", fn); return false; }
    d.N = L->graph->n_nodes; d.T = L->n_towers; d.fi = L->f_in; d.fo = L->f_out; d.S = L->n_scalers;
    if (d.T < 1 || d.fi < 1 || d.fo < 1 || d.S < 1 || d.S > 3) { set_error("%s: bad tower / scaler counts", fn); return false; }
    d.Fm = d.T * d.fi; d.Fo = d.T * d.fo;
    const int a_total = L->spec->agg_total > 0 ? L->spec->agg_total : L->spec->n_agg;
    d.K = a_total * d.fi;
    if (L->spec->n_towers != d.T || L->spec->n_scalers != 1) { set_error("%s: the sweep spec must carry the towers and ONE (identity) scaler", fn); return false; }
    if (!dgn_linear_supported(d.Fm, 2 * d.Fm, 1) || !dgn_linear_supported(2 * d.Fm, d.Fm, 0) || !dgn_linear_supported(d.K, d.S * d.fo, 1) ||
        !dgn_linear_supported(d.S * d.fo, d.K, 0) || !dgn_linear_supported(d.Fo, d.Fo, 1) || (d.Fm % 16) == 0 || (d.fo & 1) || d.Fo > 1024) {
        set_error("%s: widths outside the streaming Linear kernels (in %d, towers %d x (%d -> %d), %d scalers)", fn, d.Fm, d.T, d.fi, d.fo, d.S);
        return false;
    }
    return true;
}

// the P|Q products on the block-diagonal kernels (dgn_linear_bd_*): shape instantiated, operands 16-byte aligned (DGN_NO_BD=1: dense)
bool use_bd(const DgnTowersLayer* L, const Dims& d) {
    static const bool off = getenv("DGN_NO_BD") != nullptr;
    return !off && dgn_linear_bd_supported(d.T, d.fi) && ((reinterpret_cast<uintptr_t>(L->h) | reinterpret_cast<uintptr_t>(L->pq)) & 15) == 0;
}

DgnMsg sweep_msg(const DgnTowersLayer* L, const Dims& d) {
    DgnMsg m{};
    m.F = d.Fm;
    m.x_src = L->pq; m.ld_src = 2 * d.Fm;
    m.x_dst = L->pq + d.Fm; m.ld_dst = 2 * d.Fm;
    m.x_in = L->h; m.ld_in = d.Fm;
    return m;
}

}  // namespace
}  // namespace dgn

using namespace dgn;

#define DGN_TRY(call)          \
    do {                       \
        const int _rc = (call); \
        if (_rc != 0) return _rc; \
    } while (0)

namespace {
// DgnTowersLayer.drop_*: complete, a probability below 1, and the materialised normalised rows the mask is applied to
bool drop_ok(const DgnTowersLayer* L, const char* fn, bool forward) {
    if (L->drop_p == 0.0f) return true;
    if (!(L->drop_p > 0.0f && L->drop_p < 1.0f) || !L->y1 || (forward && !L->drop_seed) || !L->drop_mask || L->zmask || !L->z) {
        dgn::set_error("%s: dropout needs 0 < drop_p < 1, y1, z, drop_mask (the forward: drop_seed), and no zmask", fn);
        return false;
    }
    return true;
}
}  // namespace

extern "C" int dgn_towers_layer_supported(int32_t n_towers, int32_t f_in, int32_t f_out, int32_t n_scalers, int32_t n_agg_total) {
    const int Fm = n_towers * f_in, Fo = n_towers * f_out, K = n_agg_total * f_in;
    return n_towers >= 1 && n_scalers >= 1 && n_scalers <= 3 && dgn_linear_supported(Fm, 2 * Fm, 1) && dgn_linear_supported(2 * Fm, Fm, 0) &&
           dgn_linear_supported(K, n_scalers * f_out, 1) && dgn_linear_supported(n_scalers * f_out, K, 0) && dgn_linear_supported(Fo, Fo, 1) &&
           (Fm % 16) != 0 && (f_out & 1) == 0 && Fo <= 1024;
}

extern "C" size_t dgn_towers_layer_agg_aux_bytes(const DgnTowersLayer* L) {
    Dims d;
    if (!dims_of(L, d, "dgn_towers_layer_agg_aux_bytes")) return 0;
    DgnTowersLayer tmp = *L;
    static float dummy;                       // (only which operands exist matters here)
    if (!tmp.pq) tmp.pq = &dummy;
    if (!tmp.h) tmp.h = &dummy;
    const DgnMsg msg = sweep_msg(&tmp, d);
    return dgn_agg_aux_bytes(L->graph, L->spec, &msg);
}

extern "C" int dgn_towers_layer_zmask_supported(int32_t n_towers, int32_t f_out) {
    const bool off = option(OPT_NO_ZMASK) != 0;
    const int Fo = n_towers * f_out;
    return !off && n_towers >= 1 && f_out >= 1 && Fo % 16 != 0 && dgn_linear_add_supported(Fo, Fo) && dgn_linear_act_supported(Fo, Fo);
}

// BatchNorm's part of the forward workspace: bn_stats' partials, or the slots the posttrans product's epilogue fills (lin::combine_forward_stats)
static size_t fwd_bn_ws(const Dims& d) { return up256(std::max(dgn_bn_tail_workspace_bytes(d.N, d.Fo), lin::combine_forward_stats_bytes(d.T, d.fo))); }

extern "C" size_t dgn_towers_layer_forward_workspace_bytes(const DgnTowersLayer* L) {
    Dims d;
    if (!dims_of(L, d, "dgn_towers_layer_forward_workspace_bytes")) return 0;
    return fwd_bn_ws(d) + up256(dgn_agg_workspace_bytes(L->graph, L->spec, d.Fm));
}

extern "C" int dgn_towers_layer_forward(const DgnTowersLayer* L, void* stream) {
    const char* fn = "dgn_towers_layer_forward";
    Dims d;
    if (!dims_of(L, d, fn)) return DGN_ERR_INVALID;
    if (d.N == 0) return DGN_OK;
    if (!L->h || !L->w_sd || !L->w_post || !L->w_mix || !L->pq || !L->aggx || !L->y0 || (!L->z && !L->zmask) || !L->out || !L->save_mean ||
        !L->save_invstd || (d.S > 1 && !L->scale)) { set_error("%s: null operand", fn); return DGN_ERR_INVALID; }
    if (!drop_ok(L, fn, true)) return DGN_ERR_INVALID;
    const size_t bn_ws = fwd_bn_ws(d);
    if (L->ws_bytes < dgn_towers_layer_forward_workspace_bytes(L) || (!L->ws && L->ws_bytes)) { set_error("%s: workspace too small", fn); return DGN_ERR_WORKSPACE; }
    char* ws = static_cast<char*>(L->ws);
    // P | Q = h [W_s | W_d]^T + [0 | b]                                                     (dgn_layer.py:226-231, decomposed)
    // (block-diagonal weights: the towers' own [f_in, f_in] blocks only, where that kernel family has the shape)
    if (use_bd(L, d)) DGN_TRY(dgn_linear_bd_forward(d.N, d.T, d.fi, L->h, L->w_sd, d.Fm, L->bias_sd, L->pq, stream));
    else DGN_TRY(dgn_linear_forward(d.N, d.Fm, 2 * d.Fm, 1, L->h, d.Fm, 0, L->w_sd, d.Fm, 0, 0, L->bias_sd, 0, L->pq, 2 * d.Fm, 0, stream));
    // all towers' aggregators (+ the h_in block) in one sweep, tower-major                   (:237-249, :261-264)
    const DgnMsg msg = sweep_msg(L, d);
    const size_t agg_ws = L->ws_bytes - bn_ws;
    DGN_TRY(dgn_agg_forward_aux(L->graph, L->spec, &msg, L->w, L->ld_w, L->log_deg, L->aggx, d.K, L->agg_aux, ws + bn_ws, agg_ws, stream));
    // posttrans([h || agg]) with the folded scalers, bias and graph norm                     (:266-271)
    // (round 6: BatchNorm's column sums of y0 ride in the product's epilogue where nothing else needs a pass over y0 -- no bn_stats launch)
    int stat_slots = 0;
    if (option(OPT_BN_STATS_FUSED) && !L->y1 && !L->n_valid) {
        const int rc = lin::combine_forward_stats(d.N, d.K, d.T, d.S, d.fo, L->aggx, d.N * d.K, L->w_post, d.K, (int64_t)d.S * d.fo * d.K, L->scale, L->b_post,
                                                  L->snorm, L->y0, d.Fo, reinterpret_cast<double*>(ws), bn_ws, &stat_slots, stream);
        if (rc < 0) return DGN_ERR_HIP;
        if (rc > 0) stat_slots = 0;
    }
    if (stat_slots > 0) {
        // the towers' BatchNorm (training statistics)                                        (:272-273)
        DGN_TRY(bn_finalize_launch(d.N, d.Fo, stat_slots, reinterpret_cast<const double*>(ws), L->running_mean, L->running_var, L->momentum, L->eps,
                                   L->save_mean, L->save_invstd, L->num_batches_tracked, L->n_nbt, stream));
    } else {
    DGN_TRY(dgn_linear_combine_forward(d.N, d.K, d.T, d.S, d.fo, L->aggx, d.N * d.K, L->w_post, d.K, (int64_t)d.S * d.fo * d.K, L->scale,
                                       L->b_post, L->snorm, L->y0, d.Fo, stream));
    // the towers' BatchNorm (training statistics)                                            (:272-273)
    // (y1 == NULL: statistics only -- the mixing Linear normalises y0 while it stages its strips, the normalised tensor is never written)
    DGN_TRY(bn_tail_forward_nbt(d.N, d.Fo, L->y0, d.Fo, L->bn_gamma, L->bn_beta, L->running_mean, L->running_var, L->momentum, L->eps, 1, 0,
                                nullptr, L->y1, L->save_mean, L->save_invstd, ws, bn_ws, L->n_valid, L->num_batches_tracked, L->n_nbt, stream));
    }
    // the towers' dropout, on the normalised rows in place                                   (:275)
    if (L->drop_p > 0.0f) DGN_TRY(dgn_dropout_forward(d.N * d.Fo, L->y1, L->drop_p, L->drop_seed, L->drop_offset, L->y1, L->drop_mask, stream));
    // mixing network: Linear -> LeakyReLU, then the layer's residual                         (:318-324)
    static const bool no_mix_fused = getenv("DGN_NO_MIX_FUSED") != nullptr;
    const bool al = ((reinterpret_cast<uintptr_t>(L->z) | reinterpret_cast<uintptr_t>(L->zmask) | reinterpret_cast<uintptr_t>(L->out) |
                      reinterpret_cast<uintptr_t>(L->h)) & 15) == 0;
    if (!L->y1 && !no_mix_fused && al && dgn_linear_add_supported(d.Fo, d.Fo)) {
        // ... all of it in one pass: the normalised operand formed while staged, bias + LeakyReLU + residual in the epilogue
        // (with zmask: the pre-activation leaves as a sign mask, 1/8 of its bytes)
        if (L->zmask) DGN_TRY(dgn_linear_forward_bn_act_mask(d.N, d.Fo, d.Fo, L->y0, L->w_mix, d.Fo, L->save_mean, L->save_invstd, L->bn_gamma, L->bn_beta,
                                                             L->b_mix, 2, L->slope, L->residual ? L->h : nullptr, L->zmask, L->out, stream));
        else DGN_TRY(dgn_linear_forward_bn_act(d.N, d.Fo, d.Fo, L->y0, L->w_mix, d.Fo, L->save_mean, L->save_invstd, L->bn_gamma, L->bn_beta, L->b_mix, 2,
                                               L->slope, L->residual ? L->h : nullptr, L->z, L->out, stream));
        return DGN_OK;
    }
    if (!L->z) { set_error("%s: zmask without the fused mixing-network kernel (operands 16-byte aligned, no y1): give z", fn); return DGN_ERR_INVALID; }
    if (L->y1) DGN_TRY(dgn_linear_forward(d.N, d.Fo, d.Fo, 1, L->y1, d.Fo, 0, L->w_mix, d.Fo, 0, 0, nullptr, 0, L->z, d.Fo, 0, stream));
    else DGN_TRY(dgn_linear_forward_bn(d.N, d.Fo, d.Fo, L->y0, L->w_mix, d.Fo, 0, nullptr, L->z, L->save_mean, L->save_invstd, L->bn_gamma,
                                       L->bn_beta, stream));
    DGN_TRY(dgn_bias_act_forward(d.N, d.Fo, L->z, d.Fo, L->b_mix, 2, L->slope, L->residual ? L->h : nullptr, L->out, stream));
    return DGN_OK;
}

namespace {
struct BwdScratch {
    size_t g_z, g_y1, sums, g_yr, g_aggx, g_pq, g_in, g_hpq, bn_ws, comb_ws, wg_mix, wg_post, wg_sd, agg_ws, dwx, g_sum, total;
};
BwdScratch bwd_scratch(const DgnTowersLayer* L, const Dims& d) {
    BwdScratch s{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += up256(bytes); return at; };
    const size_t NF = (size_t)d.N * d.Fo * 4;
    s.g_z = take(NF); s.g_y1 = take(NF); s.sums = take((size_t)2 * d.Fo * 4);
    s.g_yr = take(NF);
    s.g_aggx = take((size_t)d.T * d.N * d.K * 4);
    s.g_pq = take((size_t)d.N * 2 * d.Fm * 4); s.g_in = take((size_t)d.N * d.Fm * 4); s.g_hpq = take((size_t)d.N * d.Fm * 4);
    s.bn_ws = take(dgn_bn_tail_workspace_bytes(d.N, d.Fo));
    s.comb_ws = take(dgn_scale_combine_backward_workspace_bytes(d.N, d.T, d.fo));
    s.wg_mix = take(dgn_linear_wgrad_workspace_bytes(d.N, d.Fo, d.Fo, 1));
    s.wg_post = take(dgn_linear_wgrad_workspace_bytes(d.N, d.K, d.S * d.fo, d.T));
    s.wg_sd = take(std::max(dgn_linear_wgrad_workspace_bytes(d.N, d.Fm, 2 * d.Fm, 1), dgn_linear_bd_wgrad_workspace_bytes(d.N, d.T, d.fi)));
    s.agg_ws = take(dgn_agg_backward_workspace_bytes(L->graph, L->spec, d.Fm, 1));
    s.dwx = take((size_t)d.Fo * d.Fo * 4);
    s.g_sum = take((size_t)d.T * d.S * d.fo * 4);
    s.total = off;
    return s;
}
}  // namespace

extern "C" size_t dgn_towers_layer_backward_workspace_bytes(const DgnTowersLayer* L) {
    Dims d;
    if (!dims_of(L, d, "dgn_towers_layer_backward_workspace_bytes")) return 0;
    return bwd_scratch(L, d).total;
}

extern "C" int dgn_towers_layer_backward(const DgnTowersLayer* L, const DgnTowersGrads* G, void* stream) {
    const char* fn = "dgn_towers_layer_backward";
    Dims d;
    if (!dims_of(L, d, fn)) return DGN_ERR_INVALID;
    if (!G) { set_error("%s: null grads", fn); return DGN_ERR_INVALID; }
    if (d.N == 0) return DGN_OK;
    if (!G->g_out || !G->g_h || !G->g_w_sd || !G->g_bias_sd || !G->g_w_post || !G->g_b_post || !G->g_gamma || !G->g_beta || !G->g_w_mix ||
        !G->g_b_mix) { set_error("%s: null gradient buffer", fn); return DGN_ERR_INVALID; }
    if (!drop_ok(L, fn, false)) return DGN_ERR_INVALID;
    const BwdScratch s = bwd_scratch(L, d);
    if (!L->ws || L->ws_bytes < s.total) { set_error("%s: workspace too small (%zu < %zu)", fn, L->ws_bytes, s.total); return DGN_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(L->ws);
    auto f = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    float *g_z = f(s.g_z), *g_y1 = f(s.g_y1), *sums = f(s.sums), *g_yr = f(s.g_yr), *g_aggx = f(s.g_aggx), *g_pq = f(s.g_pq),
          *g_in = f(s.g_in), *g_hpq = f(s.g_hpq);
    // bias + LeakyReLU (+ residual: its gradient is g_out itself, added at the end) and the mixing Linear's input gradient in ONE pass:
    // g_z = g_out * act'(z + b_mix) is formed while the strips are staged and leaves as a side output for the weight gradient, whose
    // ones-column delivers the bias gradient (the separate path: dgn_bias_act_backward, then the two products)
    const bool fused_act = L->y1 == nullptr && d.Fo % 16 != 0 && (reinterpret_cast<uintptr_t>(g_z) & 15) == 0 && dgn_linear_act_supported(d.Fo, d.Fo);
    // (no dropout between BatchNorm and the mixing network, every row of the buffer a row of the batch: BatchNorm's column sums follow
    //  from the mixing weight gradient, mix_bn_finalize above)
    const bool bn_derived = fused_act && L->drop_p == 0.0f && !L->n_valid && option(OPT_BN_FROM_WGRAD) != 0;
    // Round 6, on top of that: the weight gradient FIRST, its G operand formed from g_out and the mask while staged (g_z is never
    // written); with the column sums known, the input-gradient product turns its rows g_y1 into g_yr = snorm * BatchNorm'(g_y1) in its
    // epilogue, tower-major (g_y1 is never written, the combine-backward pass is gone), and posttrans' bias gradient comes out of
    // its weight-gradient pass as the identity scaler's column sums.  8 N Fo floats of traffic -> 5 N Fo, two launches less.
    const bool mix_fused = bn_derived && L->zmask && (reinterpret_cast<uintptr_t>(L->zmask) & 15) == 0 && L->id_slot1 >= 1 && L->id_slot1 <= d.S && d.K % 16 != 0 && d.T <= 15 &&
                           (reinterpret_cast<uintptr_t>(L->y0) & 15) == 0 && dgn_linear_bnb_supported(d.Fo, d.Fo) && option(OPT_MIX_BWD_FUSED) != 0;
    if (mix_fused) {
        DGN_TRY(dgn_linear_wgrad_bn_act_mask(d.N, d.Fo, d.Fo, G->g_out, L->zmask, 2, L->slope, L->y0, f(s.dwx), d.Fo, G->g_b_mix, L->save_mean,
                                             L->save_invstd, nullptr, nullptr, ws + s.wg_mix, dgn_linear_wgrad_workspace_bytes(d.N, d.Fo, d.Fo, 1), stream));
        hipLaunchKernelGGL(mix_bn_finalize, dim3(d.Fo), dim3(64), 0, st, d.Fo, (const float*)f(s.dwx), (const float*)G->g_b_mix, L->w_mix, (int64_t)d.Fo,
                           L->bn_gamma, L->bn_beta, G->g_w_mix, (int64_t)d.Fo, G->g_gamma, G->g_beta, sums);
        DGN_HIP_CHECK(hipGetLastError());
        DGN_TRY(dgn_linear_forward_act_mask_bnb(d.N, d.Fo, d.Fo, G->g_out, L->zmask, 2, L->slope, L->w_mix, d.Fo, 1, L->y0, L->save_mean, L->save_invstd,
                                                L->bn_gamma, sums, L->snorm, d.fo, g_yr, d.N * d.fo, stream));
    } else if (fused_act) {
        if (L->zmask) DGN_TRY(dgn_linear_forward_act_mask(d.N, d.Fo, d.Fo, G->g_out, L->zmask, 2, L->slope, L->w_mix, d.Fo, 1, g_y1, g_z, stream));
        else DGN_TRY(dgn_linear_forward_act(d.N, d.Fo, d.Fo, G->g_out, L->z, L->b_mix, 2, L->slope, L->w_mix, d.Fo, 1, g_y1, g_z, stream));
        if (bn_derived) {
            DGN_TRY(dgn_linear_wgrad_bn(d.N, d.Fo, d.Fo, g_z, L->y0, f(s.dwx), d.Fo, G->g_b_mix, L->save_mean, L->save_invstd, nullptr, nullptr,
                                        ws + s.wg_mix, dgn_linear_wgrad_workspace_bytes(d.N, d.Fo, d.Fo, 1), stream));
            hipLaunchKernelGGL(mix_bn_finalize, dim3(d.Fo), dim3(64), 0, st, d.Fo, (const float*)f(s.dwx), (const float*)G->g_b_mix, L->w_mix, (int64_t)d.Fo,
                               L->bn_gamma, L->bn_beta, G->g_w_mix, (int64_t)d.Fo, G->g_gamma, G->g_beta, sums);
            DGN_HIP_CHECK(hipGetLastError());
        } else {
            DGN_TRY(dgn_linear_wgrad_bn(d.N, d.Fo, d.Fo, g_z, L->y0, G->g_w_mix, d.Fo, G->g_b_mix, L->save_mean, L->save_invstd, L->bn_gamma, L->bn_beta,
                                        ws + s.wg_mix, dgn_linear_wgrad_workspace_bytes(d.N, d.Fo, d.Fo, 1), stream));
        }
    } else {
        if (!L->z) { set_error("%s: zmask without the fused activation-gradient kernel: give z", fn); return DGN_ERR_INVALID; }
        DGN_TRY(dgn_bias_act_backward(d.N, d.Fo, G->g_out, L->z, d.Fo, L->b_mix, 2, L->slope, g_z, G->g_b_mix, ws + s.bn_ws,
                                      dgn_bn_tail_workspace_bytes(d.N, d.Fo), stream));
        // mixing Linear: input and weight gradients
        DGN_TRY(dgn_linear_forward(d.N, d.Fo, d.Fo, 1, g_z, d.Fo, 0, L->w_mix, d.Fo, 0, 1, nullptr, 0, g_y1, d.Fo, 0, stream));
        if (L->y1) DGN_TRY(dgn_linear_wgrad(d.N, d.Fo, d.Fo, 1, g_z, d.Fo, 0, L->y1, d.Fo, 0, G->g_w_mix, d.Fo, 0, nullptr, 0, ws + s.wg_mix,
                                            dgn_linear_wgrad_workspace_bytes(d.N, d.Fo, d.Fo, 1), stream));
        else DGN_TRY(dgn_linear_wgrad_bn(d.N, d.Fo, d.Fo, g_z, L->y0, G->g_w_mix, d.Fo, nullptr, L->save_mean, L->save_invstd, L->bn_gamma, L->bn_beta,
                                         ws + s.wg_mix, dgn_linear_wgrad_workspace_bytes(d.N, d.Fo, d.Fo, 1), stream));
    }
    // the towers' dropout: the mask on the gradient of the normalised rows, in place         (:275)
    if (L->drop_p > 0.0f) DGN_TRY(dgn_dropout_backward(d.N * d.Fo, g_y1, L->drop_mask, L->drop_p, g_y1, stream));
    // BatchNorm: column sums + affine gradients; its input gradient is formed inside the combine backward
    if (!bn_derived)      // (else: the sums came out of the mixing weight gradient above)
        DGN_TRY(dgn_bn_tail_backward(d.N, d.Fo, g_y1, L->y0, d.Fo, L->bn_gamma, L->bn_beta, L->save_mean, L->save_invstd, 0, nullptr, G->g_gamma,
                                     G->g_beta, sums, ws + s.bn_ws, dgn_bn_tail_workspace_bytes(d.N, d.Fo), L->n_valid, stream));
    if (!mix_fused) {
        DgnBnGrad bn{};
        bn.g_out = g_y1; bn.y = L->y0; bn.ld = d.Fo; bn.gamma = L->bn_gamma; bn.beta = L->bn_beta; bn.mean = L->save_mean; bn.invstd = L->save_invstd;
        bn.sums = sums; bn.relu = 0; bn.n_valid = L->n_valid;
        DGN_TRY(scale_combine_backward_impl(d.N, d.T, 1, d.fo, nullptr, 0, nullptr, L->snorm, g_yr, G->g_b_post, ws + s.comb_ws,
                                           dgn_scale_combine_backward_workspace_bytes(d.N, d.T, d.fo), &bn, stream, 1));
    }
    // posttrans: the scaler expansion happens inside the two products
    DGN_TRY(dgn_linear_combine_backward_input(d.N, d.T, d.S, d.fo, d.K, g_yr, d.N * d.fo, L->scale, L->w_post, d.K, (int64_t)d.S * d.fo * d.K,
                                              g_aggx, d.N * d.K, stream));
    // (mix_fused: the column sums of the expanded gradient ride in the pass -- the identity scaler's block of them is d b_post)
    float* g_sum = mix_fused ? f(s.g_sum) : nullptr;
    // (... and the finalize kernel writes that block to g_b_post: no launch of its own)
    DGN_TRY(lin::combine_backward_weight_bias_pick(d.N, d.T, d.S, d.fo, d.K, g_yr, d.N * d.fo, L->scale, L->aggx, d.N * d.K, G->g_w_post, d.K,
                                                   (int64_t)d.S * d.fo * d.K, g_sum, mix_fused ? G->g_b_post : nullptr, mix_fused ? L->id_slot1 - 1 : 0,
                                                   ws + s.wg_post, dgn_linear_wgrad_workspace_bytes(d.N, d.K, d.S * d.fo, d.T), stream));
    // the sweep: d P | d Q in one [N, 2 Fm] buffer, d h_in
    const DgnMsg msg = sweep_msg(L, d);
    DgnMsgGrad gr{};
    gr.g_src = g_pq; gr.ld_src = 2 * d.Fm;
    gr.g_dst = g_pq + d.Fm; gr.ld_dst = 2 * d.Fm;
    gr.g_in = g_in; gr.ld_in = d.Fm;
    gr.accumulate = 0;
    DGN_TRY(dgn_agg_backward_aux(L->graph, L->spec, &msg, L->w, L->ld_w, L->log_deg, g_aggx, d.K, L->agg_aux, &gr, ws + s.agg_ws,
                                 dgn_agg_backward_workspace_bytes(L->graph, L->spec, d.Fm, 1), stream));
    // P|Q Linear: input gradient, weight + bias gradient (the bias rides in the weight-gradient pass)
    // d h = [residual] + d h_in + (d P|Q) W_sd: as the product's epilogue ((d h_in + product) + residual, add3's order) where the shapes
    // allow, else the product and a three-way add
    const float* res = L->residual ? G->g_out : nullptr;
    static const bool no_add_epilogue = getenv("DGN_NO_ADD_EPILOGUE") != nullptr;
    const bool al = ((reinterpret_cast<uintptr_t>(G->g_h) | reinterpret_cast<uintptr_t>(G->g_out)) & 15) == 0;
    const bool fused_add = !no_add_epilogue && al && (reinterpret_cast<uintptr_t>(g_in) & 15) == 0 && dgn_linear_add_supported(2 * d.Fm, d.Fm);
    if (use_bd(L, d) && al && ((reinterpret_cast<uintptr_t>(g_in) | reinterpret_cast<uintptr_t>(g_pq)) & 15) == 0) {
        // the towers' own blocks only: (d h_in + (d P|Q) W_sd) + residual in the product's epilogue, add3's order
        if (option(OPT_BD_BWD_FUSED) && d.fi <= 16 && (reinterpret_cast<uintptr_t>(L->h) & 15) == 0) {
            // ... and its weight gradient in the same pass over d(P|Q) (round 6: bd_backward_both)
            DGN_TRY(lin::bd_backward_both_launch(d.N, d.T, d.fi, g_pq, L->w_sd, d.Fm, L->h, g_in, res, G->g_h, G->g_w_sd, d.Fm, G->g_bias_sd, ws + s.wg_sd,
                                                 dgn_linear_bd_wgrad_workspace_bytes(d.N, d.T, d.fi), stream));
            return DGN_OK;
        }
        DGN_TRY(dgn_linear_bd_backward_input(d.N, d.T, d.fi, g_pq, L->w_sd, d.Fm, g_in, res, G->g_h, stream));
        DGN_TRY(dgn_linear_bd_wgrad(d.N, d.T, d.fi, g_pq, L->h, G->g_w_sd, d.Fm, G->g_bias_sd, ws + s.wg_sd,
                                    dgn_linear_bd_wgrad_workspace_bytes(d.N, d.T, d.fi), stream));
        return DGN_OK;
    }
    if (fused_add) DGN_TRY(dgn_linear_forward_add(d.N, 2 * d.Fm, d.Fm, g_pq, L->w_sd, d.Fm, 1, g_in, res, G->g_h, stream));
    else DGN_TRY(dgn_linear_forward(d.N, 2 * d.Fm, d.Fm, 1, g_pq, 2 * d.Fm, 0, L->w_sd, d.Fm, 0, 1, nullptr, 0, g_hpq, d.Fm, 0, stream));
    DGN_TRY(dgn_linear_wgrad(d.N, d.Fm, 2 * d.Fm, 1, g_pq, 2 * d.Fm, 0, L->h, d.Fm, 0, G->g_w_sd, d.Fm, 0, G->g_bias_sd, 0, ws + s.wg_sd,
                             dgn_linear_wgrad_workspace_bytes(d.N, d.Fm, 2 * d.Fm, 1), stream));
    if (fused_add) return DGN_OK;
    const int64_t n = d.N * d.Fm, n4 = n / 4;
    if (al && n4 > 0) {
        hipLaunchKernelGGL(add3_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, n4, reinterpret_cast<const float4*>(res),
                           reinterpret_cast<const float4*>(g_in), reinterpret_cast<const float4*>(g_hpq), reinterpret_cast<float4*>(G->g_h));
        if (n > 4 * n4) hipLaunchKernelGGL(add3_tail, dim3(1), dim3(256), 0, st, 4 * n4, n, res, g_in, g_hpq, G->g_h);
    } else {
        hipLaunchKernelGGL(add3_tail, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int64_t)0, n, res, g_in, g_hpq, G->g_h);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_assemble_params(int64_t n_out, const int64_t* param_ptrs, const int32_t* map_param, const int32_t* map_off, float* out,
                                   void* stream) {
    if (n_out < 0 || (n_out > 0 && (!param_ptrs || !map_param || !map_off || !out))) { set_error("dgn_assemble_params: null argument"); return DGN_ERR_INVALID; }
    if (n_out == 0) return DGN_OK;
    hipLaunchKernelGGL(assemble_params, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n_out, param_ptrs,
                       map_param, map_off, out);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
