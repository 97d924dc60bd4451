// Whole-layer entry points of the simple and complex DGN layers (include/dgn_hip.h: dgn_dense_layer_*), the counterpart of
// dgn_towers.hip for realworld_benchmark/nets/dgn_layer.py:135-202 (DGNLayerSimple) and :52-132 (DGNLayerComplex) in their fused
// form (dgn_amd/dgn_layer.py): single-affine pretrans / posttrans, degree scalers folded behind the posttrans Linear, training-mode
// BatchNorm -> ReLU -> residual, no edge features, no dropout.  ONE call enqueues a whole direction on the caller's stream:
//   forward    [hp = h padded to an even width]  ->  [complex: pq = hp [W_s | W_d]^T + [0 | b]]  ->  the sweep (+ h_in block)  ->
//              z = agg W_f^T  ->  y = snorm (b + sum_s scale_s z_s)  ->  out = relu(BatchNorm(y)) [+ h]
//   backward   the same chain in reverse, every parameter gradient in the REFERENCE's layout (posttrans weight [fo, (S A (+1)) F],
//              pretrans weight [F, 2 F]).
// What the host side did with torch ops per step -- zero-padding an odd hidden size (75 / 65 / 45 / 47), folding the posttrans
// weight scaler-major with zero columns at the padding and the h block in the identity scaler's rows, un-folding its gradient,
// adding the residual and d h_in contributions -- are four small kernels here; everything else is the library's own entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "dgn_common.hpp"

namespace dgn {
namespace {

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

struct Dims {
    int64_t N;
    int F0, Fp, fo, S, A, Ab, K, n, cx;      // Ab = aggregator blocks of the sweep (A + 1 with the h_in block), K = Ab * Fp, n = S * fo
    int dc;                                  // degree-class posttrans (dgn_dc_kernels.hpp): one f_out-column product per in-degree class
};

bool dims_of(const DgnDenseLayer* L, Dims& d, const char* fn) {
    if (!L || !L->graph || !L->spec) { set_error("%s: null layer / graph / spec", fn); return false; }
    d.N = L->graph->n_nodes; d.F0 = L->f_in; d.fo = L->f_out; d.S = L->n_scalers; d.A = L->n_agg; d.cx = L->type == 1;
    if (d.F0 < 2 || d.fo < 2 || d.S < 1 || d.S > 3 || d.A < 1 || (L->type != 0 && L->type != 1)) { set_error("%s: bad widths / counts", fn); return false; }
    d.Fp = d.F0 + (d.F0 & 1);
    d.Ab = d.A + d.cx;
    d.K = d.Ab * d.Fp;
    d.n = d.S * d.fo;
    const int a_total = L->spec->agg_total > 0 ? L->spec->agg_total : L->spec->n_agg;
    if (a_total != d.Ab || L->spec->n_towers != 1 || L->spec->n_scalers != 1) {
        set_error("%s: the sweep spec must list the aggregators%s with ONE (identity) scaler and one tower", fn, d.cx ? " + the h_in block" : "");
        return false;
    }
    if (d.cx && (L->id_slot < 0 || L->id_slot >= d.S)) { set_error("%s: complex layer needs the identity scaler's slot", fn); return false; }
    if (d.fo > 1024 || d.K < 4 || d.n < 4 || d.K > 4096 || d.n > 4096) { set_error("%s: widths outside the kernels' range", fn); return false; }
    d.dc = L->dc && d.S > 1 && dgn_dc_supported(d.K, d.fo) && dgn_dc_supported(d.fo, d.K) && dgn_dc_wgrad_supported(d.K, d.fo);
    return true;
}

// ---- the small kernels ------------------------------------------------------------------------------------------------------------
// hp[r][c] = c < F0 ? h[r][c] : 0
__global__ __launch_bounds__(256) void pad_rows(int64_t n, int F0, int Fp, const float* __restrict__ h, float* __restrict__ hp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * Fp) return;
    const int64_t r = i / Fp;
    const int c = (int)(i - r * Fp);
    hp[i] = c < F0 ? h[r * F0 + c] : 0.f;
}
// The same, four padded outputs per thread (the dense [n, Fp] output as flat 16-byte chunks; the source row's floats as one 4-byte-aligned 16-byte
// load where the chunk lies inside the row's F0 columns)
typedef float f4u_pad_t __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void pad_rows4(int64_t n, int F0, int Fp, const float* __restrict__ h, float* __restrict__ hp) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = n * Fp, i = 4 * q;
    if (i >= total) return;
    const int64_t r = i / Fp;
    const int c = (int)(i - r * Fp);
    if (i + 3 < total) {
        float v[4];
        if (c + 3 < F0) {
            const f4u_pad_t a = *reinterpret_cast<const f4u_pad_t*>(h + r * F0 + c);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ce = c + e >= Fp ? c + e - Fp : c + e;
                const int64_t re = c + e >= Fp ? r + 1 : r;
                v[e] = ce < F0 ? h[re * F0 + ce] : 0.f;
            }
        }
        reinterpret_cast<float4*>(hp)[q] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int64_t k = i; k < total; ++k) {
            const int64_t rk = k / Fp;
            const int ck = (int)(k - rk * Fp);
            hp[k] = ck < F0 ? h[rk * F0 + ck] : 0.f;
        }
    }
}
// g_h[r][c] = g_hp[r][c] (+ g_pqh[r][c]) (+ g_res[r][c]): un-pad and join the contributions to d h
__global__ __launch_bounds__(256) void unpad_add(int64_t n, int F0, int Fp, const float* __restrict__ g_hp, const float* __restrict__ g2,
                                                 const float* __restrict__ g_res, float* __restrict__ g_h) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * F0) return;
    const int64_t r = i / F0;
    const int c = (int)(i - r * F0);
    float v = g_hp[r * Fp + c];
    if (g2) v += g2[r * Fp + c];
    if (g_res) v += g_res[i];
    g_h[i] = v;
}
// The same, four outputs per thread (the dense [n, F0] output and residual as flat 16-byte chunks; the padded rows' four floats as one
// 4-byte-aligned 16-byte load where the chunk lies inside one row): the one-element-per-thread form ran at 4.5 TB/s on [275 k, 75].  Same
// additions in the same order.
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void unpad_add4(int64_t n, int F0, int Fp, const float* __restrict__ g_hp, const float* __restrict__ g2,
                                                  const float* __restrict__ g_res, float* __restrict__ g_h) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = n * F0, i = 4 * q;
    if (i >= total) return;
    const int64_t r = i / F0;
    const int c = (int)(i - r * F0);
    if (i + 3 < total) {
        float v[4];
        if (c + 3 < F0) {
            const f4u_t a = *reinterpret_cast<const f4u_t*>(g_hp + r * Fp + c);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
            if (g2) {
                const f4u_t b = *reinterpret_cast<const f4u_t*>(g2 + r * Fp + c);
                v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ce = c + e >= F0 ? c + e - F0 : c + e;
                const int64_t at = (c + e >= F0 ? r + 1 : r) * Fp + ce;
                v[e] = g_hp[at];
                if (g2) v[e] += g2[at];
            }
        }
        if (g_res) {
            const float4 w = reinterpret_cast<const float4*>(g_res)[q];
            v[0] += w.x; v[1] += w.y; v[2] += w.z; v[3] += w.w;
        }
        reinterpret_cast<float4*>(g_h)[q] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int64_t k = i; k < total; ++k) {
            const int64_t rk = k / F0;
            const int ck = (int)(k - rk * F0);
            float v = g_hp[rk * Fp + ck];
            if (g2) v += g2[rk * Fp + ck];
            if (g_res) v += g_res[k];
            g_h[k] = v;
        }
    }
}
// folded posttrans weight W_f [S fo][K] (+ its transpose [K][S fo]) from the reference layout W [fo][hoff + S A F0], hoff = F0 (complex: the h
// columns come first, dgn_layer.py:116-119) or 0: row s fo + o, column a Fp + f  <-  W[o][hoff + (s A + a) F0 + f]; the padded feature column
// (f == F0) is zero; complex: block a == A holds the h columns in the identity scaler's rows only
__global__ __launch_bounds__(256) void fold_post(int S, int fo, int A, int Ab, int F0, int Fp, int hoff, int id_slot, int64_t ldw,
                                                 const float* __restrict__ W, float* __restrict__ wf, float* __restrict__ wft) {
    const int K = Ab * Fp, n = S * fo;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int r = i / K, c = i - r * K, s = r / fo, o = r - s * fo, a = c / Fp, f = c - a * Fp;
    float v = 0.f;
    if (f < F0) {
        if (a < A) v = W[(int64_t)o * ldw + hoff + (s * A + a) * F0 + f];
        else if (s == id_slot) v = W[(int64_t)o * ldw + f];
    }
    wf[i] = v;
    wft[(int64_t)c * n + r] = v;
}
// ... and its adjoint: g_W[o][col] from g_wf [S fo][K]
__global__ __launch_bounds__(256) void unfold_post(int S, int fo, int A, int Ab, int F0, int Fp, int hoff, int id_slot, int64_t ldw,
                                                   const float* __restrict__ g_wf, float* __restrict__ g_W) {
    const int K = Ab * Fp, cols = hoff + S * A * F0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fo * cols) return;
    const int o = i / cols, col = i - o * cols;
    float v;
    if (col < hoff) v = g_wf[(int64_t)(id_slot * fo + o) * K + A * Fp + col];
    else {
        const int q = col - hoff, sa = q / F0, f = q - sa * F0, s = sa / A, a = sa - s * A;
        v = g_wf[(int64_t)(s * fo + o) * K + a * Fp + f];
    }
    g_W[(int64_t)o * ldw + col] = v;
}
// complex pretrans: w_sd [2 Fp][Fp] (+ transpose [Fp][2 Fp]) and bias_sd [2 Fp] from W_pre [F0][2 F0], b_pre [F0]:
// row j Fp + a, column b  <-  W_pre[a][j F0 + b]   (j = 0: source half -> P, j = 1: destination half -> Q, which carries the bias)
__global__ __launch_bounds__(256) void fold_pre(int F0, int Fp, int64_t ldw, const float* __restrict__ W, const float* __restrict__ b,
                                                float* __restrict__ wsd, float* __restrict__ wsdt, float* __restrict__ bsd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * Fp) {
        const int j = i / Fp, a = i - j * Fp;
        bsd[i] = (j == 1 && a < F0 && b) ? b[a] : 0.f;
    }
    if (i >= 2 * Fp * Fp) return;
    const int r = i / Fp, c = i - r * Fp, j = r / Fp, a = r - j * Fp;
    const float v = (a < F0 && c < F0) ? W[(int64_t)a * ldw + j * F0 + c] : 0.f;
    wsd[i] = v;
    wsdt[(int64_t)c * 2 * Fp + r] = v;
}
__global__ __launch_bounds__(256) void unfold_pre(int F0, int Fp, int64_t ldw, const float* __restrict__ g_wsd, const float* __restrict__ g_bsd,
                                                  float* __restrict__ g_W, float* __restrict__ g_b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F0 && g_b) g_b[i] = g_bsd[Fp + i];
    if (i >= F0 * 2 * F0) return;
    const int a = i / (2 * F0), q = i - a * 2 * F0, j = q / F0, c = q - j * F0;
    g_W[(int64_t)a * ldw + q] = g_wsd[(int64_t)(j * Fp + a) * Fp + c];
}

// C = A op(W): the streaming kernels where the shape fits them (even widths <= 160, dense rows), the wide GEMM otherwise
int lin_fwd(int64_t N, int k, int n, const float* a, const float* w_nk, const float* bias, float* c, void* stream) {
    const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(c)) & 7) == 0;
    if (al && dgn_linear_supported(k, n, 0)) return dgn_linear_forward(N, k, n, 1, a, k, 0, w_nk, k, 0, 0, bias, 0, c, n, 0, stream);
    return dgn_gemm_forward(N, k, n, a, k, w_nk, k, 0, bias, c, n, stream);
}
size_t wgrad_ws(int64_t N, int k, int n) { return std::max(dgn_linear_wgrad_workspace_bytes(N, k, n, 1), dgn_gemm_wgrad_workspace_bytes(N, k, n)); }
int lin_wgrad(int64_t N, int k, int n, const float* g, const float* x, float* dw, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    const bool al = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x)) & 7) == 0;
    if (al && dgn_linear_supported(k, n, 1) && (!dbias || k % 16 != 0))
        return dgn_linear_wgrad(N, k, n, 1, g, n, 0, x, k, 0, dw, k, 0, dbias, 0, ws, ws_bytes, stream);
    return dgn_gemm_wgrad(N, k, n, g, n, x, k, dw, k, dbias, ws, ws_bytes, stream);
}

// simple layer, odd hidden size, no hub rows, a list the sweep has odd-width kernels for: no padded copy of h
bool odd_direct(const DgnDenseLayer* L, const Dims& d) {
    return !d.cx && d.Fp != d.F0 && d.Fp >= 4 && L->graph->n_hub == 0 && option(OPT_ODD_DIRECT) != 0 && dgn_agg_f_valid_supported(L->spec) != 0;
}

DgnMsg sweep_msg(const DgnDenseLayer* L, const Dims& d, const float* hp) {
    DgnMsg m{};
    m.F = d.Fp;
    if (d.cx) {
        m.x_src = L->pq; m.ld_src = 2 * d.Fp;
        m.x_dst = L->pq + d.Fp; m.ld_dst = 2 * d.Fp;
    } else if (odd_direct(L, d)) {
        // simple layer at an odd hidden size, a list with odd-width kernels: the sweep reads the un-padded rows itself (DgnMsg.f_valid) --
        // the padded copy of h was 53 us of the ZINC simple layer's 0.95 ms step
        m.x_src = L->h; m.ld_src = d.F0; m.x_in = L->h; m.ld_in = d.F0; m.f_valid = d.F0;
        return m;
    } else {
        m.x_src = hp; m.ld_src = d.Fp;
    }
    m.x_in = hp; m.ld_in = d.Fp;
    return m;
}

struct BwdScratch { size_t g_z, sums, g_agg, g_wf, g_pq, g_hp, g_hq, g_wsd, g_bsd, bn_ws, comb_ws, wg_ws, agg_ws, total; };
BwdScratch bwd_scratch(const DgnDenseLayer* L, const Dims& d) {
    BwdScratch s{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += up256(bytes); return at; };
    s.g_z = take((size_t)d.N * d.n * 4); s.sums = take((size_t)2 * d.fo * 4);
    s.g_agg = take((size_t)d.N * d.K * 4); s.g_wf = take((size_t)d.n * d.K * 4);
    s.g_pq = take(d.cx ? (size_t)d.N * 2 * d.Fp * 4 : 0); s.g_hp = take((size_t)d.N * d.Fp * 4); s.g_hq = take(d.cx ? (size_t)d.N * d.Fp * 4 : 0);
    s.g_wsd = take(d.cx ? (size_t)2 * d.Fp * d.Fp * 4 : 0); s.g_bsd = take(d.cx ? (size_t)2 * d.Fp * 4 : 0);
    s.bn_ws = take(dgn_bn_tail_workspace_bytes(d.N, d.fo));
    s.comb_ws = take(dgn_scale_combine_backward_workspace_bytes(d.N, 1, d.fo));
    s.wg_ws = take(std::max(d.dc ? dgn_dc_wgrad_workspace_bytes(L->dc->n_units, d.K, d.fo) : wgrad_ws(d.N, d.K, d.n),
                            d.cx ? wgrad_ws(d.N, d.Fp, 2 * d.Fp) : (size_t)0));
    s.agg_ws = take(dgn_agg_backward_workspace_bytes(L->graph, L->spec, d.Fp, 1));
    s.total = off;
    return s;
}

}  // namespace
}  // namespace dgn

using namespace dgn;

#define DGN_TRY(call)              \
    do {                           \
        const int _rc = (call);    \
        if (_rc != 0) return _rc;  \
    } while (0)

extern "C" size_t dgn_dense_layer_agg_aux_bytes(const DgnDenseLayer* L) {
    Dims d;
    if (!dims_of(L, d, "dgn_dense_layer_agg_aux_bytes")) return 0;
    static float dummy;                       // (only which operands exist matters here)
    DgnDenseLayer tmp = *L;
    if (!tmp.pq) tmp.pq = &dummy;
    const DgnMsg msg = sweep_msg(&tmp, d, &dummy);
    return dgn_agg_aux_bytes(L->graph, L->spec, &msg);
}

extern "C" int dgn_dense_layer_supported(int32_t type, int32_t f_in, int32_t f_out, int32_t n_scalers, int32_t n_agg) {
    const int Fp = f_in + (f_in & 1), K = (n_agg + (type == 1)) * Fp, n = n_scalers * f_out;
    return (type == 0 || type == 1) && f_in >= 2 && f_out >= 2 && f_out <= 1024 && n_scalers >= 1 && n_scalers <= 3 && n_agg >= 1 &&
           K >= 4 && K <= 4096 && n >= 4 && n <= 4096 && Fp >= 4;
}

// BatchNorm's part of the forward workspace: bn_stats' partials, or the slots the degree-class product's epilogue fills (dc::gemm_stats)
// (... or the combine pass': one slot per slab, at most 2 048 slabs ride)
static size_t fwd_bn_ws(const Dims& d) {
    return up256(std::max(dgn_bn_tail_workspace_bytes(d.N, d.fo), d.dc ? dc::gemm_stats_bytes(d.fo) : (size_t)2 * d.fo * 2048 * sizeof(double)));
}

extern "C" size_t dgn_dense_layer_forward_workspace_bytes(const DgnDenseLayer* L) {
    Dims d;
    if (!dims_of(L, d, "dgn_dense_layer_forward_workspace_bytes")) return 0;
    return up256(d.dc ? 0 : (size_t)d.N * d.n * 4) + fwd_bn_ws(d) + up256(dgn_agg_workspace_bytes(L->graph, L->spec, d.Fp));
}

extern "C" int dgn_dense_layer_forward(const DgnDenseLayer* L, void* stream) {
    const char* fn = "dgn_dense_layer_forward";
    Dims d;
    if (!dims_of(L, d, fn)) return DGN_ERR_INVALID;
    if (d.N == 0) return DGN_OK;
    const bool padded = d.Fp != d.F0;
    if (!L->h || !L->w_post || !L->agg || !L->y || !L->wf || !L->out || !L->save_mean || !L->save_invstd || (padded && !L->hp) ||
        (d.S > 1 && !L->scale) || (d.cx && (!L->w_pre || !L->pq || !L->wsd))) { set_error("%s: null operand", fn); return DGN_ERR_INVALID; }
    if (L->ws_bytes < dgn_dense_layer_forward_workspace_bytes(L) || !L->ws) { set_error("%s: workspace too small", fn); return DGN_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(L->ws);
    float* z = reinterpret_cast<float*>(ws);
    const size_t z_b = up256(d.dc ? 0 : (size_t)d.N * d.n * 4), bn_b = fwd_bn_ws(d);
    const float* hp = L->h;
    if (padded && !odd_direct(L, d)) {
        if (d.Fp >= 4 && d.Fp - d.F0 < 4 && (reinterpret_cast<uintptr_t>(L->hp) & 15) == 0)
            hipLaunchKernelGGL(pad_rows4, dim3(nblk((d.N * d.Fp + 3) / 4)), dim3(256), 0, st, d.N, d.F0, d.Fp, L->h, L->hp);
        else
            hipLaunchKernelGGL(pad_rows, dim3(nblk(d.N * d.Fp)), dim3(256), 0, st, d.N, d.F0, d.Fp, L->h, L->hp);
        hp = L->hp;
    }
    const int hoff = d.cx ? d.F0 : 0;
    const int64_t ld_post = hoff + (int64_t)d.S * d.A * d.F0;
    float* wft = L->wf + (size_t)d.n * d.K;
    DgnDcLayout lay{};                              // (degree-class route: the kernels read / write the reference layout themselves)
    lay.n_agg = d.A; lay.f_pad = d.Fp; lay.f_in = d.F0; lay.h_off = hoff; lay.id_slot = d.cx ? L->id_slot : -1; lay.ld = ld_post;
    if (!d.dc)
        hipLaunchKernelGGL(fold_post, dim3(nblk((int64_t)d.n * d.K)), dim3(256), 0, st, d.S, d.fo, d.A, d.Ab, d.F0, d.Fp, hoff, d.cx ? L->id_slot : -1, ld_post,
                           L->w_post, L->wf, wft);
    if (d.cx) {
        float* wsdt = L->wsd + (size_t)2 * d.Fp * d.Fp;
        float* bsd = wsdt + (size_t)2 * d.Fp * d.Fp;
        hipLaunchKernelGGL(fold_pre, dim3(nblk((int64_t)2 * d.Fp * d.Fp)), dim3(256), 0, st, d.F0, d.Fp, (int64_t)2 * d.F0, L->w_pre, L->b_pre, L->wsd, wsdt, bsd);
        DGN_HIP_CHECK(hipGetLastError());
        // P | Q = hp [W_s | W_d]^T + [0 | b]                                                    (dgn_layer.py:75-80, decomposed)
        DGN_TRY(lin_fwd(d.N, d.Fp, 2 * d.Fp, hp, L->wsd, bsd, L->pq, stream));
    }
    DGN_HIP_CHECK(hipGetLastError());
    // the sweep (+ the h_in pass-through block of the complex layer)                            (:86-98 / :161-173)
    const DgnMsg msg = sweep_msg(L, d, hp);
    DGN_TRY(dgn_agg_forward_aux(L->graph, L->spec, &msg, L->w, L->ld_w, L->log_deg, L->agg, d.K, L->agg_aux, ws + z_b + bn_b, L->ws_bytes - z_b - bn_b,
                                stream));
    // posttrans with the folded scalers, bias and graph norm                                    (:116-122 / :187-193)
    if (d.dc) {
        // one product per in-degree class: y = snorm (b + agg W_class^T), W_class = sum_s scale_s(class) W_f[s]
        float* wc = L->wf + (size_t)2 * d.n * d.K;
        float* wct = wc + (size_t)DGN_DC_CLASSES * d.fo * d.K;
        DGN_TRY(dgn_dc_fold(L->dc, d.S, d.fo, d.K, 1, L->w_post, &lay, wc, wct, stream));
        // (round 6: BatchNorm's column sums of y ride in the product's epilogue -- no bn_stats pass; option bn_stats_fused)
        if (option(OPT_BN_STATS_FUSED) && !L->n_valid) {
            int slots = 0;
            DGN_TRY(dc::gemm_stats(L->dc, d.K, d.fo, 1, L->agg, d.K, 0, wc, d.K, (int64_t)d.fo * d.K, 0, L->b_post, L->snorm, L->y, d.fo, 0, 0,
                                   reinterpret_cast<double*>(ws + z_b), bn_b, &slots, stream));
            // BatchNorm -> ReLU -> residual                                                     (:123-128 / :194-199)
            if (slots > 0) return bn_tail_forward_from_partials(d.N, d.fo, slots, reinterpret_cast<const double*>(ws + z_b), L->y, d.fo, L->bn_gamma, L->bn_beta,
                                                 L->running_mean, L->running_var, L->momentum, L->eps, 1, L->residual ? L->h : nullptr, L->out,
                                                 L->save_mean, L->save_invstd, L->num_batches_tracked, L->num_batches_tracked ? 1 : 0, stream);
        } else {
            DGN_TRY(dgn_dc_gemm(L->dc, d.K, d.fo, 1, L->agg, d.K, 0, wc, d.K, (int64_t)d.fo * d.K, 0, L->b_post, L->snorm, L->y, d.fo, 0, 0, stream));
        }
    } else {
        DGN_TRY(lin_fwd(d.N, d.K, d.n, L->agg, L->wf, nullptr, z, stream));
        // (round 6: BatchNorm's column sums of y ride in the combine pass where its slabs fit the statistics workspace -- no bn_stats launch)
        int slots = 0;
        const bool ride = option(OPT_BN_STATS_FUSED) && !L->n_valid;
        DGN_TRY(scale_combine_forward_stats(d.N, 1, d.S, d.fo, z, L->scale, L->b_post, L->snorm, L->y, d.fo, ride ? reinterpret_cast<double*>(ws + z_b) : nullptr,
                                            bn_b, &slots, stream));
        if (slots > 0) return bn_tail_forward_from_partials(d.N, d.fo, slots, reinterpret_cast<const double*>(ws + z_b), L->y, d.fo, L->bn_gamma, L->bn_beta,
                                                            L->running_mean, L->running_var, L->momentum, L->eps, 1, L->residual ? L->h : nullptr, L->out,
                                                            L->save_mean, L->save_invstd, L->num_batches_tracked, L->num_batches_tracked ? 1 : 0, stream);
    }
    // BatchNorm -> ReLU -> residual                                                             (:123-128 / :194-199)
    DGN_TRY(bn_tail_forward_nbt(d.N, d.fo, L->y, d.fo, L->bn_gamma, L->bn_beta, L->running_mean, L->running_var, L->momentum, L->eps, 1, 1,
                                L->residual ? L->h : nullptr, L->out, L->save_mean, L->save_invstd, ws + z_b, bn_b, L->n_valid, L->num_batches_tracked,
                                L->num_batches_tracked ? 1 : 0, stream));
    return DGN_OK;
}

extern "C" size_t dgn_dense_layer_backward_workspace_bytes(const DgnDenseLayer* L) {
    Dims d;
    if (!dims_of(L, d, "dgn_dense_layer_backward_workspace_bytes")) return 0;
    return bwd_scratch(L, d).total;
}

extern "C" int dgn_dense_layer_backward(const DgnDenseLayer* L, const DgnDenseGrads* G, void* stream) {
    const char* fn = "dgn_dense_layer_backward";
    Dims d;
    if (!dims_of(L, d, fn)) return DGN_ERR_INVALID;
    if (!G) { set_error("%s: null grads", fn); return DGN_ERR_INVALID; }
    if (d.N == 0) return DGN_OK;
    if (!G->g_out || !G->g_h || !G->g_w_post || !G->g_b_post || !G->g_gamma || !G->g_beta || (d.cx && !G->g_w_pre)) {
        set_error("%s: null gradient buffer", fn);
        return DGN_ERR_INVALID;
    }
    const BwdScratch s = bwd_scratch(L, d);
    if (!L->ws || L->ws_bytes < s.total) { set_error("%s: workspace too small (%zu < %zu)", fn, L->ws_bytes, s.total); return DGN_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(L->ws);
    auto f = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    float *g_z = f(s.g_z), *sums = f(s.sums), *g_agg = f(s.g_agg), *g_wf = f(s.g_wf), *g_pq = f(s.g_pq), *g_hp = f(s.g_hp), *g_hq = f(s.g_hq),
          *g_wsd = f(s.g_wsd), *g_bsd = f(s.g_bsd);
    const bool padded = d.Fp != d.F0;
    const float* hp = padded ? L->hp : L->h;
    const int hoff = d.cx ? d.F0 : 0;
    const int64_t ld_post = hoff + (int64_t)d.S * d.A * d.F0;
    const float* wft = L->wf + (size_t)d.n * d.K;
    // BatchNorm (+ ReLU): column sums + affine gradients; its input gradient is formed inside the combine backward
    DGN_TRY(dgn_bn_tail_backward(d.N, d.fo, G->g_out, L->y, d.fo, L->bn_gamma, L->bn_beta, L->save_mean, L->save_invstd, 1, nullptr, G->g_gamma, G->g_beta,
                                 sums, ws + s.bn_ws, dgn_bn_tail_workspace_bytes(d.N, d.fo), L->n_valid, stream));
    DgnBnGrad bn{};
    bn.g_out = G->g_out; bn.y = L->y; bn.ld = d.fo; bn.gamma = L->bn_gamma; bn.beta = L->bn_beta; bn.mean = L->save_mean; bn.invstd = L->save_invstd;
    bn.sums = sums; bn.relu = 1; bn.n_valid = L->n_valid;
    if (d.dc) {
        // g_t = snorm * (BatchNorm input gradient) [N, f_out]; per class: d agg = g_t W_class, G_class = g_t^T agg; g_wf = sum_class scale G_class
        const float* wct = L->wf + (size_t)2 * d.n * d.K + (size_t)DGN_DC_CLASSES * d.fo * d.K;
        DGN_TRY(scale_combine_backward_impl(d.N, 1, 1, d.fo, nullptr, 0, nullptr, L->snorm, g_z, G->g_b_post, ws + s.comb_ws,
                                           dgn_scale_combine_backward_workspace_bytes(d.N, 1, d.fo), &bn, stream, 1));
        static const int dc_stream = getenv("DGN_DC_STREAM") ? atoi(getenv("DGN_DC_STREAM")) : 0;      // (experiment: nontemporal d agg rows)
        DGN_TRY(dgn_dc_gemm(L->dc, d.fo, d.K, 1, g_z, d.fo, 0, wct, d.fo, (int64_t)d.fo * d.K, 0, nullptr, nullptr, g_agg, d.K, 0, dc_stream, stream));
        DgnDcLayout lay{};
        lay.n_agg = d.A; lay.f_pad = d.Fp; lay.f_in = d.F0; lay.h_off = hoff; lay.id_slot = d.cx ? L->id_slot : -1; lay.ld = ld_post;
        DGN_TRY(dgn_dc_wgrad(L->dc, d.S, d.K, d.fo, g_z, d.fo, L->agg, d.K, G->g_w_post, 0, &lay, ws + s.wg_ws,
                             dgn_dc_wgrad_workspace_bytes(L->dc->n_units, d.K, d.fo), stream));
    } else {
        // (one scaler: the bias gradient sum_m g_z[m, :] rides in the weight-gradient pass as its ones column -- no partial sums in the
        //  combine backward, no bias_finalize launch: round 6, the 19-launch CIFAR10 step)
        const bool bias_from_wgrad = d.S == 1;
        DGN_TRY(scale_combine_backward_impl(d.N, 1, d.S, d.fo, nullptr, 0, L->scale, L->snorm, g_z, bias_from_wgrad ? nullptr : G->g_b_post, ws + s.comb_ws,
                                           dgn_scale_combine_backward_workspace_bytes(d.N, 1, d.fo), &bn, stream, 1));
        // posttrans: input gradient (on the transposed folded weight), weight gradient, un-folded into the reference's layout
        DGN_TRY(lin_fwd(d.N, d.n, d.K, g_z, wft, nullptr, g_agg, stream));
        DGN_TRY(lin_wgrad(d.N, d.K, d.n, g_z, L->agg, g_wf, bias_from_wgrad ? G->g_b_post : nullptr, ws + s.wg_ws, wgrad_ws(d.N, d.K, d.n), stream));
    }
    if (!d.dc)
        hipLaunchKernelGGL(unfold_post, dim3(nblk((int64_t)d.fo * ld_post)), dim3(256), 0, st, d.S, d.fo, d.A, d.Ab, d.F0, d.Fp, hoff, d.cx ? L->id_slot : 0, ld_post,
                           g_wf, G->g_w_post);
    DGN_HIP_CHECK(hipGetLastError());
    // the sweep
    const DgnMsg msg = sweep_msg(L, d, hp);
    DgnMsgGrad gr{};
    if (d.cx) {
        gr.g_src = g_pq; gr.ld_src = 2 * d.Fp;
        gr.g_dst = g_pq + d.Fp; gr.ld_dst = 2 * d.Fp;
        gr.g_in = g_hp; gr.ld_in = d.Fp;
    } else {
        gr.g_src = g_hp; gr.ld_src = d.Fp;        // message = h[src], h_in = h: both gradients land in one buffer
        gr.g_in = g_hp; gr.ld_in = d.Fp;
    }
    gr.accumulate = 0;
    DGN_TRY(dgn_agg_backward_aux(L->graph, L->spec, &msg, L->w, L->ld_w, L->log_deg, g_agg, d.K, L->agg_aux, &gr, ws + s.agg_ws,
                                 dgn_agg_backward_workspace_bytes(L->graph, L->spec, d.Fp, 1), stream));
    const float* g_res = L->residual ? G->g_out : nullptr;
    bool fused_dh = false;
    if (d.cx) {
        // pretrans P|Q Linear: input gradient (transposed weight), weight + bias gradient, un-folded
        const float* wsdt = L->wsd + (size_t)2 * d.Fp * d.Fp;
        // even hidden size: d h = (d h_in + (d P|Q) W_sd) + residual leaves as the product's epilogue (unpad_add's order of additions)
        const bool al16 = ((reinterpret_cast<uintptr_t>(g_pq) | reinterpret_cast<uintptr_t>(g_hp) | reinterpret_cast<uintptr_t>(G->g_h) |
                            reinterpret_cast<uintptr_t>(g_res)) & 15) == 0;
        fused_dh = d.Fp == d.F0 && al16 && dgn_linear_supported(2 * d.Fp, d.Fp, 0) && dgn_linear_add_supported(2 * d.Fp, d.Fp);
        if (fused_dh) DGN_TRY(dgn_linear_forward_add(d.N, 2 * d.Fp, d.Fp, g_pq, wsdt, 2 * d.Fp, 0, g_hp, g_res, G->g_h, stream));
        else DGN_TRY(lin_fwd(d.N, 2 * d.Fp, d.Fp, g_pq, wsdt, nullptr, g_hq, stream));
        DGN_TRY(lin_wgrad(d.N, d.Fp, 2 * d.Fp, g_pq, hp, g_wsd, g_bsd, ws + s.wg_ws, wgrad_ws(d.N, d.Fp, 2 * d.Fp), stream));
        hipLaunchKernelGGL(unfold_pre, dim3(nblk((int64_t)2 * d.F0 * d.F0)), dim3(256), 0, st, d.F0, d.Fp, (int64_t)2 * d.F0, g_wsd, g_bsd, G->g_w_pre, G->g_b_pre);
    }
    // d h = d h_in (+ d x_src, same buffer) [+ (d P|Q) W_sd] [+ residual], un-padded
    if (fused_dh) return DGN_OK;
    if (d.F0 >= 4 && ((reinterpret_cast<uintptr_t>(G->g_h) | reinterpret_cast<uintptr_t>(g_res)) & 15) == 0)
        hipLaunchKernelGGL(unpad_add4, dim3(nblk((d.N * d.F0 + 3) / 4)), dim3(256), 0, st, d.N, d.F0, d.Fp, g_hp, d.cx ? g_hq : nullptr, g_res, G->g_h);
    else
        hipLaunchKernelGGL(unpad_add, dim3(nblk(d.N * d.F0)), dim3(256), 0, st, d.N, d.F0, d.Fp, g_hp, d.cx ? g_hq : nullptr, g_res, G->g_h);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
