// The DGN layer for batches at the reference's own batch size (128 graphs: configs/molecules_graph_regression_DGN_ZINC.json:12,
// superpixels_graph_classification_DGN_CIFAR10.json:12; main_molecules.py's hot loop): device code.
//
// At 3 000 - 15 000 rows every kernel of the streaming routes fills a fraction of the chip and a layer step is the sum of ~28 launch
// latencies.  Here the step is FIVE launches.  A batch of small graphs is block diagonal, so a workgroup that owns whole graphs (a
// "block": rows [lo, hi) with every source of every one of its rows inside, DgnBlockTable) can run the layer for them out of LDS:
//
//   blk_forward   (a workgroup per block)   h rows -> LDS; edge weights from eig (aggregators.py:35-71); P | Q = pretrans on
//                 [h_src || h_dst] decomposed (dgn_layer.py:75-80, :226-231); the aggregation (reduce_func, :86-98 / :161-173 / :237-249)
//                 with the messages formed from LDS rows; posttrans on [h || scaler x aggregator blocks] with the weight in the
//                 REFERENCE's layout (:116-119, :187-190, :266-271); graph norm; y0 -> memory, per-block BatchNorm partial sums.
//   blk_tail_fwd  (16 rows per wave)        BatchNorm statistics finalised by every workgroup from the partials (fixed order),
//                 then ReLU + residual (:121-130, :192-201), or -- towers -- the mixing network Linear -> LeakyReLU + residual (:318-324).
//   blk_tail_bwd  the adjoint of that tail: g_y1, the mixing network's weight-gradient partials, BatchNorm's two column sums.
//   blk_backward  (a workgroup per block)   recomputes the block's forward in LDS (nothing but y0 was saved), BatchNorm / graph-norm
//                 adjoint, posttrans input and weight gradients, the aggregation's adjoint: per-row coefficient vectors in LDS
//                 (make_coef_from: the arithmetic of the streaming backward), then every SOURCE row gathers the gradient rows of its
//                 out-edges in (source, slot) order -- no atomics, no [E, F] staging, run-to-run reproducible --, pretrans adjoint.
//   blk_reduce    parameter gradients = fixed-order sums of the per-workgroup partials.
//
// All products are v_mfma_f32_16x16x4_f32 (exact fp32) on 16-row strips with operands read from LDS (activations) and L2 (weights,
// reference layout, no fold / assembly launches).  The regime is latency-bound, not bandwidth-bound: the design minimises dependent
// round trips (one descriptor load, then every operand of the block in flight at once), not bytes.
#pragma once
#include "dgn_agg_kernels.hpp"

namespace dgn {
namespace blk {

using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int kMaxT = DGN_BLK_MAX_TOWERS;
constexpr int kMaxCh = 3;                       // edge-weight channels of one layer on this route
using C1 = Cfg<1, kMaxCh, true, true>;          // one feature per work item, every accumulator kind (LDS-resident operands: the unused ones cost nothing that matters)
constexpr int kTile = 16;

// coefficient slots of a destination row kept in LDS for the source-side gather (only those the aggregator list needs: `cmap`)
enum { CF_C0 = 0, CF_CV = 1, CF_GMAX = 2, CF_GMIN = 3, CF_ARG = 4, CF_CS0 = 5, CF_CA0 = 8, CF_SLOTS = 11 };

struct Layout {       // offsets (in floats) into the dynamic LDS of blk_forward / blk_backward
    int hb, pq, eig, ip, cp, cur, src, dst, csci, w, fac, agg, gy, y, coef, ga, gb, gc, red;
    int ld_agg;       // row stride of the aggregate chunk: T * K
    int ld_w;         // plane stride of the edge weights (max edges of a block)
    int n_coef;       // coefficient slots present
    int total;
};

struct P {
    AggParams a;                 // the aggregator list as the sweep's device code reads it (n_agg, op_pack, ch_pack, need, eps; ONE identity scaler)
    const int32_t* desc; int32_t n_blocks;
    const int32_t* indptr; const int32_t* src; const int32_t* csc_ptr; const int32_t* csc_pos;
    const float* eig; int32_t ld_eig; int32_t n_ch;
    int32_t ch_kind[kMaxCh], ch_col[kMaxCh]; float ch_alpha[kMaxCh], ch_eps[kMaxCh];
    const float* log_deg; const float* snorm;
    int32_t has_pre, relu, mixing, residual;
    int32_t T, fi, fo, F, Fo, A, S, K, h_off, ld_pre, ld_post;
    int32_t sc_kind[3]; float avg_log;
    const float* w_pre[kMaxT]; const float* b_pre[kMaxT]; const float* w_post[kMaxT]; const float* b_post[kMaxT];
    const float* gamma[kMaxT]; const float* beta[kMaxT];
    const float* w_mix; const float* b_mix; float slope;
    const float* h; float* y0; float* out;
    double* bn_part;             // [n_blocks][2][Fo]
    float* save_mean; float* save_invstd; float* running_mean; float* running_var; int64_t* nbt; int32_t n_nbt;
    float momentum, bn_eps;
    int64_t N;
    // tail
    int32_t tail_rows;           // rows per workgroup of the tail kernels (16 x waves)
    int32_t n_tail;
    // backward
    const float* g_out; float* g_y1; double* tail_part;      // [n_tail][2][Fo]
    float* tail_wpart;           // [n_tail][Fo * Fo + Fo]  (mixing network)
    float* blk_part;             // [n_blocks][n_blk_param]
    int32_t n_blk_param;         // floats of one block's parameter-gradient partial: per tower w_pre, b_pre, w_post, b_post
    int32_t off_tower;           // floats per tower in that layout
    float* g_h; float* g_gamma; float* g_beta;
    // LDS
    Layout L; int32_t R, RC, Emax;
    int8_t cmap[CF_SLOTS];       // coefficient slot -> index in the LDS coefficient rows, -1: absent
    // tests
    float* dbg_agg; float* dbg_w; float* dbg_gagg; int64_t* dbg_time;
};

// profiling: wall-clock stamp i of this workgroup (p.dbg_time == NULL: nothing)
#define BLK_STAMP(i) do { if (p.dbg_time && threadIdx.x == 0) p.dbg_time[(int64_t)blockIdx.x * 16 + (i)] = (int64_t)wall_clock64(); } while (0)

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[i][j] += sum_{k < K} A(i, k) B(k, j): lane (i16, g) supplies fa(k) = A(its i16, k) and fb(k) = B(k, its i16) for k = 16 b + 4 g + s;
// both must return 0 at k >= K.  The lane ends up with acc[4 g + s][i16], s = 0..3.
template <class FA, class FB>
__device__ __forceinline__ void tile_mma(f4& acc, int K, int g, FA&& fa, FB&& fb) {
    for (int b0 = 0; b0 < K; b0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = b0 + 4 * g + s;
            av[s] = fa(k);
            bv[s] = fb(k);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
    }
}

// The same with the A operand in GLOBAL memory (weights, L2-resident): element k of the lane's A row is arow[k * sk].  The loads of a
// batch of KB 16-k blocks are all issued -- unconditional, clamped -- before the first MFMA consumes one: a batch costs ONE memory round
// trip (a load whose value is selected or consumed at once is waited for at once: with one wave per SIMD that made every k step a
// full L2 latency, 85 us for a forward of 23 rows).
constexpr int kBatchKB = 8;
template <int KB = 2, class FB>
__device__ __forceinline__ void tile_mma_g(f4& acc, int K, int g, const float* arow, int64_t sk, bool aok, FB&& fb) {
    for (int c0 = 0; c0 < K; c0 += 16 * KB) {
        float av[KB][4];
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) av[b][s] = arow[(int64_t)min(c0 + 16 * b + 4 * g + s, K - 1) * sk];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            if (c0 + 16 * b < K) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = c0 + 16 * b + 4 * g + s;
                    acc = mfma4((aok && k < K) ? av[b][s] : 0.f, fb(k), acc);
                }
            }
        }
    }
}

// posttrans' input row is the virtual concatenation [h block (h_off columns) | scaler 0 x aggregates (K) | scaler 1 ... ]: position of a
// column in it, advanced in steps of 16 (no division per element)
struct VirtCol { int seg, rel; };      // seg 0: the h block, seg s + 1: scaler s
__device__ __forceinline__ VirtCol vcol_at(int k, int h_off, int K) {
    VirtCol v;
    if (k < h_off) { v.seg = 0; v.rel = k; return v; }
    v.seg = 1; v.rel = k - h_off;
    while (v.rel >= K) { v.rel -= K; ++v.seg; }
    return v;
}
__device__ __forceinline__ void vcol_step16(VirtCol& v, int h_off, int K) {
    v.rel += 16;
    int len = v.seg == 0 ? h_off : K;
    while (v.rel >= len) { v.rel -= len; ++v.seg; len = K; }
}

struct RowFeat { int r, f; };
__device__ __forceinline__ RowFeat rf_at(int idx, int F) { RowFeat x; x.r = idx / F; x.f = idx - x.r * F; return x; }
__device__ __forceinline__ void rf_step(RowFeat& x, int step, int F) { x.f += step; while (x.f >= F) { x.f -= F; ++x.r; } }
// a step given as (whole rows, remainder): one compare instead of a loop when the stride is many rows (1024 threads over 70 features)
struct RowStep { int dr, df; };
__device__ __forceinline__ RowStep rf_stride(int step, int F) { RowStep q; q.dr = step / F; q.df = step - q.dr * F; return q; }
__device__ __forceinline__ void rf_step(RowFeat& x, const RowStep& q, int F) { x.r += q.dr; x.f += q.df; if (x.f >= F) { x.f -= F; ++x.r; } }

__device__ __forceinline__ int tower_of(int f, int fi, int T) {
    int t = 0;
    for (int q = 1; q < T; ++q) t += (f >= q * fi) ? 1 : 0;
    return t;
}

struct Ctx {
    int lo, hi, e0, R, Eb;
    float *HB, *PQ, *EIG, *W, *FAC, *AGG, *GY, *Y, *COEF, *GA, *GB, *GC;
    int *IP, *CP, *CUR, *SRC, *DST, *CSCI;      // CUR [2][R]: first unprocessed (source, slot) rank of every source row, double-buffered per chunk
    double* RED;
};

__device__ __forceinline__ float weight_from_stats(int kind, float alpha, float eps, float d, float sabs, float spos, float sneg, float mx, float se) {
    if (kind == DGN_W_ABSNORM) return d / (sabs + eps);
    if (kind == DGN_W_BALANCED) return (fmaxf(d, 0.f) / (spos + eps) + fmaxf(-d, 0.f) / (sneg + eps)) / 2.f;
    return expf(alpha * fabsf(d) - mx) / se;
}

// descriptor -> every operand of the block in LDS: h rows, CSR rows re-based on the block, eig columns, scaler factors / graph norm,
// (backward) the transposed view; then the edge weights (aggregators.py:36-69) and P | Q.  Ends on a barrier.
__device__ __forceinline__ void block_prologue(const P& p, Ctx& c, float* lds, bool bwd) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int4 d = reinterpret_cast<const int4*>(p.desc)[blockIdx.x];
    c.lo = d.x; c.hi = d.y; c.e0 = d.z; c.R = d.y - d.x; c.Eb = d.w - d.z;
    const Layout& L = p.L;
    c.HB = lds + L.hb; c.PQ = lds + L.pq; c.EIG = lds + L.eig; c.W = lds + L.w; c.FAC = lds + L.fac; c.AGG = lds + L.agg;
    c.GY = lds + L.gy; c.Y = lds + L.y; c.COEF = lds + L.coef; c.GA = lds + L.ga; c.GB = lds + L.gb; c.GC = lds + L.gc;
    c.IP = reinterpret_cast<int*>(lds + L.ip); c.CP = reinterpret_cast<int*>(lds + L.cp); c.SRC = reinterpret_cast<int*>(lds + L.src);
    c.DST = reinterpret_cast<int*>(lds + L.dst); c.CSCI = reinterpret_cast<int*>(lds + L.csci); c.CUR = reinterpret_cast<int*>(lds + L.cur);
    c.RED = reinterpret_cast<double*>(lds + L.red);
    const int R = c.R, Eb = c.Eb, F = p.F;
    {   // (everything below is issued before anything is waited for: one memory round trip)
        const float* hrow = p.h + (int64_t)c.lo * F;
        for (int i = tid; i < R * F; i += NT) c.HB[i] = hrow[i];
        for (int i = tid; i <= R; i += NT) c.IP[i] = p.indptr[c.lo + i] - c.e0;
        for (int i = tid; i < Eb; i += NT) c.SRC[i] = p.src[c.e0 + i] - c.lo;
        for (int i = tid; i < R * p.n_ch; i += NT) {
            const int r = i / p.n_ch, ch = i - r * p.n_ch;
            c.EIG[i] = p.eig[(int64_t)(c.lo + r) * p.ld_eig + p.ch_col[ch]];
        }
        for (int i = tid; i < R; i += NT) {
            const int deg = p.indptr[c.lo + i + 1] - p.indptr[c.lo + i];
            const float logd = p.log_deg[c.lo + i];
            f4 fc;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int kind = s < p.S ? p.sc_kind[s] : DGN_SCALE_IDENTITY;
                // (rows without messages aggregate to zeros: keep avg / log(1) = inf away from 0 * inf)
                fc[s] = kind == DGN_SCALE_IDENTITY ? 1.f : (deg > 0 ? scaler_factor(kind, logd, p.avg_log) : 0.f);
            }
            fc[3] = p.snorm ? p.snorm[c.lo + i] : 1.f;
            *reinterpret_cast<f4*>(c.FAC + 4 * i) = fc;
        }
        if (bwd) {
            // closed block: the (source, slot) ranks of its slots are exactly [e0, e0 + Eb)
            for (int i = tid; i <= R; i += NT) {
                const int v = p.csc_ptr[c.lo + i] - c.e0;
                c.CP[i] = v;
                if (i < R) c.CUR[i] = v;
            }
            for (int i = tid; i < Eb; i += NT) c.CSCI[p.csc_pos[c.e0 + i] - c.e0] = i;
            for (int i = tid; i < R * F; i += NT) c.GA[i] = 0.f;
        }
    }
    __syncthreads();
    BLK_STAMP(1);
    for (int r = tid; r < R; r += NT)
        for (int j = c.IP[r]; j < c.IP[r + 1]; ++j) c.DST[j] = r;
    __syncthreads();
    BLK_STAMP(2);
    // delta_jc = eig[src_j, c] - eig[i, c], parked in the weight planes
    for (int i = tid; i < Eb * p.n_ch; i += NT) {
        const int ch = i / Eb, j = i - ch * Eb;
        c.W[ch * L.ld_w + j] = c.EIG[c.SRC[j] * p.n_ch + ch] - c.EIG[c.DST[j] * p.n_ch + ch];
    }
    __syncthreads();
    BLK_STAMP(3);
    // a thread per (row, channel): the row's normalisers in slot order, then its weights
    for (int i = tid; i < R * p.n_ch; i += NT) {
        const int r = i / p.n_ch, ch = i - r * p.n_ch;
        const int kind = p.ch_kind[ch];
        const float alpha = p.ch_alpha[ch], eps = p.ch_eps[ch];
        float* wp = c.W + ch * L.ld_w;
        float sabs = 0.f, spos = 0.f, sneg = 0.f, mx = -INFINITY, se = 0.f;
        const int beg = c.IP[r], end = c.IP[r + 1];
        for (int j = beg; j < end; ++j) {
            const float dl = wp[j];
            sabs += fabsf(dl); spos += fmaxf(dl, 0.f); sneg += fmaxf(-dl, 0.f);
            mx = fmaxf(mx, alpha * fabsf(dl));
        }
        if (kind == DGN_W_SOFTMAX)
            for (int j = beg; j < end; ++j) se += expf(alpha * fabsf(wp[j]) - mx);
        for (int j = beg; j < end; ++j) wp[j] = weight_from_stats(kind, alpha, eps, wp[j], sabs, spos, sneg, mx, se);
    }
    // P | Q = h [W_s | W_d]^T + [0 | b]: jobs of (16-row strip, tower, half, 16-column tile)
    if (p.has_pre) {
        const int lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
        const int nstrip = (R + 15) >> 4, ntq = (p.fi + 15) >> 4;
        const int njobs = nstrip * p.T * 2 * ntq;
        for (int job = wave; job < njobs; job += nw) {
            int q = job;
            const int tq = q % ntq; q /= ntq;
            const int half = q & 1; q >>= 1;
            const int t = q % p.T; const int strip = q / p.T;
            const int n = tq * 16 + i16, m = strip * 16 + i16;
            const float* wrow = p.w_pre[t] + (int64_t)min(n, p.fi - 1) * p.ld_pre + half * p.fi;
            const float* xrow = c.HB + min(m, R - 1) * F + t * p.fi;
            const bool nok = n < p.fi, mok = m < R;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma_g(acc, p.fi, g, wrow, 1, nok, [&](int k) { const float v = xrow[min(k, p.fi - 1)]; return (mok && k < p.fi) ? v : 0.f; });
            if (mok) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = tq * 16 + 4 * g + s;
                    if (o < p.fi) c.PQ[m * 2 * F + half * F + t * p.fi + o] = acc[s] + (half ? p.b_pre[t][o] : 0.f);
                }
            }
        }
    }
    __syncthreads();
    BLK_STAMP(4);
}

// message of slot j (source row s) into row r, feature f -- x_dst + x_src, the sweep's rounding order (load_msg)
__device__ __forceinline__ float msg_at(const P& p, const Ctx& c, int s, int r, int f) {
    return p.has_pre ? c.PQ[r * 2 * p.F + p.F + f] + c.PQ[s * 2 * p.F + f] : c.HB[s * p.F + f];
}

// the accumulators of row r, feature f, slots in ascending order (positions tracked as block-local slot ids)
template <bool TRACK>
__device__ __forceinline__ void accumulate_row(Acc<C1, TRACK>& acc, const P& p, const Ctx& c, int r, int f) {
    acc.init();
    const float q = p.has_pre ? c.PQ[r * 2 * p.F + p.F + f] : 0.f;
    const float* xs = p.has_pre ? c.PQ + f : c.HB + f;
    const int ldx = p.has_pre ? 2 * p.F : p.F;
    const int beg = c.IP[r], end = c.IP[r + 1];
    // four slots per step, every LDS operand of the group requested before the first is consumed (one slot at a time is a chain of
    // dependent LDS latencies: index -> source row -> accumulate; 8 us for ONE 8-edge row per thread)
    for (int j0 = beg; j0 < end; j0 += 4) {
        int sv[4];
        float xv[4], wv[4][kMaxCh];
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = c.SRC[min(j0 + u, end - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ch = 0; ch < kMaxCh; ++ch) wv[u][ch] = c.W[min(ch, max(p.n_ch - 1, 0)) * p.L.ld_w + min(j0 + u, end - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = xs[sv[u] * ldx];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u < end) {
                float m[1], wk[kMaxCh];
                m[0] = p.has_pre ? q + xv[u] : xv[u];
#pragma unroll
                for (int ch = 0; ch < kMaxCh; ++ch) wk[ch] = ch < p.n_ch ? wv[u][ch] : 0.f;
                acc.add(m, wk, j0 + u);
            }
        }
    }
}

// column of (aggregator a, feature f) in an aggregate row: [tower][aggregator][fi]
__device__ __forceinline__ int agg_col(const P& p, int t, int ft, int a) { return t * p.K + a * p.fi + ft; }

// ---- forward -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void blk_forward(const P p) {
    extern __shared__ float lds[];
    Ctx c;
    BLK_STAMP(0);
    block_prologue(p, c, lds, false);
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int R = c.R, F = p.F, Fo = p.Fo, RC = p.RC;
    double s0 = 0.0, s1 = 0.0;                                 // BatchNorm partial sums of column tid
    for (int c0 = 0; c0 < R; c0 += RC) {
        const int rc = min(RC, R - c0);
        // the aggregation: a work item per (row, feature)
        {
            RowFeat x = rf_at(tid, F);
            const RowStep st_ = rf_stride(NT, F);
            for (; x.r < rc; rf_step(x, st_, F)) {
                const int r = c0 + x.r, f = x.f;
                const int t = tower_of(f, p.fi, p.T), ft = f - t * p.fi;
                float* arow = c.AGG + x.r * p.L.ld_agg;
                const int deg = c.IP[r + 1] - c.IP[r];
                if (deg == 0) {
                    for (int a = 0; a < p.A; ++a) arow[agg_col(p, t, ft, a)] = 0.f;
                    continue;
                }
                Acc<C1, false> acc;
                accumulate_row<false>(acc, p, c, r, f);
                RowStats<1> st;
                row_stats<C1, false>(st, acc, (float)deg, p.a);
                const float xin[1] = {c.HB[r * F + f]};
                for (int a = 0; a < p.A; ++a) {
                    float val[1];
                    agg_value<C1, false>(val, agg_op(p.a, a), agg_ch(p.a, a), acc, st, xin);
                    arow[agg_col(p, t, ft, a)] = val[0];
                }
            }
        }
        __syncthreads();
        BLK_STAMP(5);
        if (p.dbg_agg)
            for (int i = tid; i < rc * p.L.ld_agg; i += NT) p.dbg_agg[(int64_t)(c.lo + c0) * p.L.ld_agg + i] = c.AGG[i];
        // posttrans([h || scaler x aggregate blocks]) (+ bias) * graph norm: jobs of (strip, tower, 16-column tile)
        {
            const int nstrip = (rc + 15) >> 4, ntq = (p.fo + 15) >> 4;
            const int njobs = nstrip * p.T * ntq;
            for (int job = wave; job < njobs; job += nw) {
                int q = job;
                const int tq = q % ntq; q /= ntq;
                const int t = q % p.T; const int strip = q / p.T;
                const int n = tq * 16 + i16, m = strip * 16 + i16;
                const bool nok = n < p.fo, mok = m < rc;
                const int mr = c0 + min(m, rc - 1);
                const float* wrow = p.w_post[t] + (int64_t)min(n, p.fo - 1) * p.ld_post;
                const float* xrow = c.HB + mr * F + t * p.fi;
                const float* arow = c.AGG + min(m, rc - 1) * p.L.ld_agg + t * p.K;
                const f4 fc = *reinterpret_cast<const f4*>(c.FAC + 4 * mr);
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                // ONE product over the weight row's ld_post columns, the input row [h | scale_s * aggregates] formed on the fly
                VirtCol vc[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) vc[s] = vcol_at(4 * g + s, p.h_off, p.K);
                for (int c1 = 0; c1 < p.ld_post; c1 += 16 * kBatchKB) {
                    float av[kBatchKB][4];
#pragma unroll
                    for (int b = 0; b < kBatchKB; ++b)
#pragma unroll
                        for (int s = 0; s < 4; ++s) av[b][s] = wrow[min(c1 + 16 * b + 4 * g + s, p.ld_post - 1)];
#pragma unroll
                    for (int b = 0; b < kBatchKB; ++b) {
                        if (c1 + 16 * b < p.ld_post) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const int k = c1 + 16 * b + 4 * g + s;
                                const bool kok = k < p.ld_post;
                                const int seg = vc[s].seg, rel = kok ? vc[s].rel : 0;
                                float xv = seg == 0 ? xrow[rel] : arow[rel] * (seg == 1 ? fc[0] : (seg == 2 ? fc[1] : fc[2]));
                                acc = mfma4((nok && kok) ? av[b][s] : 0.f, (mok && kok) ? xv : 0.f, acc);
                                vcol_step16(vc[s], p.h_off, p.K);
                            }
                        }
                    }
                }
                if (mok) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int o = tq * 16 + 4 * g + s;
                        if (o < p.fo) c.Y[m * Fo + t * p.fo + o] = (acc[s] + p.b_post[t][o]) * fc[3];
                    }
                }
            }
        }
        __syncthreads();
        BLK_STAMP(6);
        {
            float* yrow = p.y0 + (int64_t)(c.lo + c0) * Fo;
            for (int i = tid; i < rc * Fo; i += NT) yrow[i] = c.Y[i];
            if (tid < Fo)
                for (int m = 0; m < rc; ++m) {
                    const double v = (double)c.Y[m * Fo + tid];
                    s0 += v; s1 += v * v;
                }
        }
        __syncthreads();
        BLK_STAMP(7);
    }
    if (tid < Fo) {
        p.bn_part[((int64_t)blockIdx.x * 2 + 0) * Fo + tid] = s0;
        p.bn_part[((int64_t)blockIdx.x * 2 + 1) * Fo + tid] = s1;
    }
    BLK_STAMP(10);
}

// column sums of a [parts][2][Fo] table of doubles (a part = one row of 2 Fo doubles) in a fixed order: NT / Fo groups of threads (at most
// 16) take interleaved parts, a thread owns one 16-byte pair of the row, six loads in flight, and the groups are added in order.  Result
// in RED[0 .. 2 Fo) ([sum | second sum]); ends on a barrier.  RED: 2 * Fo * 17 doubles.
constexpr int kRedGroups = 16;
__device__ __forceinline__ void column_sums(const double* part, int parts, int Fo, double* RED) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int G = max(1, min(NT / Fo, kRedGroups));
    const int grp = tid / Fo, q = tid - grp * Fo;          // the pair (2 q, 2 q + 1) of the 2 Fo doubles
    if (grp < G) {
        double2 a = make_double2(0.0, 0.0);
        for (int b0 = grp; b0 < parts; b0 += 6 * G) {
            double2 v[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) v[u] = *reinterpret_cast<const double2*>(part + (int64_t)min(b0 + u * G, parts - 1) * 2 * Fo + 2 * q);
#pragma unroll
            for (int u = 0; u < 6; ++u)
                if (b0 + u * G < parts) { a.x += v[u].x; a.y += v[u].y; }
        }
        *reinterpret_cast<double2*>(RED + (grp + 1) * 2 * Fo + 2 * q) = a;
    }
    __syncthreads();
    if (tid < 2 * Fo) {
        double t = 0.0;
        for (int k = 0; k < G; ++k) t += RED[(k + 1) * 2 * Fo + tid];
        RED[tid] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ float col_param(const float* const (&ptrs)[kMaxT], int col, int fo, int T) {
    const int t = tower_of(col, fo, T);
    return ptrs[t][col - t * fo];
}

// ---- forward tail: BatchNorm (training statistics) -> ReLU -> + h   or   -> mixing Linear -> LeakyReLU -> + h ------------------------
// LDS: [mean | invstd | gamma | beta] (4 Fo floats), Y1 [rows][Fo], W_mix [Fo][Fo] (towers), then the reduction scratch (doubles)
__global__ __launch_bounds__(512) void blk_tail_fwd(const P p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int Fo = p.Fo, RW = p.tail_rows;
    float* MEAN = lds; float* INVSTD = lds + Fo; float* GAM = lds + 2 * Fo; float* BET = lds + 3 * Fo;
    float* Y1 = lds + 4 * Fo;
    float* WM = Y1 + RW * Fo;
    const int n_wm = p.mixing ? Fo * Fo : 0;
    double* RED = reinterpret_cast<double*>(lds + ((4 * Fo + RW * Fo + n_wm + 1) & ~1));
    const int64_t m0 = (int64_t)blockIdx.x * RW;
    const int rows = (int)min((int64_t)RW, p.N - m0);
    // this workgroup's rows and the mixing weight are requested before the statistics are summed
    for (int i = tid; i < RW * Fo; i += NT) Y1[i] = i < rows * Fo ? p.y0[m0 * Fo + i] : 0.f;
    for (int i = tid; i < n_wm; i += NT) WM[i] = p.w_mix[i];
    column_sums(p.bn_part, p.n_blocks, Fo, RED);
    if (tid < Fo) {
        const double n = (double)p.N;
        const double mu = RED[tid] / n;
        double m2 = RED[Fo + tid] - mu * RED[tid];
        if (m2 < 0.0) m2 = 0.0;
        const float mean = (float)mu, invstd = (float)(1.0 / sqrt(m2 / n + (double)p.bn_eps));
        MEAN[tid] = mean; INVSTD[tid] = invstd;
        GAM[tid] = col_param(p.gamma, tid, p.fo, p.T); BET[tid] = col_param(p.beta, tid, p.fo, p.T);
        if (blockIdx.x == 0) {
            p.save_mean[tid] = mean; p.save_invstd[tid] = invstd;
            const float unbiased = (float)(p.N > 1 ? m2 / (n - 1.0) : m2 / n);
            p.running_mean[tid] = (1.f - p.momentum) * p.running_mean[tid] + p.momentum * mean;
            p.running_var[tid] = (1.f - p.momentum) * p.running_var[tid] + p.momentum * unbiased;
            if (tid < p.n_nbt) p.nbt[tid] += 1;
        }
    }
    __syncthreads();
    {
        RowFeat x = rf_at(tid, Fo);
        for (int i = tid; i < RW * Fo; i += NT, rf_step(x, NT, Fo)) {
            const bool in = i < rows * Fo;
            float v = in ? (Y1[i] - MEAN[x.f]) * INVSTD[x.f] * GAM[x.f] + BET[x.f] : 0.f;
            if (!p.mixing) {
                if (p.relu) v = fmaxf(v, 0.f);
                if (in) p.out[m0 * Fo + i] = p.residual ? v + p.h[m0 * Fo + i] : v;
            } else {
                Y1[i] = v;
            }
        }
    }
    if (!p.mixing) return;
    __syncthreads();
    // out = LeakyReLU(y1 W_mix^T + b_mix) (+ h): jobs of (strip, 16-column tile)
    const int nstrip = RW >> 4, ntq = (Fo + 15) >> 4;
    for (int job = wave; job < nstrip * ntq; job += nw) {
        const int tq = job % ntq, strip = job / ntq;
        const int n = tq * 16 + i16, m = strip * 16 + i16;
        const bool nok = n < Fo;
        const float* wrow = WM + min(n, Fo - 1) * Fo;
        const float* xrow = Y1 + m * Fo;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        tile_mma(acc, Fo, g,
                 [&](int k) { const float v = wrow[min(k, Fo - 1)]; return (nok && k < Fo) ? v : 0.f; },
                 [&](int k) { const float v = xrow[min(k, Fo - 1)]; return k < Fo ? v : 0.f; });
        if (m < rows) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = tq * 16 + 4 * g + s;
                if (o < Fo) {
                    const float z = acc[s] + p.b_mix[o];
                    float v = z > 0.f ? z : z * p.slope;
                    if (p.residual) v += p.h[(m0 + m) * Fo + o];
                    p.out[(m0 + m) * Fo + o] = v;
                }
            }
        }
    }
}

// ---- backward tail: g_out -> g_y1 (the gradient at BatchNorm's output), BatchNorm's column sums, the mixing network's parameters --------
// LDS: [mean | invstd | gamma | beta], XH [RW][Fo] (normalised y0), Y1 [RW][Fo], GZ [RW][Fo], GY1 [RW][Fo], W_mix [Fo][Fo] (towers)
__global__ __launch_bounds__(512) void blk_tail_bwd(const P p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int Fo = p.Fo, RW = p.tail_rows;
    float* MEAN = lds; float* INVSTD = lds + Fo; float* GAM = lds + 2 * Fo; float* BET = lds + 3 * Fo;
    float* XH = lds + 4 * Fo; float* Y1 = XH + RW * Fo; float* GZ = Y1 + RW * Fo; float* GY1 = GZ + RW * Fo; float* WM = GY1 + RW * Fo;
    const int64_t m0 = (int64_t)blockIdx.x * RW;
    const int rows = (int)min((int64_t)RW, p.N - m0);
    if (tid < Fo) {
        MEAN[tid] = p.save_mean[tid]; INVSTD[tid] = p.save_invstd[tid];
        GAM[tid] = col_param(p.gamma, tid, p.fo, p.T); BET[tid] = col_param(p.beta, tid, p.fo, p.T);
    }
    for (int i = tid; i < RW * Fo; i += NT) {
        XH[i] = i < rows * Fo ? p.y0[m0 * Fo + i] : 0.f;
        GZ[i] = i < rows * Fo ? p.g_out[m0 * Fo + i] : 0.f;
    }
    if (p.mixing)
        for (int i = tid; i < Fo * Fo; i += NT) WM[i] = p.w_mix[i];
    __syncthreads();
    {
        RowFeat x = rf_at(tid, Fo);
        for (int i = tid; i < RW * Fo; i += NT, rf_step(x, NT, Fo)) {
            const bool in = i < rows * Fo;
            const float xh = in ? (XH[i] - MEAN[x.f]) * INVSTD[x.f] : 0.f;
            const float y1 = xh * GAM[x.f] + BET[x.f];
            XH[i] = xh;
            Y1[i] = in ? y1 : 0.f;
            if (!p.mixing) GY1[i] = (in && (!p.relu || y1 > 0.f)) ? GZ[i] : 0.f;      // ReLU: the gradient passes where BatchNorm's output is > 0
        }
    }
    __syncthreads();
    if (p.mixing) {
        const int nstrip = RW >> 4, ntq = (Fo + 15) >> 4;
        // g_z = g_out * LeakyReLU'(y1 W_mix^T + b_mix), in place over the staged g_out
        for (int job = wave; job < nstrip * ntq; job += nw) {
            const int tq = job % ntq, strip = job / ntq;
            const int n = tq * 16 + i16, m = strip * 16 + i16;
            const bool nok = n < Fo;
            const float* wrow = WM + min(n, Fo - 1) * Fo;
            const float* xrow = Y1 + m * Fo;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma(acc, Fo, g,
                     [&](int k) { const float v = wrow[min(k, Fo - 1)]; return (nok && k < Fo) ? v : 0.f; },
                     [&](int k) { const float v = xrow[min(k, Fo - 1)]; return k < Fo ? v : 0.f; });
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = tq * 16 + 4 * g + s;
                if (o < Fo) {
                    const float z = acc[s] + p.b_mix[o];
                    GZ[m * Fo + o] *= (z > 0.f ? 1.f : p.slope);
                }
            }
        }
        __syncthreads();
        // g_y1 = g_z W_mix
        for (int job = wave; job < nstrip * ntq; job += nw) {
            const int tq = job % ntq, strip = job / ntq;
            const int kk = tq * 16 + i16, m = strip * 16 + i16;
            const bool kok = kk < Fo;
            const float* wcol = WM + min(kk, Fo - 1);
            const float* grow = GZ + m * Fo;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma(acc, Fo, g,
                     [&](int n) { const float v = wcol[min(n, Fo - 1) * Fo]; return (kok && n < Fo) ? v : 0.f; },
                     [&](int n) { const float v = grow[min(n, Fo - 1)]; return n < Fo ? v : 0.f; });
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = tq * 16 + 4 * g + s;
                if (o < Fo) GY1[m * Fo + o] = acc[s];
            }
        }
        // d W_mix partial: D[n][k] = sum_m g_z[m][n] y1[m][k] over this workgroup's rows; d b_mix partial
        float* wpart = p.tail_wpart + (int64_t)blockIdx.x * (Fo * Fo + Fo);
        for (int job = wave; job < ntq * ntq; job += nw) {
            const int tk = job % ntq, tn = job / ntq;
            const int n = tn * 16 + i16, kk = tk * 16 + i16;
            const bool nok = n < Fo, kok = kk < Fo;
            const float* gcol = GZ + min(n, Fo - 1);
            const float* ycol = Y1 + min(kk, Fo - 1);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma(acc, RW, g,
                     [&](int m) { const float v = gcol[min(m, RW - 1) * Fo]; return (nok && m < RW) ? v : 0.f; },
                     [&](int m) { const float v = ycol[min(m, RW - 1) * Fo]; return (kok && m < RW) ? v : 0.f; });
            if (kok) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int nn = tn * 16 + 4 * g + s;
                    if (nn < Fo) wpart[nn * Fo + kk] = acc[s];
                }
            }
        }
        if (tid < Fo) {
            float a = 0.f;
            for (int m = 0; m < RW; ++m) a += GZ[m * Fo + tid];
            wpart[Fo * Fo + tid] = a;
        }
        __syncthreads();
    }
    for (int i = tid; i < rows * Fo; i += NT) p.g_y1[m0 * Fo + i] = GY1[i];
    if (tid < Fo) {
        double a0 = 0.0, a1 = 0.0;
        for (int m = 0; m < rows; ++m) {
            const double gv = (double)GY1[m * Fo + tid];
            a0 += gv; a1 += gv * (double)XH[m * Fo + tid];
        }
        p.tail_part[((int64_t)blockIdx.x * 2 + 0) * Fo + tid] = a0;
        p.tail_part[((int64_t)blockIdx.x * 2 + 1) * Fo + tid] = a1;
    }
}

// d m_j of slot j (destination row at chunk row mc, source u) at feature f, from the destination's coefficient rows
__device__ __forceinline__ float edge_grad(const P& p, const Ctx& c, int mc, int j, int f, float m_j) {
    const float* cf = c.COEF + (mc * p.L.n_coef) * p.F + f;
    float gm = p.cmap[CF_C0] >= 0 ? cf[p.cmap[CF_C0] * p.F] : 0.f;
    if (p.cmap[CF_CV] >= 0) gm = fmaf(cf[p.cmap[CF_CV] * p.F], m_j, gm);
#pragma unroll
    for (int ch = 0; ch < kMaxCh; ++ch) {
        if (ch < p.n_ch) {
            const float w = c.W[ch * p.L.ld_w + j];
            if (p.cmap[CF_CS0 + ch] >= 0) gm = fmaf(w, cf[p.cmap[CF_CS0 + ch] * p.F], gm);
            if (p.cmap[CF_CA0 + ch] >= 0) gm = fmaf(fabsf(w), cf[p.cmap[CF_CA0 + ch] * p.F], gm);
        }
    }
    if (p.cmap[CF_ARG] >= 0) {
        const unsigned arg = __float_as_uint(cf[p.cmap[CF_ARG] * p.F]);  // (amax + 1) | (amin + 1) << 16, block-local slot ids
        if ((arg & 0xffffu) == (unsigned)(j + 1) && p.cmap[CF_GMAX] >= 0) gm += cf[p.cmap[CF_GMAX] * p.F];
        if ((arg >> 16) == (unsigned)(j + 1) && p.cmap[CF_GMIN] >= 0) gm += cf[p.cmap[CF_GMIN] * p.F];
    }
    return gm;
}

// ---- backward of a block ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void blk_backward(const P p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int F = p.F, Fo = p.Fo, RC = p.RC;
    Ctx c;
    BLK_STAMP(0);
    {   // BatchNorm's column sums (= d beta, d gamma) from the tail's partials; requested first: they land during the prologue
        double* RED = reinterpret_cast<double*>(lds + p.L.red);
        column_sums(p.tail_part, p.n_tail, Fo, RED);
        if (blockIdx.x == 0 && tid < Fo) { p.g_beta[tid] = (float)RED[tid]; p.g_gamma[tid] = (float)RED[Fo + tid]; }
    }
    block_prologue(p, c, lds, true);
    const int R = c.R;
    float* SUMS = reinterpret_cast<float*>(c.RED + 2 * Fo);      // [mean | invstd | gamma | beta | sum g / N | sum g xhat / N] as floats
    if (tid < Fo) {
        const float inv_n = 1.f / (float)p.N;
        SUMS[tid] = p.save_mean[tid]; SUMS[Fo + tid] = p.save_invstd[tid];
        SUMS[2 * Fo + tid] = col_param(p.gamma, tid, p.fo, p.T); SUMS[3 * Fo + tid] = col_param(p.beta, tid, p.fo, p.T);
        SUMS[4 * Fo + tid] = (float)c.RED[tid] * inv_n; SUMS[5 * Fo + tid] = (float)c.RED[Fo + tid] * inv_n;
    }
    __syncthreads();
    BLK_STAMP(5);
    float* bpart = p.blk_part + (int64_t)blockIdx.x * p.n_blk_param;
    const int off_wpost = p.has_pre ? p.fi * p.ld_pre + p.fi : 0, off_bpost = off_wpost + p.fo * p.ld_post;
    float gb_post = 0.f;                                       // d b_post of column tid
    for (int c0 = 0; c0 < R; c0 += RC) {
        const int rc = min(RC, R - c0);
        // g_yr = snorm * BatchNorm'(g_y1): the gradient at posttrans' output (bias included)
        {
            RowFeat x = rf_at(tid, Fo);
            const int64_t base = (int64_t)(c.lo + c0) * Fo;
            for (int i = tid; i < RC * Fo; i += NT, rf_step(x, NT, Fo)) {
                float v = 0.f;
                if (i < rc * Fo) {
                    const float xh = (p.y0[base + i] - SUMS[x.f]) * SUMS[Fo + x.f];
                    v = SUMS[2 * Fo + x.f] * SUMS[Fo + x.f] * (p.g_y1[base + i] - SUMS[4 * Fo + x.f] - xh * SUMS[5 * Fo + x.f]);
                    v *= c.FAC[4 * (c0 + x.r) + 3];
                }
                c.GY[i] = v;
            }
        }
        __syncthreads();
        BLK_STAMP(6);
        if (tid < Fo)
            for (int m = 0; m < rc; ++m) gb_post += c.GY[m * Fo + tid];
        // d aggregate rows = sum_s scale_s (g_yr W_post[:, block s]); d h through posttrans' h block
        {
            const int nstrip = (rc + 15) >> 4, ntk = (p.K + 15) >> 4, nth = p.has_pre ? (p.fi + 15) >> 4 : 0;
            const int njobs = nstrip * p.T * (ntk + nth);
            for (int job = wave; job < njobs; job += nw) {
                int q = job;
                const int tk = q % (ntk + nth); q /= (ntk + nth);
                const int t = q % p.T; const int strip = q / p.T;
                const int m = strip * 16 + i16;
                const bool mok = m < rc;
                const float* grow = c.GY + min(m, rc - 1) * Fo + t * p.fo;
                const f4 fc = *reinterpret_cast<const f4*>(c.FAC + 4 * (c0 + min(m, rc - 1)));
                if (tk < ntk) {
                    const int kk = tk * 16 + i16;
                    const bool kok = kk < p.K;
                    f4 out = {0.f, 0.f, 0.f, 0.f};
                    const float* wcol = p.w_post[t] + p.h_off + min(kk, p.K - 1);
                    f4 acc3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                    for (int n0 = 0; n0 < p.fo; n0 += 16) {      // (the three scalers' columns of a 16-n block are requested together)
                        float av[3][4], gv[4];
#pragma unroll
                        for (int s = 0; s < 3; ++s)
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                av[s][u] = wcol[(int64_t)min(n0 + 4 * g + u, p.fo - 1) * p.ld_post + min(s, p.S - 1) * p.K];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int n = n0 + 4 * g + u;
                            const float v = grow[min(n, p.fo - 1)];
                            gv[u] = (mok && n < p.fo) ? v : 0.f;
                        }
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            if (s < p.S) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) acc3[s] = mfma4((kok && n0 + 4 * g + u < p.fo) ? av[s][u] : 0.f, gv[u], acc3[s]);
                            }
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        if (s < p.S) {
                            const float sc = s == 0 ? fc[0] : (s == 1 ? fc[1] : fc[2]);
#pragma unroll
                            for (int u = 0; u < 4; ++u) out[u] += sc * acc3[s][u];
                        }
                    }
                    if (mok) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k2 = tk * 16 + 4 * g + u;
                            if (k2 < p.K) c.AGG[m * p.L.ld_agg + t * p.K + k2] = out[u];
                        }
                    }
                } else {
                    const int th = tk - ntk;
                    const int kk = th * 16 + i16;
                    const bool kok = kk < p.fi;
                    const float* wcol = p.w_post[t] + min(kk, p.fi - 1);
                    f4 acc = {0.f, 0.f, 0.f, 0.f};
                    tile_mma_g(acc, p.fo, g, wcol, p.ld_post, kok, [&](int n) { const float v = grow[min(n, p.fo - 1)]; return (mok && n < p.fo) ? v : 0.f; });
                    if (mok) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k2 = th * 16 + 4 * g + u;
                            if (k2 < p.fi) c.GC[(c0 + m) * F + t * p.fi + k2] = acc[u];
                        }
                    }
                }
            }
        }
        __syncthreads();
        BLK_STAMP(7);
        if (p.dbg_gagg)
            for (int i = tid; i < rc * p.L.ld_agg; i += NT) p.dbg_gagg[(int64_t)(c.lo + c0) * p.L.ld_agg + i] = c.AGG[i];
        // recompute the rows' accumulators (first-occurrence arg tracking), coefficient rows -> LDS; the aggregate values take the
        // place of their own upstream gradients (a (row, aggregator, feature) entry belongs to exactly one work item)
        {
            RowFeat x = rf_at(tid, F);
            const RowStep st_ = rf_stride(NT, F);
            for (; x.r < rc; rf_step(x, st_, F)) {
                const int r = c0 + x.r, f = x.f;
                const int t = tower_of(f, p.fi, p.T), ft = f - t * p.fi;
                float* arow = c.AGG + x.r * p.L.ld_agg;
                float* cf = c.COEF + (x.r * p.L.n_coef) * F + f;
                const int deg = c.IP[r + 1] - c.IP[r];
                if (deg == 0) {
                    for (int a = 0; a < p.A; ++a) arow[agg_col(p, t, ft, a)] = 0.f;
                    for (int q = 0; q < p.L.n_coef; ++q) cf[q * F] = 0.f;
                    continue;
                }
                Acc<C1, true> acc;
                accumulate_row<true>(acc, p, c, r, f);
                const float xin[1] = {c.HB[r * F + f]};
                Coef<C1> k;
                float gxin[1];
                make_coef_from<C1, DynOps>(k, gxin, acc, p.a, [&](int a, int, float (&gv)[1]) { gv[0] = arow[agg_col(p, t, ft, a)]; },
                                           deg, xin, 0.f);
                RowStats<1> st;
                row_stats<C1, true>(st, acc, (float)deg, p.a);
                for (int a = 0; a < p.A; ++a) {
                    float val[1];
                    agg_value<C1, true>(val, agg_op(p.a, a), agg_ch(p.a, a), acc, st, xin);
                    arow[agg_col(p, t, ft, a)] = val[0];
                }
                if (p.cmap[CF_C0] >= 0) cf[p.cmap[CF_C0] * F] = k.c0[0];
                if (p.cmap[CF_CV] >= 0) cf[p.cmap[CF_CV] * F] = k.cv[0];
                if (p.cmap[CF_GMAX] >= 0) cf[p.cmap[CF_GMAX] * F] = k.gmax[0];
                if (p.cmap[CF_GMIN] >= 0) cf[p.cmap[CF_GMIN] * F] = k.gmin[0];
                if (p.cmap[CF_ARG] >= 0) cf[p.cmap[CF_ARG] * F] = __uint_as_float((unsigned)(k.amax[0] + 1) | ((unsigned)(k.amin[0] + 1) << 16));
#pragma unroll
                for (int ch = 0; ch < kMaxCh; ++ch) {
                    if (p.cmap[CF_CS0 + ch] >= 0) cf[p.cmap[CF_CS0 + ch] * F] = k.cs[ch][0];
                    if (p.cmap[CF_CA0 + ch] >= 0) cf[p.cmap[CF_CA0 + ch] * F] = k.ca[ch][0];
                }
                // d x_in of the dx aggregators (x_in = this layer's / tower's input row)
                if (p.has_pre) c.GC[r * F + f] += gxin[0];
                else c.GA[r * F + f] += gxin[0];
            }
        }
        __syncthreads();
        BLK_STAMP(8);
        // d W_post partial: D[n][kk] (+)= sum over the chunk's rows of g_yr[m][n] * [h | scale_s * aggregate][m][kk]
        {
            const int ntn = (p.fo + 15) >> 4, ntk = (p.ld_post + 15) >> 4;
            const int njobs = p.T * ntn * ntk;
            for (int job = wave; job < njobs; job += nw) {
                int q = job;
                const int tk = q % ntk; q /= ntk;
                const int tn = q % ntn; const int t = q / ntn;
                const int n = tn * 16 + i16, kk = tk * 16 + i16;
                const bool nok = n < p.fo, kok = kk < p.ld_post;
                const float* gcol = c.GY + t * p.fo + min(n, p.fo - 1);
                // column kk of [h | s-blocks]: its LDS column and scaler
                const bool in_h = kk < p.h_off;
                int s = 0, k2 = 0;
                if (!in_h && kok) { s = (kk - p.h_off) / p.K; k2 = (kk - p.h_off) - s * p.K; }
                const float* xcol = in_h ? c.HB + c0 * F + t * p.fi + min(kk, p.fi - 1) : c.AGG + t * p.K + k2;
                const int ldx = in_h ? F : p.L.ld_agg;
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                tile_mma(acc, rc, g,
                         [&](int m) { const float v = gcol[min(m, rc - 1) * Fo]; return (nok && m < rc) ? v : 0.f; },
                         [&](int m) {
                             const int mm = min(m, rc - 1);
                             float v = xcol[mm * ldx];
                             if (!in_h) v *= c.FAC[4 * (c0 + mm) + s];
                             return (kok && m < rc) ? v : 0.f;
                         });
                if (kok) {
                    float* dst = bpart + t * p.off_tower + off_wpost + kk;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int nn = tn * 16 + 4 * g + u;
                        if (nn < p.fo) {
                            float* at = dst + nn * p.ld_post;
                            *at = c0 == 0 ? acc[u] : *at + acc[u];
                        }
                    }
                }
            }
        }
        // every source row gathers the gradient rows of its out-edges whose destination lies in this chunk, (source, slot) order
        {
            RowFeat x = rf_at(tid, F);
            const RowStep st_ = rf_stride(NT, F);
            for (; x.r < R; rf_step(x, st_, F)) {
                const int u = x.r, f = x.f;
                const int* cur = c.CUR + (((c0 / RC) & 1) ? R : 0);
                float a = 0.f;
                int rank = cur[u];
                for (const int end = c.CP[u + 1]; rank < end; ++rank) {      // (a source's slots ascend with their destinations)
                    const int j = c.CSCI[rank], i = c.DST[j];
                    if (i >= c0 + rc) break;
                    a += edge_grad(p, c, i - c0, j, f, msg_at(p, c, u, i, f));
                }
                if (f == 0) c.CUR[(((c0 / RC) & 1) ? 0 : R) + u] = rank;
                c.GA[u * F + f] += a;
            }
        }
        // d Q: the row sums of the same gradient rows, by the destination's work item
        if (p.has_pre) {
            RowFeat x = rf_at(tid, F);
            const RowStep st_ = rf_stride(NT, F);
            for (; x.r < rc; rf_step(x, st_, F)) {
                const int r = c0 + x.r, f = x.f;
                float a = 0.f;
                for (int j = c.IP[r]; j < c.IP[r + 1]; ++j) a += edge_grad(p, c, x.r, j, f, msg_at(p, c, c.SRC[j], r, f));
                c.GB[r * F + f] = a;
            }
        }
        __syncthreads();
        BLK_STAMP(9);
    }
    if (tid < Fo) {
        const int t = tower_of(tid, p.fo, p.T);
        bpart[t * p.off_tower + off_bpost + (tid - t * p.fo)] = gb_post;
    }
    BLK_STAMP(10);
    if (!p.has_pre) {
        // simple layer: x_src = x_in = h: d h = d x_src + d x_in (both in GA) + the residual's share
        const int64_t base = (int64_t)c.lo * F;
        for (int i = tid; i < R * F; i += NT) p.g_h[base + i] = c.GA[i] + (p.residual ? p.g_out[base + i] : 0.f);
        return;
    }
    // pretrans adjoint: d h = d P W_s + d Q W_d (+ posttrans' h block, d x_in, residual); d W_pre, d b_pre partials
    {
        const int nstrip = (R + 15) >> 4, nti = (p.fi + 15) >> 4;
        for (int job = wave; job < nstrip * p.T * nti; job += nw) {
            int q = job;
            const int ti = q % nti; q /= nti;
            const int t = q % p.T; const int strip = q / p.T;
            const int ii = ti * 16 + i16, m = strip * 16 + i16;
            const bool iok = ii < p.fi, mok = m < R;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int half = 0; half < 2; ++half) {
                const float* wcol = p.w_pre[t] + half * p.fi + min(ii, p.fi - 1);
                const float* grow = (half ? c.GB : c.GA) + min(m, R - 1) * F + t * p.fi;
                tile_mma_g(acc, p.fi, g, wcol, p.ld_pre, iok, [&](int o) { const float v = grow[min(o, p.fi - 1)]; return (mok && o < p.fi) ? v : 0.f; });
            }
            if (mok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i2 = ti * 16 + 4 * g + u;
                    if (i2 < p.fi) {
                        const int64_t at = (int64_t)(c.lo + m) * F + t * p.fi + i2;
                        p.g_h[at] = acc[u] + c.GC[m * F + t * p.fi + i2] + (p.residual ? p.g_out[at] : 0.f);
                    }
                }
            }
        }
        const int nto = (p.fi + 15) >> 4;
        for (int job = wave; job < p.T * 2 * nto * nti; job += nw) {
            int q = job;
            const int tc = q % nti; q /= nti;
            const int to = q % nto; q /= nto;
            const int half = q & 1; const int t = q >> 1;
            const int o = to * 16 + i16, cc = tc * 16 + i16;
            const bool ook = o < p.fi, cok = cc < p.fi;
            const float* gcol = (half ? c.GB : c.GA) + t * p.fi + min(o, p.fi - 1);
            const float* hcol = c.HB + t * p.fi + min(cc, p.fi - 1);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma(acc, R, g,
                     [&](int m) { const float v = gcol[min(m, R - 1) * F]; return (ook && m < R) ? v : 0.f; },
                     [&](int m) { const float v = hcol[min(m, R - 1) * F]; return (cok && m < R) ? v : 0.f; });
            if (cok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int oo = to * 16 + 4 * g + u;
                    if (oo < p.fi) bpart[t * p.off_tower + oo * p.ld_pre + half * p.fi + cc] = acc[u];
                }
            }
        }
        if (tid < F) {
            const int t = tower_of(tid, p.fi, p.T);
            float a = 0.f;
            for (int m = 0; m < R; ++m) a += c.GB[m * F + tid];
            bpart[t * p.off_tower + p.fi * p.ld_pre + (tid - t * p.fi)] = a;
        }
    }
    BLK_STAMP(11);
}

// out[i] = sum over parts of part[q][i]: eight lanes per output take interleaved parts (all of a lane's loads in flight: the partials are
// L2 / MALL resident, the kernel is latency-bound), added across the lanes in a fixed order; two segments (block partials, tail partials)
// behind each other in `out`
__global__ __launch_bounds__(256) void blk_reduce(const float* __restrict__ part_a, int n_a, int parts_a, const float* __restrict__ part_b, int n_b,
                                                  int parts_b, float* __restrict__ out) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gt >> 3, sub = gt & 7;
    const bool live = i < n_a + n_b;
    const int ii = live ? i : 0;
    const bool a = ii < n_a;
    const float* src = a ? part_a + ii : part_b + (ii - n_a);
    const int stride = a ? n_a : n_b, parts = a ? parts_a : parts_b;
    float acc = 0.f;
    for (int q0 = sub; q0 < parts; q0 += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)min(q0 + 8 * u, parts - 1) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (q0 + 8 * u < parts) ? v[u] : 0.f;
    }
    acc += __shfl_xor(acc, 1, kWave);
    acc += __shfl_xor(acc, 2, kWave);
    acc += __shfl_xor(acc, 4, kWave);
    if (live && sub == 0) out[i] = acc;
}

}  // namespace blk
}  // namespace dgn
