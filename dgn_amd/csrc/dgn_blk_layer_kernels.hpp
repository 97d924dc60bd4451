// The DGN layer for batches at the reference's own batch size (128 graphs: configs/molecules_graph_regression_DGN_ZINC.json:12,
// superpixels_graph_classification_DGN_CIFAR10.json:12; main_molecules.py's hot loop): device code.
//
// At 3 000 - 15 000 rows every kernel of the streaming routes fills a fraction of the chip and a layer step is the sum of ~28 launch
// latencies.  Here the step is FIVE launches.  A batch of small graphs is block diagonal, so a workgroup that owns whole graphs (a
// "block": rows [lo, hi) with every source of every one of its rows inside, DgnBlockTable) can run the layer for them out of LDS:
//
//   blk_forward   (a workgroup per block and tower)   h rows -> LDS; edge weights from eig (aggregators.py:35-71); P | Q = pretrans on
//                 [h_src || h_dst] decomposed (dgn_layer.py:75-80, :226-231); the aggregation (reduce_func, :86-98 / :161-173 / :237-249)
//                 with the messages formed from LDS rows, leaving posttrans' input row [h | scaler x aggregator blocks] in LDS; posttrans
//                 with the weight in the REFERENCE's layout (:116-119, :187-190, :266-271); graph norm; y0 -> memory, BatchNorm partials.
//   blk_tail_fwd  (16 rows per wave)        BatchNorm statistics finalised by every workgroup from the partials (fixed order),
//                 then ReLU + residual (:121-130, :192-201), or -- towers -- the mixing network Linear -> LeakyReLU + residual (:318-324).
//   blk_tail_bwd  the adjoint of that tail: g_y1, the mixing network's weight-gradient partials, BatchNorm's two column sums.
//   blk_backward  (a workgroup per block and tower)   recomputes the block's forward in LDS (nothing but y0 was saved), BatchNorm /
//                 graph-norm adjoint, posttrans input and weight gradients, the aggregation's adjoint: per-row coefficient vectors in LDS
//                 (make_coef_from: the arithmetic of the streaming backward), then every SOURCE row gathers the gradient rows of its
//                 out-edges in (source, slot) order -- no atomics, no [E, F] staging, run-to-run reproducible --, pretrans adjoint.
//   blk_reduce    parameter gradients = fixed-order sums of the per-workgroup partials.
//
// The regime is latency-bound, and -- measured with wall-clock stamps per phase (BLK_STAMP, tools/blk_phases.py) -- INSTRUCTION-bound: a
// workgroup passes through every line of the kernel once, sixteen waves share four SIMDs, and the first version's generic, masked,
// fully unrolled product loops cost ~7 ns per ISA line (sweep 8 us, one posttrans tile job 22 us).  Hence: products are mask-free
// 16-k blocks (one ds_read_b128 + four dword loads + four v_mfma_f32_16x16x4_f32, masks only in the K tail), the hot aggregator lists of
// the reference's configs are baked in (StaticOps, dgn_agg_hot.hpp), and the towers layer runs one workgroup per (block, tower): towers
// are independent up to BatchNorm, and five times as many, five times smaller workgroups use every CU.
#pragma once
#include "dgn_agg_kernels.hpp"

namespace dgn {
namespace blk {

using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int kMaxT = DGN_BLK_MAX_TOWERS;
constexpr int kMaxCh = 3;                       // edge-weight channels of one layer on this route
constexpr int kRedGroups = 16;

// coefficient slots of a destination row kept in LDS for the source-side gather (only those the aggregator list needs: `cmap`)
enum { CF_C0 = 0, CF_CV = 1, CF_GMAX = 2, CF_GMIN = 3, CF_ARG = 4, CF_CS0 = 5, CF_CA0 = 8, CF_SLOTS = 11 };

struct Layout {       // offsets (in floats) into the dynamic LDS of blk_forward / blk_backward; row strides are multiples of 4 floats
    int hb, pq, eig, ip, cp, cur, src, dst, csci, w, fac, xp, y, y0s, coef, ga, gb, gc, vec, red;
    int ldh;          // row stride of the h rows (up4(f_in)); P | Q rows are 2 ldh wide
    int ldy;          // row stride of the posttrans output / its gradient rows (up4(f_out))
    int kp;           // row stride of the rows [h (up4(f_in), complex / towers) | aggregator blocks (up4(K))] posttrans' input is formed from
    int ho;           // where the aggregator blocks start in such a row
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
    int32_t has_pre, relu, mixing, residual, eval_mode;
    int32_t T, fi, fo, F, Fo, A, S, K, h_off, ld_pre, ld_post;      // fi / fo: per tower; F = T fi, Fo = T fo: row strides of h / y0
    int32_t sc_kind[3]; float avg_log;
    const float* w_pre[kMaxT]; const float* b_pre[kMaxT]; const float* w_post[kMaxT]; const float* b_post[kMaxT];
    const float* gamma[kMaxT]; const float* beta[kMaxT];
    const float* w_mix; const float* b_mix; float slope;
    const float* h; float* y0; float* out;
    double* bn_part;             // [n_blocks][2][Fo]
    float* save_mean; float* save_invstd; float* running_mean; float* running_var; int64_t* nbt; int32_t n_nbt;
    float momentum, bn_eps;
    int64_t N;
    // towers' F.dropout between BatchNorm and the mixing network (nets/dgn_layer.py:275), inside the tails (round 6): drop_scale == 0: none.
    // Simple / complex layers (!mixing): their F.dropout is the layer's LAST op (:130, :201) -- on the finished output rows in the forward tail; the
    // backward tail masks the staged output gradient, blk_backward the residual's share of d h.
    // Philox keep bits as dgn_dropout_forward draws them (group g = 8 consecutive elements of the dense [N, Fo] tensor, one mask byte)
    float drop_scale; uint32_t drop_threshold; const int64_t* drop_seed; uint64_t drop_offset; unsigned char* drop_mask;
    const int64_t* n_valid;      // DEVICE: rows of the batch inside a buffer of N rows (padded batches, hipgraph.PaddedBatch); NULL: N
    int32_t* overflow;           // DEVICE, may be NULL: set to 1 by a block larger than the LDS plan (padded batches: the table changes per batch)
    // tail
    int32_t tail_rows;           // rows per workgroup of the tail kernels
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
    // tests / profiling
    float* dbg_agg; float* dbg_gagg; int64_t* dbg_time;
};

// profiling: wall-clock stamp i of this workgroup (p.dbg_time == NULL: nothing)
#define BLK_STAMP(i) do { if (p.dbg_time && threadIdx.x == 0) p.dbg_time[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = (int64_t)wall_clock64(); } while (0)

// rows of the batch (padded batches: a device scalar)
__device__ __forceinline__ int64_t valid_rows(const P& p) { return p.n_valid ? min(*p.n_valid, p.N) : p.N; }
// a block's descriptor: first row, end row, first slot, end slot (negative slots: taken from the row pointers -- padded batches, whose
// table is written by the host from the graph sizes alone); an EMPTY or over-sized block is skipped by its workgroups
// `skipped_end`: the end row of a block that was skipped because it is OVER-SIZED (else its first row: nothing skipped) -- the rows
// [d.x, skipped_end) are valid rows of the batch that no workgroup computes: the caller writes zeros for them (y0 in the forward, d h in
// the backward), so that the tails, the BatchNorm statistics and a captured optimizer step read defined values until
// DGNGraph.check_deferred raises (they read uninitialised memory before).
__device__ __forceinline__ int4 block_desc(const P& p, int& skipped_end) {
    int4 d = reinterpret_cast<const int4*>(p.desc)[blockIdx.x];
    if (d.z < 0 && d.y > d.x) { d.z = p.indptr[d.x]; d.w = p.indptr[d.y]; }
    skipped_end = d.x;
    if (d.y - d.x > p.R || d.w - d.z > p.Emax) {
        if (p.overflow && threadIdx.x == 0) *p.overflow = 1;
        skipped_end = (int)min((int64_t)d.y, p.N);
        d.y = d.x;
    }
    return d;
}

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- 16 x 16 tile products on v_mfma_f32_16x16x4_f32 ---------------------------------------------------------------------------------
// acc[i][j] += sum_k A(i, k) B(k, j); lane (i16, g) supplies A(its i16, k) and B(k, its i16) at k = 16 b + 4 g + s and ends up with
// acc[4 g + s][i16], s = 0..3.  Full 16-k blocks are mask-free; only the K tail selects.  Rows past a tile's valid range are CLAMPED by
// the caller (their results are discarded, a product never mixes rows), so nothing else needs a mask.

// A: a row in GLOBAL memory (weights, L2-resident), k contiguous; B: a row in LDS, 16-byte aligned.  The A loads of KB blocks are
// issued -- unconditional, clamped to the last full block -- before the first MFMA: one memory round trip per 16 KB k.
template <int KB = 2>
__device__ __forceinline__ void mma_g_l(f4& acc, const float* __restrict__ arow, const float* brow, int K, int g) {
    const int kfull = K & ~15;
    for (int k0 = 0; k0 < kfull; k0 += 16 * KB) {
        float av[KB][4];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const float* ap = arow + min(k0 + 16 * q, kfull - 16) + 4 * g;
#pragma unroll
            for (int s = 0; s < 4; ++s) av[q][s] = ap[s];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            if (k0 + 16 * q < kfull) {
                const f4 bv = *reinterpret_cast<const f4*>(brow + k0 + 16 * q + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma4(av[q][s], bv[s], acc);
            }
        }
    }
    if (kfull < K) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kfull + 4 * g + s, kc = min(k, K - 1);
            const float a = arow[kc], b = brow[kc];
            av[s] = k < K ? a : 0.f;
            bv[s] = k < K ? b : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
    }
}

// A: a COLUMN of a global matrix (element k at acol[k * ld]); B: a row in LDS.  K is a layer width here (f_in / f_out): four blocks a batch.
__device__ __forceinline__ void mma_gs_l(f4& acc, const float* __restrict__ acol, int ld, const float* brow, int K, int g) {
    const int kfull = K & ~15;
    for (int k0 = 0; k0 < kfull; k0 += 64) {
        float av[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* ap = acol + (int64_t)(min(k0 + 16 * q, kfull - 16) + 4 * g) * ld;
#pragma unroll
            for (int s = 0; s < 4; ++s) av[q][s] = ap[(int64_t)s * ld];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (k0 + 16 * q < kfull) {
                const f4 bv = *reinterpret_cast<const f4*>(brow + k0 + 16 * q + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma4(av[q][s], bv[s], acc);
            }
        }
    }
    if (kfull < K) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kfull + 4 * g + s, kc = min(k, K - 1);
            const float a = acol[(int64_t)kc * ld], b = brow[kc];
            av[s] = k < K ? a : 0.f;
            bv[s] = k < K ? b : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
    }
}

// both operands rows in LDS (16-byte aligned)
__device__ __forceinline__ void mma_l_l(f4& acc, const float* arow, const float* brow, int K, int g) {
    const int kfull = K & ~15;
    for (int k0 = 0; k0 < kfull; k0 += 16) {
        const f4 av = *reinterpret_cast<const f4*>(arow + k0 + 4 * g), bv = *reinterpret_cast<const f4*>(brow + k0 + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
    }
    if (kfull < K) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kfull + 4 * g + s, kc = min(k, K - 1);
            const float a = arow[kc], b = brow[kc];
            acc = mfma4(k < K ? a : 0.f, k < K ? b : 0.f, acc);
        }
    }
}

// both operands COLUMNS in LDS (the row index is the reduction index: weight gradients); M rows, always masked (M is a row count)
__device__ __forceinline__ void mma_cols(f4& acc, const float* acol, int lda, const float* bcol, int ldb, int M, int g) {
    for (int m0 = 0; m0 < M; m0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + 4 * g + s, mc = min(m, M - 1);
            const float a = acol[mc * lda], b = bcol[mc * ldb];
            av[s] = m < M ? a : 0.f;
            bv[s] = m < M ? b : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
    }
}

// the same with B scaled per ROW (the reduction index) by fac[4 * row]: posttrans' scaler blocks in its weight gradient (fac == NULL: 1)
__device__ __forceinline__ void mma_cols_f(f4& acc, const float* acol, int lda, const float* bcol, int ldb, const float* fac, int M, int g) {
    for (int m0 = 0; m0 < M; m0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + 4 * g + s, mc = min(m, M - 1);
            const float a = acol[mc * lda];
            float b = bcol[mc * ldb];
            if (fac) b *= fac[4 * mc];
            av[s] = m < M ? a : 0.f;
            bv[s] = m < M ? b : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
    }
}

// posttrans' scaler blocks: acc += sum_s fc[s] * (W[n][s K : (s + 1) K] . agg): the S weight segments (a0 + s K, global) against ONE
// LDS row, scaled per segment.  The segments' loads of a 32-k step are all in flight before the step's first MFMA.
__device__ __forceinline__ void mma_scaled3(f4& acc, const float* __restrict__ a0, int K, int S, const float* brow, const f4 fc, int g) {
    const int kfull = K & ~15;
    for (int k0 = 0; k0 < kfull; k0 += 32) {
        float av[3][2][4];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* ap = a0 + min(s, S - 1) * K + min(k0 + 16 * q, kfull - 16) + 4 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e) av[s][q][e] = ap[e];
            }
        __builtin_amdgcn_sched_barrier(0);      // (the batch's loads stay together: hoisting the next batch's doubles the registers)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (k0 + 16 * q < kfull) {
                const f4 bv = *reinterpret_cast<const f4*>(brow + k0 + 16 * q + 4 * g);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (s < S) {
                        const float f = s == 0 ? fc[0] : (s == 1 ? fc[1] : fc[2]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = mfma4(av[s][q][e], bv[e] * f, acc);
                    }
                }
            }
        }
    }
    if (kfull < K) {
        float at[3][4], bt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kc = min(kfull + 4 * g + e, K - 1);
#pragma unroll
            for (int s = 0; s < 3; ++s) at[s][e] = a0[min(s, S - 1) * K + kc];
            bt[e] = brow[kc];
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < S) {
                const float f = s == 0 ? fc[0] : (s == 1 ? fc[1] : fc[2]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = kfull + 4 * g + e < K;
                    acc = mfma4(in ? at[s][e] : 0.f, in ? bt[e] * f : 0.f, acc);
                }
            }
        }
    }
}

// its adjoint: acc[s] += W[:, s K + kk] . g_yr for the S segments at once (A: columns of the global weight, element n at acol[n * ld
// + s * K]; B: a g_yr row in LDS), N = f_out
__device__ __forceinline__ void mma_gs3_l(f4 (&acc)[3], const float* __restrict__ acol, int ld, int K, int S, const float* brow, int N, int g) {
    for (int n0 = 0; n0 < N; n0 += 16) {
        float av[3][4], bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * g + e, nc = min(n, N - 1);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float a = acol[(int64_t)nc * ld + min(s, S - 1) * K];
                av[s][e] = n < N ? a : 0.f;
            }
            const float b = brow[nc];
            bv[e] = n < N ? b : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < S) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[s] = mfma4(av[s][e], bv[e], acc[s]);
            }
        }
    }
}

// ---- work items over (row, column) ----------------------------------------------------------------------------------------------------
struct RowFeat { int r, f; };
struct RowStep { int dr, df; };
__device__ __forceinline__ RowFeat rf_at(int idx, int F) { RowFeat x; x.r = idx / F; x.f = idx - x.r * F; return x; }
__device__ __forceinline__ RowStep rf_stride(int step, int F) { RowStep q; q.dr = step / F; q.df = step - q.dr * F; return q; }
__device__ __forceinline__ void rf_step(RowFeat& x, const RowStep& q, int F) { x.r += q.dr; x.f += q.df; if (x.f >= F) { x.f -= F; ++x.r; } }

struct Ctx {
    int lo, e0, R, Eb, t;
    float *HB, *PQ, *EIG, *W, *FAC, *XP, *Y, *Y0S, *COEF, *GA, *GB, *GC, *VEC;
    int *IP, *CP, *CUR, *SRC, *DST, *CSCI;
    double* RED;
};

__device__ __forceinline__ float weight_from_stats(int kind, float alpha, float eps, float d, float sabs, float spos, float sneg, float mx, float se) {
    if (kind == DGN_W_ABSNORM) return d / (sabs + eps);
    if (kind == DGN_W_BALANCED) return (fmaxf(d, 0.f) / (spos + eps) + fmaxf(-d, 0.f) / (sneg + eps)) / 2.f;
    return expf(alpha * fabsf(d) - mx) / se;
}

// descriptor -> every operand of the (block, tower) in LDS: h rows, CSR rows re-based on the block, eig columns, scaler factors / graph
// norm, biases, (backward) the transposed view; then the edge weights (aggregators.py:36-69) and P | Q.  Ends on a barrier.
template <bool HAS_PRE>
__device__ __forceinline__ void block_prologue(const P& p, Ctx& c, float* lds, bool bwd, const int4 d) {
    const int tid = threadIdx.x, NT = blockDim.x;
    c.lo = d.x; c.e0 = d.z; c.R = d.y - d.x; c.Eb = d.w - d.z; c.t = blockIdx.y;
    const Layout& L = p.L;
    c.HB = lds + L.hb; c.PQ = lds + L.pq; c.EIG = lds + L.eig; c.W = lds + L.w; c.FAC = lds + L.fac; c.XP = lds + L.xp;
    c.Y = lds + L.y; c.Y0S = lds + L.y0s; c.COEF = lds + L.coef; c.GA = lds + L.ga; c.GB = lds + L.gb; c.GC = lds + L.gc; c.VEC = lds + L.vec;
    c.IP = reinterpret_cast<int*>(lds + L.ip); c.CP = reinterpret_cast<int*>(lds + L.cp); c.SRC = reinterpret_cast<int*>(lds + L.src);
    c.DST = reinterpret_cast<int*>(lds + L.dst); c.CSCI = reinterpret_cast<int*>(lds + L.csci); c.CUR = reinterpret_cast<int*>(lds + L.cur);
    c.RED = reinterpret_cast<double*>(lds + L.red);
    const int R = c.R, Eb = c.Eb, fi = p.fi, ldh = L.ldh;
    {   // (everything below is issued before anything is waited for: one memory round trip)
        const float* hrow = p.h + (int64_t)c.lo * p.F + c.t * fi;
        RowFeat x = rf_at(tid, fi);
        const RowStep st = rf_stride(NT, fi);
        for (; x.r < R; rf_step(x, st, fi)) c.HB[x.r * ldh + x.f] = hrow[(int64_t)x.r * p.F + x.f];
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
        if (HAS_PRE)
            for (int i = tid; i < fi; i += NT) c.VEC[i] = p.b_pre[c.t][i];
        for (int i = tid; i < p.fo; i += NT) c.VEC[fi + i] = p.b_post[c.t][i];
        if (bwd) {
            // closed block: the (source, slot) ranks of its slots are exactly [e0, e0 + Eb)
            for (int i = tid; i <= R; i += NT) {
                const int v = p.csc_ptr[c.lo + i] - c.e0;
                c.CP[i] = v;
                if (i < R) c.CUR[i] = v;
            }
            for (int i = tid; i < Eb; i += NT) c.CSCI[p.csc_pos[c.e0 + i] - c.e0] = i;
            for (int i = tid; i < R * ldh; i += NT) c.GA[i] = 0.f;
            // the block's y0 and g_y1 rows (this tower's columns): raw here, turned into g_yr once BatchNorm's sums are known
            const int fo = p.fo, ldy = L.ldy;
            const int64_t ybase = (int64_t)c.lo * p.Fo + c.t * fo;
            RowFeat y = rf_at(tid, fo);
            const RowStep sy = rf_stride(NT, fo);
            for (; y.r < R; rf_step(y, sy, fo)) {
                c.Y0S[y.r * ldy + y.f] = p.y0[ybase + (int64_t)y.r * p.Fo + y.f];
                c.Y[y.r * ldy + y.f] = p.g_y1[ybase + (int64_t)y.r * p.Fo + y.f];
            }
        }
    }
    __syncthreads();
    BLK_STAMP(1);
    for (int r = tid; r < R; r += NT)
        for (int j = c.IP[r]; j < c.IP[r + 1]; ++j) c.DST[j] = r;
    __syncthreads();
    // delta_jc = eig[src_j, c] - eig[i, c], parked in the weight planes
    for (int i = tid; i < Eb * p.n_ch; i += NT) {
        const int ch = i / Eb, j = i - ch * Eb;
        c.W[ch * L.ld_w + j] = c.EIG[c.SRC[j] * p.n_ch + ch] - c.EIG[c.DST[j] * p.n_ch + ch];
    }
    __syncthreads();
    BLK_STAMP(2);
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
    // P | Q = h [W_s | W_d]^T + [0 | b]: jobs of (16-row strip, half, 16-column tile)
    if constexpr (HAS_PRE) {
        const int lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
        const int nstrip = (R + 15) >> 4, ntq = (fi + 15) >> 4;
        const float* wpre = p.w_pre[c.t];
        for (int job = wave; job < nstrip * 2 * ntq; job += nw) {
            const int tq = job % ntq, half = (job / ntq) & 1, strip = job / (2 * ntq);
            const int n = tq * 16 + i16, m = strip * 16 + i16;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_g_l(acc, wpre + (int64_t)min(n, fi - 1) * p.ld_pre + half * fi, c.HB + min(m, R - 1) * ldh, fi, g);
            if (m < R) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = tq * 16 + 4 * g + s;
                    if (o < fi) c.PQ[m * 2 * ldh + half * ldh + o] = acc[s] + (half ? c.VEC[o] : 0.f);
                }
            }
        }
    }
    __syncthreads();
    BLK_STAMP(3);
}

// message of slot j (source row s) into row r, feature f -- x_dst + x_src, the sweep's rounding order (load_msg)
template <bool HAS_PRE>
__device__ __forceinline__ float msg_at(const P& p, const Ctx& c, int s, int r, int f) {
    const int ldh = p.L.ldh;
    return HAS_PRE ? c.PQ[r * 2 * ldh + ldh + f] + c.PQ[s * 2 * ldh + f] : c.HB[s * ldh + f];
}

// the accumulators of row r, feature f, slots in ascending order (positions tracked as block-local slot ids).  Four slots per step,
// every LDS operand of the group requested before the first is consumed (one slot at a time is a chain of dependent LDS latencies).
template <class C, bool TRACK, bool HAS_PRE>
__device__ __forceinline__ void accumulate_row(Acc<C, TRACK>& acc, const P& p, const Ctx& c, int r, int f) {
    acc.init();
    const int ldh = p.L.ldh;
    const float q = HAS_PRE ? c.PQ[r * 2 * ldh + ldh + f] : 0.f;
    const float* xs = HAS_PRE ? c.PQ + f : c.HB + f;
    const int ldx = HAS_PRE ? 2 * ldh : ldh;
    const int beg = c.IP[r], end = c.IP[r + 1];
    for (int j0 = beg; j0 < end; j0 += 4) {
        int sv[4];
        float xv[4], wv[4][C::NW];
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = c.SRC[min(j0 + u, end - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int ch = 0; ch < C::NW; ++ch) wv[u][ch] = 0.f;
#pragma unroll
            for (int ch = 0; ch < C::NCH; ++ch) wv[u][ch] = c.W[ch * p.L.ld_w + min(j0 + u, end - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = xs[sv[u] * ldx];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u < end) {
                float m[1];
                m[0] = HAS_PRE ? q + xv[u] : xv[u];
                acc.add(m, wv[u], j0 + u);
            }
        }
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------------------
// (512 threads at most: the forward's uniform state spills past the 128 registers a 1024-thread workgroup leaves a lane)
template <class O, class C, bool HAS_PRE>
__global__ __launch_bounds__(512) void blk_forward(const P p) {
    extern __shared__ float lds[];
    Ctx c;
    BLK_STAMP(0);
    int skipped_end;
    const int4 desc = block_desc(p, skipped_end);
    if (desc.y <= desc.x) {      // (an unused entry of a padded batch's table: zero BatchNorm partials)
        if ((int)threadIdx.x < p.fo) {
            p.bn_part[((int64_t)blockIdx.x * 2 + 0) * p.Fo + blockIdx.y * p.fo + threadIdx.x] = 0.0;
            p.bn_part[((int64_t)blockIdx.x * 2 + 1) * p.Fo + blockIdx.y * p.fo + threadIdx.x] = 0.0;
        }
        // an over-sized block: zeros for this tower's columns of its y0 rows (the tails read every valid row)
        for (int i = threadIdx.x; i < (skipped_end - desc.x) * p.fo; i += blockDim.x)
            p.y0[(int64_t)(desc.x + i / p.fo) * p.Fo + blockIdx.y * p.fo + i % p.fo] = 0.f;
        return;
    }
    block_prologue<HAS_PRE>(p, c, lds, false, desc);
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int R = c.R, fi = p.fi, fo = p.fo, RC = p.RC, ldh = p.L.ldh, ldy = p.L.ldy, kp = p.L.kp;
    const float* wpost = p.w_post[c.t];
    double s0 = 0.0, s1 = 0.0;                                 // BatchNorm partial sums of column tid
    for (int c0 = 0; c0 < R; c0 += RC) {
        const int rc = min(RC, R - c0);
        // the aggregation, a work item per (row, feature), straight into posttrans' input rows [h | scale_s * aggregator blocks]
        {
            RowFeat x = rf_at(tid, fi);
            const RowStep st_ = rf_stride(NT, fi);
            for (; x.r < rc; rf_step(x, st_, fi)) {
                const int r = c0 + x.r, f = x.f;
                float* xrow = c.XP + x.r * kp;
                const float xin[1] = {c.HB[r * ldh + f]};
                if (HAS_PRE) xrow[f] = xin[0];
                const int deg = c.IP[r + 1] - c.IP[r];
                float* out = xrow + p.L.ho + f;
                if (deg == 0) {
                    for (int a = 0; a < p.A; ++a) out[a * fi] = 0.f;
                    continue;
                }
                Acc<C, false> acc;
                accumulate_row<C, false, HAS_PRE>(acc, p, c, r, f);
                RowStats<1> st;
                row_stats<C, false>(st, acc, (float)deg, p.a);
                for_each_agg<O>(p.a, [&](int a) {
                    float val[1];
                    agg_value<C, false>(val, O::op(p.a, a), O::ch(p.a, a), acc, st, xin);
                    out[a * fi] = val[0];
                });
            }
        }
        __syncthreads();
        BLK_STAMP(5);
        if (p.dbg_agg) {      // (tests: the identity scaler's block = the aggregate rows)
            RowFeat x = rf_at(tid, p.K);
            const RowStep st_ = rf_stride(NT, p.K);
            for (; x.r < rc; rf_step(x, st_, p.K)) p.dbg_agg[(int64_t)(c.lo + c0 + x.r) * (p.T * p.K) + c.t * p.K + x.f] = c.XP[x.r * kp + p.L.ho + x.f];
        }
        // posttrans (+ bias) * graph norm: jobs of (strip, 16-column tile); the scalers are applied to the aggregate row on the fly
        {
            const int nstrip = (rc + 15) >> 4, ntq = (fo + 15) >> 4;
            for (int job = wave; job < nstrip * ntq; job += nw) {
                const int tq = job % ntq, strip = job / ntq;
                const int n = tq * 16 + i16, m = strip * 16 + i16;
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* wrow = wpost + (int64_t)min(n, fo - 1) * p.ld_post;
                const float* xrow = c.XP + min(m, rc - 1) * kp;
                const f4 fc = *reinterpret_cast<const f4*>(c.FAC + 4 * (c0 + min(m, rc - 1)));
                mma_scaled3(acc, wrow + p.h_off, p.K, p.S, xrow + p.L.ho, fc, g);
                if (HAS_PRE) mma_g_l(acc, wrow, xrow, fi, g);
                if (m < rc) {
                    const float sn = fc[3];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int o = tq * 16 + 4 * g + s;
                        if (o < fo) c.Y[m * ldy + o] = (acc[s] + c.VEC[fi + o]) * sn;
                    }
                }
            }
        }
        __syncthreads();
        BLK_STAMP(6);
        {
            float* yrow = p.y0 + (int64_t)(c.lo + c0) * p.Fo + c.t * fo;
            RowFeat x = rf_at(tid, fo);
            const RowStep st_ = rf_stride(NT, fo);
            for (; x.r < rc; rf_step(x, st_, fo)) yrow[(int64_t)x.r * p.Fo + x.f] = c.Y[x.r * ldy + x.f];
            if (tid < fo)
                for (int m = 0; m < rc; ++m) {
                    const double v = (double)c.Y[m * ldy + tid];
                    s0 += v; s1 += v * v;
                }
        }
        if (c0 + RC < R) __syncthreads();
    }
    if (tid < fo) {
        p.bn_part[((int64_t)blockIdx.x * 2 + 0) * p.Fo + c.t * fo + tid] = s0;
        p.bn_part[((int64_t)blockIdx.x * 2 + 1) * p.Fo + c.t * fo + tid] = s1;
    }
    BLK_STAMP(10);
}

// column sums of a [parts][2][Fo] table of doubles (a part = one row of 2 Fo doubles) in a fixed order: NT / Fo groups of threads (at most
// 16) take interleaved parts, a thread owns one 16-byte pair of the row, twelve loads in flight, and the groups are added in order.  Result
// in RED[0 .. 2 Fo) ([sum | second sum]); ends on a barrier.  RED: 2 * Fo * 17 doubles.
__device__ __forceinline__ void column_sums(const double* part, int parts, int Fo, double* RED) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int G = max(1, min(NT / Fo, kRedGroups));
    const int grp = tid / Fo, q = tid - grp * Fo;          // the pair (2 q, 2 q + 1) of the 2 Fo doubles
    if (grp < G) {
        double2 a = make_double2(0.0, 0.0);
        for (int b0 = grp; b0 < parts; b0 += 12 * G) {
            double2 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) v[u] = *reinterpret_cast<const double2*>(part + (int64_t)min(b0 + u * G, parts - 1) * 2 * Fo + 2 * q);
#pragma unroll
            for (int u = 0; u < 12; ++u)
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

// the same for `n` columns from `col0` on (a tower's): RED[0 .. n) / RED[n .. 2 n); RED: 2 * n * 17 doubles
__device__ __forceinline__ void column_sums_cols(const double* part, int parts, int Fo, int col0, int n, double* RED) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int G = max(1, min(NT / n, kRedGroups));
    const int grp = tid / n, q = tid - grp * n;
    if (grp < G) {
        double a0 = 0.0, a1 = 0.0;
        for (int b0 = grp; b0 < parts; b0 += 4 * G) {
            double v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double* at = part + (int64_t)min(b0 + u * G, parts - 1) * 2 * Fo + col0 + q;
                v0[u] = at[0]; v1[u] = at[Fo];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (b0 + u * G < parts) { a0 += v0[u]; a1 += v1[u]; }
        }
        RED[(grp + 1) * 2 * n + q] = a0;
        RED[(grp + 1) * 2 * n + n + q] = a1;
    }
    __syncthreads();
    if (tid < 2 * n) {
        double t = 0.0;
        for (int k = 0; k < G; ++k) t += RED[(k + 1) * 2 * n + tid];
        RED[tid] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ float col_param(const float* const (&ptrs)[kMaxT], int col, int fo, int T) {
    int t = 0;
    for (int q = 1; q < T; ++q) t += (col >= q * fo) ? 1 : 0;
    return ptrs[t][col - t * fo];
}

// Philox4x32-10 exactly as dgn_bn_tail.hip's dropout_fwd keys it: counter = (group lo, group hi * 2 + half, offset lo, offset hi), key = *seed
__device__ __forceinline__ void blk_philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// the residual's share of d h at element `at` of the dense [N, F] tensors: the output gradient -- through the layer's final dropout where the simple /
// complex layer has one (its keep bit from the mask the forward tail stored; blk_dropout_rows' arithmetic)
__device__ __forceinline__ float residual_grad(const P& p, int64_t at) {
    const float gv = p.g_out[at];
    if (p.mixing || p.drop_scale == 0.f) return gv;
    return (p.drop_mask[at >> 3] >> (at & 7)) & 1u ? gv * p.drop_scale : 0.f;
}

// The tail's rows [m0, m0 + RW) of the dense [N, Fo] tensor are whole groups of 8 elements (RW is a multiple of 16).  `draw`: the keep
// bits are drawn (forward) and the byte stored; else read back.  Every element of a kept position is scaled, a dropped one zeroed.
__device__ __forceinline__ void blk_dropout_rows(const P& p, float* rows_lds, int ldk, int64_t m0, int rows, int RW, bool draw) {
    const int Fo = p.Fo;
    const int64_t g0 = m0 * Fo / 8;
    const int n_groups = RW * Fo / 8;
    uint64_t sd = 0;
    if (draw) sd = (uint64_t)*p.drop_seed;
    for (int gi = threadIdx.x; gi < n_groups; gi += blockDim.x) {
        if ((gi * 8) / Fo >= rows) continue;                    // (groups wholly past the batch's rows: nothing drawn, nothing stored)
        const int64_t gg = g0 + gi;
        unsigned m;
        if (draw) {
            uint32_t r[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint32_t c[4] = {(uint32_t)gg, (uint32_t)((uint64_t)gg >> 32) * 2u + (uint32_t)half, (uint32_t)p.drop_offset, (uint32_t)(p.drop_offset >> 32)};
                blk_philox(c, (uint32_t)sd, (uint32_t)(sd >> 32));
#pragma unroll
                for (int i = 0; i < 4; ++i) r[4 * half + i] = c[i];
            }
            m = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) m |= (r[i] >= p.drop_threshold ? 1u : 0u) << i;
            p.drop_mask[gg] = (unsigned char)m;
        } else {
            m = p.drop_mask[gg];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = gi * 8 + i, r_ = e / Fo, c_ = e - r_ * Fo;
            if (r_ < rows) {
                float* at = rows_lds + r_ * ldk + c_;
                *at = (m >> i) & 1u ? *at * p.drop_scale : 0.f;
            }
        }
    }
}

// ---- forward tail: BatchNorm (training statistics) -> ReLU -> + h   or   -> mixing Linear -> LeakyReLU -> + h ------------------------
// LDS: [mean | invstd | gamma | beta] (4 Fo floats, padded to a multiple of 4), Y1 [rows][ldk], W_mix [Fo][ldk] (towers; ldk = up4(Fo)),
// then the reduction scratch (doubles)
__global__ __launch_bounds__(512) void blk_tail_fwd(const P p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int Fo = p.Fo, RW = p.tail_rows, ldk = (Fo + 3) & ~3, c4 = (4 * Fo + 3) & ~3;
    float* MEAN = lds; float* INVSTD = lds + Fo; float* GAM = lds + 2 * Fo; float* BET = lds + 3 * Fo;
    float* Y1 = lds + c4;
    float* WM = Y1 + RW * ldk;
    const int n_wm = p.mixing ? Fo * ldk : 0;
    double* RED = reinterpret_cast<double*>(lds + ((c4 + RW * ldk + n_wm + 1) & ~1));
    const int64_t m0 = (int64_t)blockIdx.x * RW, Nv = valid_rows(p);
    const int rows = (int)max((int64_t)0, min((int64_t)RW, Nv - m0)), rows_buf = (int)min((int64_t)RW, p.N - m0);
    const RowStep st_ = rf_stride(NT, Fo);
    const bool drop_last = !p.mixing && p.drop_scale != 0.f && !p.eval_mode;
    if (rows < rows_buf) {      // padding rows of the buffer: zeros (their readers -- the readout's padding row -- must see finite values)
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < rows_buf; rf_step(x, st_, Fo)) if (x.r >= rows) p.out[(m0 + x.r) * Fo + x.f] = 0.f;
    }
    if (rows == 0 && blockIdx.x != 0) return;
    {   // this workgroup's rows and the mixing weight are requested before the statistics are summed
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < RW; rf_step(x, st_, Fo)) Y1[x.r * ldk + x.f] = x.r < rows ? p.y0[(m0 + x.r) * Fo + x.f] : 0.f;
        if (p.mixing) {
            RowFeat w = rf_at(tid, Fo);
            for (; w.r < Fo; rf_step(w, st_, Fo)) WM[w.r * ldk + w.f] = p.w_mix[w.r * Fo + w.f];
        }
    }
    if (p.eval_mode) {      // evaluation: the running statistics, nothing updated (nn.BatchNorm1d.eval())
        if (tid < Fo) {
            MEAN[tid] = p.running_mean[tid]; INVSTD[tid] = 1.f / sqrtf(p.running_var[tid] + p.bn_eps);
            GAM[tid] = col_param(p.gamma, tid, p.fo, p.T); BET[tid] = col_param(p.beta, tid, p.fo, p.T);
        }
    } else {
    column_sums(p.bn_part, p.n_blocks, Fo, RED);
    if (tid < Fo) {
        const double n = (double)Nv;
        const double mu = RED[tid] / n;
        double m2 = RED[Fo + tid] - mu * RED[tid];
        if (m2 < 0.0) m2 = 0.0;
        const float mean = (float)mu, invstd = (float)(1.0 / sqrt(m2 / n + (double)p.bn_eps));
        MEAN[tid] = mean; INVSTD[tid] = invstd;
        GAM[tid] = col_param(p.gamma, tid, p.fo, p.T); BET[tid] = col_param(p.beta, tid, p.fo, p.T);
        if (blockIdx.x == 0) {
            p.save_mean[tid] = mean; p.save_invstd[tid] = invstd;
            const float unbiased = (float)(Nv > 1 ? m2 / (n - 1.0) : m2 / n);
            p.running_mean[tid] = (1.f - p.momentum) * p.running_mean[tid] + p.momentum * mean;
            p.running_var[tid] = (1.f - p.momentum) * p.running_var[tid] + p.momentum * unbiased;
            if (tid < p.n_nbt) p.nbt[tid] += 1;
        }
    }
    }
    __syncthreads();
    {
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < rows; rf_step(x, st_, Fo)) {
            float v = (Y1[x.r * ldk + x.f] - MEAN[x.f]) * INVSTD[x.f] * GAM[x.f] + BET[x.f];
            if (!p.mixing) {
                if (p.relu) v = fmaxf(v, 0.f);
                const int64_t at = (m0 + x.r) * Fo + x.f;
                if (p.residual) v += p.h[at];
                if (drop_last) Y1[x.r * ldk + x.f] = v;      // (the finished rows wait in LDS for their keep bits)
                else p.out[at] = v;
            } else {
                Y1[x.r * ldk + x.f] = v;
            }
        }
    }
    if (drop_last) {                                  // the layer's last op: F.dropout(h, p, training)     (:130, :201)
        __syncthreads();
        blk_dropout_rows(p, Y1, ldk, m0, rows, RW, true);
        __syncthreads();
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < rows; rf_step(x, st_, Fo)) p.out[(m0 + x.r) * Fo + x.f] = Y1[x.r * ldk + x.f];
    }
    if (!p.mixing) return;
    __syncthreads();
    if (p.drop_scale != 0.f && !p.eval_mode) {      // the towers' dropout on the normalised rows (:275)
        blk_dropout_rows(p, Y1, ldk, m0, rows, RW, true);
        __syncthreads();
    }
    // out = LeakyReLU(y1 W_mix^T + b_mix) (+ h): jobs of (strip, 16-column tile)
    const int nstrip = RW >> 4, ntq = (Fo + 15) >> 4;
    for (int job = wave; job < nstrip * ntq; job += nw) {
        const int tq = job % ntq, strip = job / ntq;
        const int n = tq * 16 + i16, m = strip * 16 + i16;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_l_l(acc, WM + min(n, Fo - 1) * ldk, Y1 + m * ldk, Fo, g);
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
// LDS: [mean | invstd | gamma | beta], XH [RW][ldk] (normalised y0), Y1, GZ, GY1 [RW][ldk] each, W_mix [Fo][ldk] (towers)
__global__ __launch_bounds__(512) void blk_tail_bwd(const P p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int Fo = p.Fo, RW = p.tail_rows, ldk = (Fo + 3) & ~3, c4 = (4 * Fo + 3) & ~3;
    float* MEAN = lds; float* INVSTD = lds + Fo; float* GAM = lds + 2 * Fo; float* BET = lds + 3 * Fo;
    float* XH = lds + c4; float* Y1 = XH + RW * ldk; float* GZ = Y1 + RW * ldk; float* GY1 = GZ + RW * ldk; float* WM = GY1 + RW * ldk;
    const int64_t m0 = (int64_t)blockIdx.x * RW, Nv = valid_rows(p);
    const int rows = (int)max((int64_t)0, min((int64_t)RW, Nv - m0)), rows_buf = (int)min((int64_t)RW, p.N - m0);
    const RowStep st_ = rf_stride(NT, Fo);
    if (rows < rows_buf) {      // padding rows: d h = 0 (no block owns them; their cotangent reaches the embedding's gradient otherwise)
        RowFeat x = rf_at(tid, p.F);
        const RowStep sf = rf_stride(NT, p.F);
        for (; x.r < rows_buf; rf_step(x, sf, p.F)) if (x.r >= rows) p.g_h[(m0 + x.r) * p.F + x.f] = 0.f;
    }
    if (tid < Fo) {
        MEAN[tid] = p.save_mean[tid]; INVSTD[tid] = p.save_invstd[tid];
        GAM[tid] = col_param(p.gamma, tid, p.fo, p.T); BET[tid] = col_param(p.beta, tid, p.fo, p.T);
    }
    {
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < RW; rf_step(x, st_, Fo)) {
            const bool in = x.r < rows;
            XH[x.r * ldk + x.f] = in ? p.y0[(m0 + x.r) * Fo + x.f] : 0.f;
            GZ[x.r * ldk + x.f] = in ? p.g_out[(m0 + x.r) * Fo + x.f] : 0.f;
        }
        if (p.mixing) {
            RowFeat w = rf_at(tid, Fo);
            for (; w.r < Fo; rf_step(w, st_, Fo)) WM[w.r * ldk + w.f] = p.w_mix[w.r * Fo + w.f];
        }
    }
    __syncthreads();
    if (!p.mixing && p.drop_scale != 0.f) {          // the layer's final dropout: its adjoint on the staged output gradient
        blk_dropout_rows(p, GZ, ldk, m0, rows, RW, false);
        __syncthreads();
    }
    {
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < RW; rf_step(x, st_, Fo)) {
            const int i = x.r * ldk + x.f;
            const bool in = x.r < rows;
            const float xh = in ? (XH[i] - MEAN[x.f]) * INVSTD[x.f] : 0.f;
            const float y1 = xh * GAM[x.f] + BET[x.f];
            XH[i] = xh;
            Y1[i] = in ? y1 : 0.f;
            if (!p.mixing) GY1[i] = (in && (!p.relu || y1 > 0.f)) ? GZ[i] : 0.f;      // ReLU: the gradient passes where BatchNorm's output is > 0
        }
    }
    __syncthreads();
    if (p.mixing && p.drop_scale != 0.f) {           // the mixing network saw the DROPPED normalised rows
        blk_dropout_rows(p, Y1, ldk, m0, rows, RW, false);
        __syncthreads();
    }
    if (p.mixing) {
        const int nstrip = RW >> 4, ntq = (Fo + 15) >> 4;
        // g_z = g_out * LeakyReLU'(y1 W_mix^T + b_mix), in place over the staged g_out
        for (int job = wave; job < nstrip * ntq; job += nw) {
            const int tq = job % ntq, strip = job / ntq;
            const int n = tq * 16 + i16, m = strip * 16 + i16;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_l_l(acc, WM + min(n, Fo - 1) * ldk, Y1 + m * ldk, Fo, g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = tq * 16 + 4 * g + s;
                if (o < Fo) {
                    const float z = acc[s] + p.b_mix[o];
                    GZ[m * ldk + o] *= (z > 0.f ? 1.f : p.slope);
                }
            }
        }
        __syncthreads();
        // g_y1 = g_z W_mix: the reduction runs over W_mix's ROWS: column reads of the staged weight
        for (int job = wave; job < nstrip * ntq; job += nw) {
            const int tq = job % ntq, strip = job / ntq;
            const int kk = tq * 16 + i16, m = strip * 16 + i16;
            const float* wcol = WM + min(kk, Fo - 1);
            const float* grow = GZ + m * ldk;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int n0 = 0; n0 < Fo; n0 += 16) {
                float av[4], bv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int n = n0 + 4 * g + s, nc = min(n, Fo - 1);
                    const float a = wcol[nc * ldk], b = grow[nc];
                    av[s] = n < Fo ? a : 0.f;
                    bv[s] = n < Fo ? b : 0.f;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma4(av[s], bv[s], acc);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = tq * 16 + 4 * g + s;
                if (o < Fo) GY1[m * ldk + o] = acc[s];
            }
        }
        // d W_mix partial: D[n][k] = sum_m g_z[m][n] y1[m][k] over this workgroup's rows; d b_mix partial
        float* wpart = p.tail_wpart + (int64_t)blockIdx.x * (Fo * Fo + Fo);
        for (int job = wave; job < ntq * ntq; job += nw) {
            const int tk = job % ntq, tn = job / ntq;
            const int kk = tk * 16 + i16;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_cols(acc, GZ + min(tn * 16 + i16, Fo - 1), ldk, Y1 + min(kk, Fo - 1), ldk, RW, g);
            if (kk < Fo) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int nn = tn * 16 + 4 * g + s;
                    if (nn < Fo) wpart[nn * Fo + kk] = acc[s];
                }
            }
        }
        if (tid < Fo) {
            float a = 0.f;
            for (int m = 0; m < RW; ++m) a += GZ[m * ldk + tid];
            wpart[Fo * Fo + tid] = a;
        }
        __syncthreads();
        if (p.drop_scale != 0.f) {                   // dropout's adjoint on the gradient of the normalised rows
            blk_dropout_rows(p, GY1, ldk, m0, rows, RW, false);
            __syncthreads();
        }
    }
    {
        RowFeat x = rf_at(tid, Fo);
        for (; x.r < rows; rf_step(x, st_, Fo)) p.g_y1[(m0 + x.r) * Fo + x.f] = GY1[x.r * ldk + x.f];
    }
    if (tid < Fo) {
        double a0 = 0.0, a1 = 0.0;
        for (int m = 0; m < rows; ++m) {
            const double gv = (double)GY1[m * ldk + tid];
            a0 += gv; a1 += gv * (double)XH[m * ldk + tid];
        }
        p.tail_part[((int64_t)blockIdx.x * 2 + 0) * Fo + tid] = a0;
        p.tail_part[((int64_t)blockIdx.x * 2 + 1) * Fo + tid] = a1;
    }
}

// d m_j of slot j (destination row at chunk row mc, source u) at feature f, from the destination's coefficient rows
template <class C>
__device__ __forceinline__ float edge_grad(const P& p, const Ctx& c, int mc, int j, int f, float m_j) {
    const int fi = p.fi;
    const float* cf = c.COEF + (mc * p.L.n_coef) * fi + f;
    float gm = p.cmap[CF_C0] >= 0 ? cf[p.cmap[CF_C0] * fi] : 0.f;
    if constexpr (C::STATS) {
        if (p.cmap[CF_CV] >= 0) gm = fmaf(cf[p.cmap[CF_CV] * fi], m_j, gm);
    }
#pragma unroll
    for (int ch = 0; ch < C::NCH; ++ch) {
        const float w = c.W[ch * p.L.ld_w + j];
        if (p.cmap[CF_CS0 + ch] >= 0) gm = fmaf(w, cf[p.cmap[CF_CS0 + ch] * fi], gm);
        if constexpr (C::AV) {
            if (p.cmap[CF_CA0 + ch] >= 0) gm = fmaf(fabsf(w), cf[p.cmap[CF_CA0 + ch] * fi], gm);
        }
    }
    if constexpr (C::STATS) {
        if (p.cmap[CF_ARG] >= 0) {
            const unsigned arg = __float_as_uint(cf[p.cmap[CF_ARG] * fi]);  // (amax + 1) | (amin + 1) << 16, block-local slot ids
            if ((arg & 0xffffu) == (unsigned)(j + 1) && p.cmap[CF_GMAX] >= 0) gm += cf[p.cmap[CF_GMAX] * fi];
            if ((arg >> 16) == (unsigned)(j + 1) && p.cmap[CF_GMIN] >= 0) gm += cf[p.cmap[CF_GMIN] * fi];
        }
    }
    return gm;
}

// posttrans' adjoint for NS 16-row strips of g_yr at once (gy: strip st's rows at gy + (16 st + i16) * ldy): acc[st][s] += W[:, s K + kk] . g_yr.
// The weight columns -- global loads, the job's only memory latency -- are fetched ONCE per 16-n block instead of once per strip.
// S == 0: one unscaled segment (the h block of posttrans' input), result in acc[st][0].
template <int NS>
__device__ __forceinline__ void mma_gs3_strips(f4 (&acc)[NS][3], const float* __restrict__ acol, int ld, int K, int S, const float* gy, int ldy,
                                               int N, int i16, int g) {
    const int S_ = max(S, 1);
    for (int n0 = 0; n0 < N; n0 += 16) {
        float av[3][4];
        int nc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * g + e;
            nc[e] = min(n, N - 1);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float a = acol[(int64_t)nc[e] * ld + min(s, S_ - 1) * K];
                av[s][e] = n < N ? a : 0.f;
            }
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const float* brow = gy + (16 * st + i16) * ldy;
            float bv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float b = brow[nc[e]];
                bv[e] = n0 + 4 * g + e < N ? b : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s < S_) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[st][s] = mfma4(av[s][e], bv[e], acc[st][s]);
                }
            }
        }
    }
}

// d [h | aggregates] of a chunk of NS strips: d agg = sum_s scale_s (g_yr W_post[:, block s]), d h through posttrans' h block.  A job =
// one 16-column tile for ALL strips (a job per (strip, tile) gave the largest blocks, three strips, five rounds of one-latency jobs on
// four waves where the small blocks ran two).
template <int NS, bool HAS_PRE>
__device__ __forceinline__ void dagg_chunk(const P& p, const Ctx& c, float* XP, const float* __restrict__ wpost, const float* GY, int c0, int rc, int wave,
                                           int nw, int i16, int g) {
    // (XP, GY, c0: the pass's first row -- of the chunk's LDS rows, of g_yr, in the block; rc: rows from there to the chunk's end)
    const int fi = p.fi, fo = p.fo, ldy = p.L.ldy, kp = p.L.kp;
    const int ntk = (p.K + 15) >> 4, nth = HAS_PRE ? (fi + 15) >> 4 : 0;
    for (int job = wave; job < ntk + nth; job += nw) {
        f4 acc[NS][3];
#pragma unroll
        for (int st = 0; st < NS; ++st)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[st][q] = f4{0.f, 0.f, 0.f, 0.f};
        if (job < ntk) {
            const int tk = job, kk = tk * 16 + i16;
            mma_gs3_strips<NS>(acc, wpost + p.h_off + min(kk, p.K - 1), p.ld_post, p.K, p.S, GY, ldy, fo, i16, g);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int m = st * 16 + i16;
                const f4 fc = *reinterpret_cast<const f4*>(c.FAC + 4 * (c0 + min(m, rc - 1)));
                f4 out = acc[st][0] * fc[0];
                if (p.S > 1) out += acc[st][1] * fc[1];
                if (p.S > 2) out += acc[st][2] * fc[2];
                if (m < rc && p.L.ho + tk * 16 + 4 * g < kp) *reinterpret_cast<f4*>(XP + m * kp + p.L.ho + tk * 16 + 4 * g) = out;
            }
        } else {
            const int th = job - ntk, kk = th * 16 + i16;
            mma_gs3_strips<NS>(acc, wpost + min(kk, fi - 1), p.ld_post, 0, 0, GY, ldy, fo, i16, g);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int m = st * 16 + i16;
                if (m < rc && th * 16 + 4 * g < p.L.ho) *reinterpret_cast<f4*>(XP + m * kp + th * 16 + 4 * g) = acc[st][0];
            }
        }
    }
}

// ---- backward of a (block, tower) -------------------------------------------------------------------------------------------------
template <class O, class C, bool HAS_PRE>
__global__ __launch_bounds__(1024) void blk_backward(const P p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = uniform_i(tid >> 6), nw = NT >> 6, i16 = lane & 15, g = lane >> 4;
    const int fi = p.fi, fo = p.fo, RC = p.RC, ldh = p.L.ldh, ldy = p.L.ldy, kp = p.L.kp;
    const int t = blockIdx.y, yc0 = t * fo;
    Ctx c;
    BLK_STAMP(0);
    int skipped_end;
    const int4 desc = block_desc(p, skipped_end);
    const bool skip = desc.y <= desc.x;
    if (!skip || blockIdx.x == 0) {
        // BatchNorm's column sums of this tower's columns (= d beta, d gamma) from the tail's partials (block 0 writes them whether or
        // not it has rows of its own: a skipped block 0 used to leave them uninitialised)
        double* RED = reinterpret_cast<double*>(lds + p.L.red);
        column_sums_cols(p.tail_part, p.n_tail, p.Fo, yc0, fo, RED);
        if (blockIdx.x == 0 && tid < fo) { p.g_beta[yc0 + tid] = (float)RED[tid]; p.g_gamma[yc0 + tid] = (float)RED[fo + tid]; }
    }
    if (skip) {      // (an unused entry of a padded batch's table: a zero parameter-gradient partial)
        float* bpart0 = p.blk_part + (int64_t)blockIdx.x * p.n_blk_param + t * p.off_tower;
        for (int i = tid; i < p.off_tower; i += NT) bpart0[i] = 0.f;
        // an over-sized block: zeros for this tower's columns of its d h rows
        for (int i = tid; i < (skipped_end - desc.x) * fi; i += NT) p.g_h[(int64_t)(desc.x + i / fi) * p.F + t * fi + i % fi] = 0.f;
        return;
    }
    block_prologue<HAS_PRE>(p, c, lds, true, desc);
    const int R = c.R;
    // BatchNorm constants of the tower's columns: [mean | invstd | gamma | beta | sum g / N | sum g xhat / N]
    float* BN = c.VEC + fi + fo;
    if (tid < fo) {
        const float inv_n = 1.f / (float)valid_rows(p);
        BN[tid] = p.save_mean[yc0 + tid]; BN[fo + tid] = p.save_invstd[yc0 + tid];
        BN[2 * fo + tid] = p.gamma[t][tid]; BN[3 * fo + tid] = p.beta[t][tid];
        BN[4 * fo + tid] = (float)c.RED[tid] * inv_n; BN[5 * fo + tid] = (float)c.RED[fo + tid] * inv_n;
    }
    __syncthreads();
    // g_yr = snorm * BatchNorm'(g_y1): the gradient at posttrans' output (bias included), all rows of the block; rows up to the strip
    // boundary are zeros
    {
        const int R16 = (R + 15) & ~15;
        RowFeat x = rf_at(tid, fo);
        const RowStep st_ = rf_stride(NT, fo);
        for (; x.r < R16; rf_step(x, st_, fo)) {
            float v = 0.f;
            if (x.r < R) {
                const float xh = (c.Y0S[x.r * ldy + x.f] - BN[x.f]) * BN[fo + x.f];
                v = BN[2 * fo + x.f] * BN[fo + x.f] * (c.Y[x.r * ldy + x.f] - BN[4 * fo + x.f] - xh * BN[5 * fo + x.f]);
                v *= c.FAC[4 * x.r + 3];
            }
            c.Y[x.r * ldy + x.f] = v;
        }
    }
    __syncthreads();
    BLK_STAMP(4);
    float* bpart = p.blk_part + (int64_t)blockIdx.x * p.n_blk_param + t * p.off_tower;
    const int off_wpost = HAS_PRE ? fi * p.ld_pre + fi : 0, off_bpost = off_wpost + fo * p.ld_post;
    const float* wpost = p.w_post[t];
    float gb_post = 0.f;                                       // d b_post of column tid
    if (tid < fo)
        for (int m = 0; m < R; ++m) gb_post += c.Y[m * ldy + tid];
    for (int c0 = 0; c0 < R; c0 += RC) {
        const int rc = min(RC, R - c0), rc16 = (rc + 15) & ~15;
        const float* GY = c.Y + c0 * ldy;
        // the gradient of the rows [h | aggregates]: d agg = sum_s scale_s (g_yr W_post[:, block s]), d h through posttrans' h block:
        // jobs of one 16-column tile of the aggregate blocks or of the h block, all strips of the chunk at once (dagg_chunk)
        // (two strips per pass: the kernel sits at its 128-register cap, four strips' accumulators spill)
        for (int s0 = 0; s0 < rc16; s0 += 32) {
            if (rc16 - s0 >= 32) dagg_chunk<2, HAS_PRE>(p, c, c.XP + s0 * kp, wpost, GY + s0 * ldy, c0 + s0, rc - s0, wave, nw, i16, g);
            else dagg_chunk<1, HAS_PRE>(p, c, c.XP + s0 * kp, wpost, GY + s0 * ldy, c0 + s0, rc - s0, wave, nw, i16, g);
        }
        __syncthreads();
        BLK_STAMP(6);
        // recompute the rows' accumulators (first-occurrence arg tracking), coefficient rows -> LDS; posttrans' input VALUES take the
        // place of their own gradients (an entry of the row belongs to exactly one work item)
        {
            RowFeat x = rf_at(tid, fi);
            const RowStep st_ = rf_stride(NT, fi);
            for (; x.r < rc; rf_step(x, st_, fi)) {
                const int r = c0 + x.r, f = x.f;
                float* xrow = c.XP + x.r * kp;
                float* cf = c.COEF + (x.r * p.L.n_coef) * fi + f;
                const float xin[1] = {c.HB[r * ldh + f]};
                float gdir = 0.f;
                if (HAS_PRE) { gdir = xrow[f]; xrow[f] = xin[0]; }
                const int deg = c.IP[r + 1] - c.IP[r];
                float* out = xrow + p.L.ho + f;
                if (deg == 0) {
                    for (int a = 0; a < p.A; ++a) out[a * fi] = 0.f;
                    for (int q = 0; q < p.L.n_coef; ++q) cf[q * fi] = 0.f;
                    if (HAS_PRE) c.GC[r * ldh + f] = gdir;
                    continue;
                }
                Acc<C, true> acc;
                accumulate_row<C, true, HAS_PRE>(acc, p, c, r, f);
                Coef<C> k;
                float gxin[1];
                make_coef_from<C, O>(k, gxin, acc, p.a, [&](int a, int, float (&gv)[1]) { gv[0] = out[a * fi]; }, deg, xin, 0.f);
                RowStats<1> st;
                row_stats<C, true>(st, acc, (float)deg, p.a);
                for_each_agg<O>(p.a, [&](int a) {
                    float val[1];
                    agg_value<C, true>(val, O::op(p.a, a), O::ch(p.a, a), acc, st, xin);
                    out[a * fi] = val[0];
                });
                if (p.cmap[CF_C0] >= 0) cf[p.cmap[CF_C0] * fi] = k.c0[0];
                if constexpr (C::STATS) {
                    if (p.cmap[CF_CV] >= 0) cf[p.cmap[CF_CV] * fi] = k.cv[0];
                    if (p.cmap[CF_GMAX] >= 0) cf[p.cmap[CF_GMAX] * fi] = k.gmax[0];
                    if (p.cmap[CF_GMIN] >= 0) cf[p.cmap[CF_GMIN] * fi] = k.gmin[0];
                    if (p.cmap[CF_ARG] >= 0) cf[p.cmap[CF_ARG] * fi] = __uint_as_float((unsigned)(k.amax[0] + 1) | ((unsigned)(k.amin[0] + 1) << 16));
                }
#pragma unroll
                for (int ch = 0; ch < C::NCH; ++ch) {
                    if (p.cmap[CF_CS0 + ch] >= 0) cf[p.cmap[CF_CS0 + ch] * fi] = k.cs[ch][0];
                    if constexpr (C::AV) {
                        if (p.cmap[CF_CA0 + ch] >= 0) cf[p.cmap[CF_CA0 + ch] * fi] = k.ca[ch][0];
                    }
                }
                // d x_in of the dx aggregators (x_in = this layer's / tower's input row), with posttrans' h block
                if (HAS_PRE) c.GC[r * ldh + f] = gdir + gxin[0];
                else c.GA[r * ldh + f] += gxin[0];
            }
        }
        __syncthreads();
        BLK_STAMP(7);
        // d W_post partial: D[n][kk] (+)= sum over the chunk's rows of g_yr[m][n] * [h | scale_s * aggregate][m][kk] (later chunks add to
        // the partial buffer itself: L2-resident, the same lane)
        {
            const int ntn = (fo + 15) >> 4, ntk = (p.ld_post + 15) >> 4;
            for (int job = wave; job < ntn * ntk; job += nw) {
                const int tk = job % ntk, tn = job / ntk;
                const int kk = tk * 16 + i16, kc = min(kk, p.ld_post - 1);
                int sblk = -1, col = kc;                      // column kk of [h | s-blocks]: its column in the LDS row and its scaler
                if (kc >= p.h_off) {
                    const int rel = kc - p.h_off;
                    sblk = rel / p.K;
                    col = p.L.ho + rel - sblk * p.K;
                }
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                mma_cols_f(acc, GY + min(tn * 16 + i16, fo - 1), ldy, c.XP + col, kp, sblk >= 0 ? c.FAC + 4 * c0 + sblk : nullptr, rc, g);
                if (kk < p.ld_post) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int nn = tn * 16 + 4 * g + u;
                        if (nn < fo) {
                            float* at = bpart + off_wpost + nn * p.ld_post + kk;
                            *at = c0 == 0 ? acc[u] : *at + acc[u];
                        }
                    }
                }
            }
        }
        // every source row gathers the gradient rows of its out-edges whose destination lies in this chunk, (source, slot) order
        {
            RowFeat x = rf_at(tid, fi);
            const RowStep st_ = rf_stride(NT, fi);
            const int par = (c0 / RC) & 1;
            for (; x.r < R; rf_step(x, st_, fi)) {
                const int u = x.r, f = x.f;
                float a = 0.f;
                int rank = c.CUR[par * R + u];
                for (const int end = c.CP[u + 1]; rank < end; ++rank) {      // (a source's slots ascend with their destinations)
                    const int j = c.CSCI[rank], i = c.DST[j];
                    if (i >= c0 + rc) break;
                    a += edge_grad<C>(p, c, i - c0, j, f, msg_at<HAS_PRE>(p, c, u, i, f));
                }
                if (f == 0) c.CUR[(par ^ 1) * R + u] = rank;
                c.GA[u * ldh + f] += a;
            }
        }
        // d Q: the row sums of the same gradient rows, by the destination's work item
        if constexpr (HAS_PRE) {
            RowFeat x = rf_at(tid, fi);
            const RowStep st_ = rf_stride(NT, fi);
            for (; x.r < rc; rf_step(x, st_, fi)) {
                const int r = c0 + x.r, f = x.f;
                float a = 0.f;
                for (int j = c.IP[r]; j < c.IP[r + 1]; ++j) a += edge_grad<C>(p, c, x.r, j, f, msg_at<HAS_PRE>(p, c, c.SRC[j], r, f));
                c.GB[r * ldh + f] = a;
            }
        }
        __syncthreads();
        BLK_STAMP(8);
    }
    if (tid < fo) bpart[off_bpost + tid] = gb_post;
    BLK_STAMP(9);
    if constexpr (!HAS_PRE) {
        // simple layer: x_src = x_in = h: d h = d x_src + d x_in (both in GA) + the residual's share
        RowFeat x = rf_at(tid, fi);
        const RowStep st_ = rf_stride(NT, fi);
        for (; x.r < R; rf_step(x, st_, fi)) {
            const int64_t at = (int64_t)(c.lo + x.r) * p.F + t * fi + x.f;
            p.g_h[at] = c.GA[x.r * ldh + x.f] + (p.residual ? residual_grad(p, at) : 0.f);
        }
    } else {
        // pretrans adjoint: d h = d P W_s + d Q W_d (+ posttrans' h block, d x_in, residual); d W_pre, d b_pre partials
        const float* wpre = p.w_pre[t];
        const int nstrip = (R + 15) >> 4, nti = (fi + 15) >> 4;
        for (int job = wave; job < nstrip * nti; job += nw) {
            const int ti = job % nti, strip = job / nti;
            const int ii = ti * 16 + i16, m = strip * 16 + i16, mc = min(m, R - 1);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_gs_l(acc, wpre + min(ii, fi - 1), p.ld_pre, c.GA + mc * ldh, fi, g);
            mma_gs_l(acc, wpre + fi + min(ii, fi - 1), p.ld_pre, c.GB + mc * ldh, fi, g);
            if (m < R) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i2 = ti * 16 + 4 * g + u;
                    if (i2 < fi) {
                        const int64_t at = (int64_t)(c.lo + m) * p.F + t * fi + i2;
                        p.g_h[at] = acc[u] + c.GC[m * ldh + i2] + (p.residual ? residual_grad(p, at) : 0.f);
                    }
                }
            }
        }
        for (int job = wave; job < 2 * nti * nti; job += nw) {
            const int tc = job % nti, to = (job / nti) % nti, half = job / (nti * nti);
            const int cc = tc * 16 + i16;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_cols(acc, (half ? c.GB : c.GA) + min(to * 16 + i16, fi - 1), ldh, c.HB + min(cc, fi - 1), ldh, R, g);
            if (cc < fi) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int oo = to * 16 + 4 * g + u;
                    if (oo < fi) bpart[oo * p.ld_pre + half * fi + cc] = acc[u];
                }
            }
        }
        if (tid < fi) {
            float a = 0.f;
            for (int m = 0; m < R; ++m) a += c.GB[m * ldh + tid];
            bpart[fi * p.ld_pre + tid] = a;
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
