// Backward of the sweep for batches of graphs too large for a wave's LDS block (k-NN superpixel graphs, data/superpixels.py:139-145:
// 85 - 150 nodes; SBM graphs): ONE WORKGROUP owns a graph (DgnGraph.gblk_desc: whole graphs, every source of every row inside).
//   phase 1  a wave per DESTINATION row: the row's coefficient vectors (make_coef: c0, cs_c, ca_c -- the arithmetic of the staged
//            backward, dx signs from the forward's aux table, sum_j w_jc from the weights alone: no message is gathered) go to LDS
//            ([rows][slots][F]: 118 x 3 x 66 floats = 93 KB for CIFAR10's list); d x_dst (the row sum of the per-edge gradients, in
//            slot order: no var / std on these lists, hence no message term) and d x_in leave at once.
//   phase 2  a wave per SOURCE row: its out-edges in (source, slot) order -- DgnGraph.csc_order -- each adding
//            c0_i + sum_c (w_jc cs_ic + |w_jc| ca_ic) from the destination's LDS rows; d x_src[u] is written ONCE.
// No float atomics, no [E, F] staging buffer (the staged path writes every per-edge gradient row and reads it back: 1.85x the
// algorithmic bytes on CIFAR10's batch, profiles/r04), the adds in the staged path's own order: run-to-run reproducible.
// Lists without max / min / std / var (their coefficient rows would not fit), messages x_src (+ x_dst), 8-byte lanes, one feature tile.
// Reference semantics: autograd through nets/dgn_layer.py:183-186 (apply_edges gather + update_all reduce), nets/aggregators.py:35-71.
#pragma once
#include "dgn_agg_kernels.hpp"

namespace dgn {

constexpr int kGraphThreads = 1024;

template <class C>
__host__ __device__ constexpr int graph_coef_slots() { return 1 + C::NCH * (C::AV ? 2 : 1); }

// A thread owns a (row slot, feature pair): blockDim / pairs rows are in flight per pass, every lane works, and the index chains of a
// row (row pointer -> slot -> destination / weights) are per-thread loads the compiler batches eight edges deep -- the first version (a
// wave per row, slots broadcast across lanes) spent ~2 us of dependent load latency per row: 0.044 ms on CIFAR10's batch where the
// staged path takes 0.026.  blockIdx.y = a tile of feature pairs: batches of fewer graphs than CUs are split so the chip is covered
// (the row sums of the weights are recomputed per tile: E floats).
// d x_dst of a row = deg c0 + sum_c (sum_j w_jc) cs_c + (sum_j |w_jc|) ca_c: the row sum of the per-edge gradients in closed form.
template <class C, class O>
__global__ __launch_bounds__(kGraphThreads) void agg_bwd_graph(const AggParams p, const int pairs_per_tile) {
    static_assert(C::VEC == 2 && !C::STATS, "8-byte lanes, lists without max / min / std / var");
    constexpr int VEC = 2, NCO = graph_coef_slots<C>(), NCH = C::NCH, NSW = NCH * (C::AV ? 2 : 1);
    extern __shared__ float coef_lds[];
    const int4 d = reinterpret_cast<const int4*>(p.gblk_desc)[blockIdx.x];
    const int lo = d.x, hi = d.y, rows = hi - lo;
    const int pair0 = (int)blockIdx.y * pairs_per_tile;
    const int NP = min(pairs_per_tile, ((p.F + 1) >> 1) - pair0);      // this tile's pairs
    const int Fl = pairs_per_tile * VEC;                                // floats of a coefficient vector in LDS
    const bool alias = p.g_in && p.g_in == p.g_src;
    float* GX = coef_lds + (size_t)p.gblk_rows * NCO * Fl;          // d x_in rows where d x_in aliases d x_src
    float* SW = GX + (alias ? (size_t)p.gblk_rows * Fl : 0);        // sum_j w_jc (and sum_j |w_jc|) per (row, channel)
    const int tid = (int)threadIdx.x;
    const int RS = (int)blockDim.x / NP;                            // row slots
    const int slot = tid / NP, fl = (tid - slot * NP) * VEC, f0 = pair0 * VEC + fl;
    const bool on = slot < RS;
    const bool signs = (p.need & NEED_RECOMP) != 0;          // (the host takes this kernel only with the aux table then)
    // ---- phase 0: sum_j w_jc in slot order (the order of Acc::add: the forward's bits) ----
    if constexpr (NCH > 0) {
        for (int r = tid; r < rows * NCH; r += (int)blockDim.x) {
            const int row = r / NCH, c = r - row * NCH;
            const int beg = p.indptr[lo + row], end = p.indptr[lo + row + 1];
            const float* w = p.w + (int64_t)c * p.ld_w;
            float s = 0.f, sa = 0.f;
#pragma unroll 8
            for (int e = beg; e < end; ++e) {
                const float we = w[e];
                s += we;
                if constexpr (C::AV) sa += fabsf(we);
            }
            SW[row * NSW + c] = s;
            if constexpr (C::AV) SW[row * NSW + NCH + c] = sa;
        }
        __syncthreads();
    }
    // ---- phase 1: destination rows ----
    if (on) {
        for (int row = lo + slot; row < hi; row += RS) {
            const int deg = p.indptr[row + 1] - p.indptr[row];
            const float logd = p.log_deg ? p.log_deg[row] : 0.f;
            float* crow = coef_lds + (size_t)(row - lo) * NCO * Fl + fl;
            const float* grow = p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0);
            const float* sw = SW + (row - lo) * NSW;
            float gxin[VEC] = {0.f, 0.f};
            Coef<C> k;
#pragma unroll
            for (int i = 0; i < VEC; ++i) k.c0[i] = 0.f;
#pragma unroll
            for (int c = 0; c < C::NW; ++c) {
                k.cs[c][0] = 0.f; k.cs[c][1] = 0.f;
                if constexpr (C::AV) { k.ca[c][0] = 0.f; k.ca[c][1] = 0.f; }
            }
            if (deg == 0) {
                // no messages: zero coefficients; the x_in pass-through block still carries its gradient
                if (p.need & NEED_XPASS) {
                    for (int a = 0; a < O::n_agg(p); ++a)
                        if (O::op(p, a) == DGN_AGG_X_IN) {
                            float g[VEC];
                            ldv<VEC>(g, grow + sa_col(p, 0, a));
                            gxin[0] += g[0]; gxin[1] += g[1];
                        }
                }
            } else {
                Acc<C, true> acc;
                acc.init();
                float xin[VEC] = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc.sw[c] = sw[c];
                if constexpr (NCH >= 1 && NCH <= 2) {
                    if (signs) acc_signs_from_aux<C, true>(acc, load_aux_row<VEC>(p.aux + (int64_t)row * p.F + f0));
                }
                make_coef<C, O>(k, gxin, acc, p, grow, deg, xin, logd);
            }
            stv<VEC>(crow, k.c0);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                stv<VEC>(crow + (1 + c) * Fl, k.cs[c]);
                if constexpr (C::AV) stv<VEC>(crow + (1 + NCH + c) * Fl, k.ca[c]);
            }
            if (p.g_dst) {
                float rsum[VEC] = {(float)deg * k.c0[0], (float)deg * k.c0[1]};
                if (deg > 0) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        rsum[0] = fmaf(sw[c], k.cs[c][0], rsum[0]); rsum[1] = fmaf(sw[c], k.cs[c][1], rsum[1]);
                        if constexpr (C::AV) { rsum[0] = fmaf(sw[NCH + c], k.ca[c][0], rsum[0]); rsum[1] = fmaf(sw[NCH + c], k.ca[c][1], rsum[1]); }
                    }
                }
                stv<VEC>(p.g_dst + (int64_t)row * p.ldg_dst + f0, rsum);
            }
            if (alias) stv<VEC>(GX + (size_t)(row - lo) * Fl + fl, gxin);
            else if (p.g_in) stv<VEC>(p.g_in + (int64_t)row * p.ldg_in + f0, gxin);
        }
    }
    __syncthreads();
    // ---- phase 2: source rows, out-edges in (source, slot) order ----
    if (on) {
        for (int u = lo + slot; u < hi; u += RS) {
            const int r0 = p.csc_ptr[u], r1 = p.csc_ptr[u + 1];
            float a[VEC] = {0.f, 0.f};
#pragma unroll 8
            for (int rank = r0; rank < r1; ++rank) {
                const int j = p.csc_order[rank];
                const float* crow = coef_lds + (size_t)(p.dst_csr[j] - lo) * NCO * Fl + fl;
                float gm[VEC];
                ldv<VEC>(gm, crow);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float wq = p.w[(int64_t)c * p.ld_w + j];
                    float cs[VEC];
                    ldv<VEC>(cs, crow + (1 + c) * Fl);
                    gm[0] = fmaf(wq, cs[0], gm[0]); gm[1] = fmaf(wq, cs[1], gm[1]);
                    if constexpr (C::AV) {
                        float ca[VEC];
                        ldv<VEC>(ca, crow + (1 + NCH + c) * Fl);
                        gm[0] = fmaf(fabsf(wq), ca[0], gm[0]); gm[1] = fmaf(fabsf(wq), ca[1], gm[1]);
                    }
                }
                a[0] += gm[0]; a[1] += gm[1];
            }
            if (alias) {
                float gx[VEC];
                ldv<VEC>(gx, GX + (size_t)(u - lo) * Fl + fl);
                a[0] += gx[0]; a[1] += gx[1];
            }
            stv<VEC>(p.g_src + (int64_t)u * p.ldg_src + f0, a);
        }
    }
}

// DGN_OK when launched, 1 when this (list, graph) has no such kernel (the caller runs the staged path)
template <class C, class O>
int launch_backward_graph_cfg(const AggParams& p, hipStream_t stream) {
    if constexpr (C::STATS || C::VEC != 2) {
        return 1;
    } else {
        constexpr int NCO = graph_coef_slots<C>(), NSW = C::NCH * (C::AV ? 2 : 1);
        const int pairs = (p.F + 1) >> 1;
        const bool alias = p.g_in && p.g_in == p.g_src;
        auto lds_of = [&](int ppt) { return ((size_t)p.gblk_rows * (NCO + (alias ? 1 : 0)) * ppt * 2 + (size_t)p.gblk_rows * NSW) * sizeof(float); };
        // feature tiles: enough workgroups to cover the chip on small batches (CIFAR10's 128 graphs: 0.0199 -> 0.0143 ms, PATTERN's:
        // 0.070 -> 0.044); large batches stay whole (two half-width workgroups per CU measured slower: 0.77 vs 0.71 ms on 8192 graphs)
        int tiles = (int)option(OPT_GRAPH_BWD_TILES);
        if (tiles <= 0) {
            tiles = 1;
            while (tiles < 4 && (int64_t)p.n_gblk * tiles < 256 && (pairs + 2 * tiles - 1) / (2 * tiles) >= 8) tiles *= 2;
        }
        tiles = std::min(tiles, pairs);
        const int ppt = (pairs + tiles - 1) / tiles;
        tiles = (pairs + ppt - 1) / ppt;
        const size_t lds = lds_of(ppt);
        if (lds > 160 * 1024 || ((p.need & NEED_RECOMP) && !(p.aux && p.aux_rows))) return 1;
        static bool attr = false;
        if (!attr) {
            DGN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_bwd_graph<C, O>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        AggParams q = p;
        q.stage = nullptr; q.fresh = true; q.seg_add = false;
        hipLaunchKernelGGL((agg_bwd_graph<C, O>), dim3((unsigned)p.n_gblk, (unsigned)tiles), dim3(kGraphThreads), lds, stream, q, ppt);
        DGN_HIP_CHECK(hipGetLastError());
        return DGN_OK;
    }
}

inline int launch_graph_v2(const AggParams& p, hipStream_t stream) {
    static const bool no_hot = getenv("DGN_NO_HOT") != nullptr;
    if (no_hot) return 1;
#define DGN_HOT(NA, OPS, CHS, NS, SCS, N, S, A)                                                                  \
    if (p.n_agg == NA && p.op_pack == OPS && p.ch_pack == CHS && p.n_scalers == NS && p.scaler_pack == SCS &&    \
        p.agg_total == NA && p.agg_offset == 0 && p.n_ch == N) {                                                 \
        using O = StaticOps<NA, OPS, CHS, NS, SCS>;                                                              \
        return launch_backward_graph_cfg<Cfg<2, N, S, A>, O>(p, stream);                                         \
    }
#include "dgn_agg_hot.hpp"
#undef DGN_HOT
    return 1;
}

}  // namespace dgn
