// Block backward of the sweep (include/dgn_hip.h: DgnGraph.blk_cut): batches of small graphs are block-diagonal -- every source of a
// row lies in the row's own graph -- so ONE WAVE that owns a run of whole graphs can accumulate d x_src of its rows in its own LDS rows
// and write each row once.  The staged backward (agg_bwd_short + seg_sum_rows) writes an [E, F] per-edge gradient row and reads it
// back: 1.2-1.7x the algorithmic bytes on the molecule configs (profiles/pmc_traffic.json); this one moves the
// algorithmic bytes, in one kernel.  Same row routine (bwd_short_group with BLK = true): identical per-edge gradient rows, added in
// ascending (source, slot) order -- the order of seg_sum_rows: run-to-run reproducible, and bit-identical to the staged path wherever
// d x_in does not alias d x_src.  Because a wave walks CONSECUTIVE rows, the row pointers of its whole block are one load and the slot
// batch of group g + 1 is requested while group g is worked on: one memory round trip per group instead of three dependent ones.
// Reference semantics: autograd through nets/dgn_layer.py:183-186 (apply_edges gather + update_all reduce).
#pragma once
#include "dgn_agg_kernels.hpp"

namespace dgn {

constexpr int kBlkRowsMax = 56;     // rows of a block: with up to 3 rows of a straddling group either side its row pointers fit one wave

template <class C, class O, bool AUX>
__global__ __launch_bounds__(kWave) void agg_bwd_block(const AggParams p0) {
    extern __shared__ float blk_acc[];
    constexpr int VEC = C::VEC, RB = kBwdShortRows;
    static_assert(RB == 4, "groups of four rows");
    const int64_t n_bins = (p0.n_nodes + p0.blk_bin - 1) / p0.blk_bin;
    const int64_t b = xcd_remap(blockIdx.x, n_bins);
    if (b < 0) return;
    const int N = (int)p0.n_nodes;
    const int lo = p0.blk_cut[b * p0.blk_bin];
    const int hi = p0.blk_cut[min((b + 1) * (int64_t)p0.blk_bin, p0.n_nodes)];
    if (lo >= hi) return;                                       // (a bin without a closed cut of its own: its rows belong to a neighbour)
    AggParams p = p0;
    p.blk_lds = blk_acc; p.blk_lo = lo; p.blk_hi = hi;
    // row pointers of the block, on the global four-row grid (the aux table's layout): lane i holds indptr[lo4 + i]
    const int lo4 = lo & ~(RB - 1), hi4 = min(N, (hi + RB - 1) & ~(RB - 1));
    const int n_ptr = hi4 - lo4 + 1;                            // <= 64 (kBlkRowsMax)
    const int ipv_all = p.indptr[lo4 + min(lane_id(), n_ptr - 1)];
    const int n4 = ((hi - lo) * p.F + 3) >> 2;                 // (the allocation is a multiple of 16 bytes)
    for (int i = lane_id(); i < n4; i += kWave) reinterpret_cast<float4*>(blk_acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int f0 = lane_id() * VEC;
    const bool active = f0 < p.F;
    auto ptr_at = [&](int i) { return bcast_i(ipv_all, min(i, n_ptr - 1)); };
    // The row operands of group g + 1 are requested while group g is worked on where they are few (lists of two upstream blocks per
    // row: c1 backward 0.140 -> 0.128 ms); the headline list's 6 blocks x 4 rows cost 106 more VGPRs (233: two waves per SIMD) and
    // measured 0.207 -> 0.27 ms, four blocks (zinc_json) 0.131 -> 0.159: there the group requests them itself, as in the staged kernel.
    constexpr bool AHEAD = n_gout_blocks<O>() <= 2;
    SlotBatch<C::NCH, C::NW> cur, nxt;
    GroupRows<C, RB, n_gout_blocks<O>(), AUX> rows, rows_nxt;
    cur.load_raw(p, ptr_at(0));
    if constexpr (AHEAD) load_group_rows<C, O, RB, AUX>(rows, p, lo4, f0);
    // a group across the block's boundary is visited by both neighbours, each working on its own rows only
    for (int row0 = lo4; row0 < hi; row0 += RB) {
        const int off = row0 - lo4;
        nxt.load_raw(p, ptr_at(off + RB));                      // (past the end: clamped, unread loads)
        if constexpr (AHEAD) load_group_rows<C, O, RB, AUX>(rows_nxt, p, row0 + RB, f0);
        const int nrows = min(RB, N - row0);
        const int ipv = __shfl(ipv_all, min(off + min(lane_id(), RB), n_ptr - 1), kWave);
        bwd_short_group<C, O, RB, false, false, AUX, true, AHEAD>(p, row0, nrows, f0, active, ipv, &cur, &rows);
        cur = nxt;
        if constexpr (AHEAD) rows = rows_nxt;
    }
    if (active) {
        for (int r = 0; r < hi - lo; ++r) {
            float v[VEC];
            ldv<VEC>(v, blk_acc + r * p.F + f0);
            stv<VEC>(p.g_src + (int64_t)(lo + r) * p.ldg_src + f0, v);
        }
    }
}

struct BlkPlan { int bin, rows; size_t lds; };
// One wave per workgroup.  Its LDS rows hold a bin (blk_bin rows) plus the graph that straddles the bin's upper end (gap - 1 rows):
// 13 KB by default (12 waves per CU), at most kBlkRowsMax rows.
inline bool blk_plan(BlkPlan& out, int64_t F, int gap) {
    const size_t budget = (size_t)option(OPT_BLK_LDS_KB) * 1024;
    // (at most 16 rows of bin beyond the straddling graph: narrow rows would otherwise give a wave 14 groups in a row -- zinc_json,
    //  F = 46: 0.141 ms with 56-row blocks, 0.135 with 50)
    const int rcap = std::min(std::min((int)(budget / (4 * (size_t)F)), kBlkRowsMax), gap + 15);
    const int bin = rcap - gap + 1;
    if (bin < 4) return false;
    out.bin = bin; out.rows = rcap;
    out.lds = (((size_t)rcap * F * 4) + 15) & ~(size_t)15;
    return true;
}

// returns DGN_OK when launched, 1 when this (list, graph) has no block kernel (the caller runs the staged path)
template <class C, class O>
int launch_backward_block_cfg(const AggParams& p, int gap, hipStream_t stream) {
    BlkPlan plan;
    // (small batches: one wave per graph leaves the chip under-filled where the staged kernels start four rows per wave -- c4, 52 k
    // nodes: 0.048 vs 0.033 ms)
    if (p.n_nodes < option(OPT_BLK_MIN_NODES) || !short_rows(p) || option(OPT_BWD_ROWS_PER_WAVE) <= 1 || !blk_plan(plan, p.F, gap)) return 1;
    AggParams q = p;
    q.stage = nullptr; q.csc_pos = nullptr; q.csc_ptr = nullptr; q.fresh = true; q.seg_add = false;
    q.blk_bin = plan.bin; q.blk_rows = plan.rows;
    const int64_t n_bins = (p.n_nodes + plan.bin - 1) / plan.bin;
    const dim3 grid((unsigned)xcd_grid(n_bins)), block(kWave);
    plan.lds += (size_t)std::max<int64_t>(0, option(OPT_BLK_LDS_PAD_KB)) * 1024;      // (what-if: fewer resident waves, same work)
    if constexpr (C::NCH <= 2) {
        if (p.aux) {
            hipLaunchKernelGGL((agg_bwd_block<C, O, true>), grid, block, plan.lds, stream, q);
            DGN_HIP_CHECK(hipGetLastError());
            return DGN_OK;
        }
    }
    hipLaunchKernelGGL((agg_bwd_block<C, O, false>), grid, block, plan.lds, stream, q);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

template <int VEC>
int launch_block_vec(const AggParams& p, int gap, hipStream_t stream) {
    static const bool no_hot = getenv("DGN_NO_HOT") != nullptr;
    if (no_hot) return 1;
#define DGN_HOT(NA, OPS, CHS, NS, SCS, N, S, A)                                                                  \
    if (p.n_agg == NA && p.op_pack == OPS && p.ch_pack == CHS && p.n_scalers == NS && p.scaler_pack == SCS &&    \
        p.agg_total == NA && p.agg_offset == 0 && p.n_ch == N) {                                                 \
        using O = StaticOps<NA, OPS, CHS, NS, SCS>;                                                              \
        if (p.Fv != p.F) {                                                                                       \
            if constexpr (VEC == 2 && odd_width_list(NA, OPS, CHS)) return launch_backward_block_cfg<Cfg<VEC, N, S, A, true>, O>(p, gap, stream); \
            return 1;                                                                                            \
        }                                                                                                        \
        return launch_backward_block_cfg<Cfg<VEC, N, S, A>, O>(p, gap, stream);                                  \
    }
#include "dgn_agg_hot.hpp"
#undef DGN_HOT
    return 1;
}

}  // namespace dgn
