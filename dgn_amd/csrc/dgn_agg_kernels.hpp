// Fused DGN aggregation sweep for gfx950 (MI355X): device code, forward and backward.
//
// One wavefront owns one destination row (x one 64*VEC-wide feature tile).  Lane l holds VEC
// consecutive features, so every gathered source row is one fully coalesced load instruction.
// The row's CSR slots are fetched 64 at a time (source id + per-edge weights, coalesced) and
// broadcast lane -> SGPR with v_readlane, so the gather address is scalar-base + lane offset.
// All aggregators requested for the layer share the single read of each message; the degree
// scalers and the reference's concat order are applied in the epilogue.
//
// The per-edge loop is branch-free: WHAT is accumulated is a compile-time configuration
// Cfg<VEC, NCH, STATS, AV> (NCH weight channels; STATS = sum of squares + max + min; AV = the
// |w|-weighted sums of dirK-av), picked by the host from the aggregator list.  Only the epilogue,
// which runs once per row, interprets the runtime aggregator list.
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
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "dgn_common.hpp"

#ifndef DGN_UNROLL
#define DGN_UNROLL 8
#endif

namespace dgn {

enum : uint32_t {
    NEED_SQ = 2u, NEED_MAX = 4u, NEED_MIN = 8u,
    NEED_XIN = 16u,      // some dx aggregator reads x_in
    NEED_RECOMP = 32u,   // backward must recompute the accumulators
    NEED_M_EMIT = 64u,   // backward emit pass needs the message value (var/std)
    NEED_XPASS = 128u    // the list contains the x_in pass-through (zero-degree rows still carry its gradient)
};

struct AggParams {
    // graph
    const int32_t* indptr;
    const int32_t* src;
    int64_t n_nodes;
    int64_t n_src;        // rows of x_src / g_src (== n_nodes unless the CSR is bipartite)
    int64_t n_edges;
    int32_t hub_threshold;
    int32_t hub_chunk;
    const int32_t* hub_rows;
    const int32_t* hub_chunk_ptr;
    const int32_t* chunk_hub;
    int64_t n_hub;
    int64_t n_chunks;
    // message (all strides are int32: a 32x32->64 multiply is 2 scalar ops, a 64x64 one is 9)
    int32_t F;
    int32_t Ft;  // F / n_towers
    int32_t Fv;  // columns the rows of x_src / x_in really hold (F, or F - 1: DgnMsg.f_valid, kernels with Cfg::ODD)
    const float* x_src;  int32_t ld_src;
    const float* x_dst;  int32_t ld_dst;
    const float* m_edge; int32_t ld_edge;
    const int32_t* edge_type;  // per CSR slot: m_edge is a [n_edge_types, ld_edge] TABLE and slot j adds row edge_type[j] (NULL: row j)
    int32_t n_edge_types;
    int32_t tab_off;           // agg_fwd_short: the table's copy starts this many floats into the dynamic LDS
    const float* x_in;   int32_t ld_in;
    const float* w;      int32_t ld_w;
    const float* log_deg;
    // spec
    int32_t n_agg;
    int32_t agg_total;
    int32_t agg_offset;
    int32_t n_ch;
    int32_t n_scalers;
    int32_t n_towers;
    int64_t tower_stride;  // elements between the column blocks of consecutive towers
    // aggregator / scaler lists packed into scalars (4 bits per op, 3 per channel, 2 per scaler): the
    // epilogue decodes them with scalar ALU ops instead of a chain of dependent kernarg loads
    uint64_t op_pack;
    uint64_t ch_pack;      // 3 bits per channel on purpose, see agg_ch()
    uint32_t scaler_pack;
    float avg_log;
    float eps;
    uint32_t need;
    bool any_av;
    // forward output / backward input
    float* out;           int32_t ld_out;
    const float* g_out;   int32_t ld_gout;
    // backward sinks
    float* g_src;  int32_t ldg_src;
    float* g_dst;  int32_t ldg_dst;
    float* g_edge; int32_t ldg_edge;
    float* g_in;   int32_t ldg_in;
    // workspace (hub rows)
    float* part;          // [n_chunks][n_slots][F]
    float* part_sw;       // [n_chunks][DGN_MAX_CH]
    float* coef;          // [n_hub][n_coef][F]  (backward)
    float* stage;         // [n_edges][F] per-edge gradient rows in csc order (atomic-free backward), or NULL
    // Per-row facts the backward otherwise recomputes from the messages (a second gather of every source row), one byte per (row,
    // feature): bits 0-1 / 2-3 = slot (within the row) of the first maximum / minimum, bits 4-5 / 6-7 = sign of the dx residual of
    // weight channel 0 / 1 (0: zero, 1: positive, 2: negative).  Written by agg_fwd_short for groups of rows with at most kShortDeg
    // in-edges, read by agg_bwd_short for the same groups (lists with at most two channels and no std / var).  NULL = recompute.
    unsigned char* aux;
    // aux_rows: the graph runs the row-per-wave kernels (longer rows) and the list has no max / min: the table then holds the dx signs
    // only, row-major [n_nodes][F] (bits 4-7 of aux_byte), written by fwd_one_row and read by the per-row backward
    int32_t aux_rows;
    int32_t stage_out;    // forward, agg_fwd_short: rows go through the wave's LDS slice (tower-major output, one feature tile)
    bool fresh;           // backward: g_dst / g_in rows are WRITTEN by the row kernel (buffers arrive uninitialised)
    bool seg_add;         // backward: seg_sum_rows adds to g_src (accumulate mode, or g_in aliases g_src) instead of writing it
    const int32_t* csc_ptr;
    const int32_t* csc_pos;
    int32_t n_slots;
    int32_t n_coef;
    // Block backward (agg_bwd_block): blk_cut[i] = the last CLOSED cut <= i, i in [0, N] -- a cut c is closed when no edge joins a node
    // below c to a node at or above it (graph boundaries of a batch); workgroup b owns rows [blk_cut[b * blk_bin],
    // blk_cut[min((b + 1) * blk_bin, N)]): every source of every one of its rows lies in that range, so d x_src of the range is
    // accumulated in LDS (ds_add_f32, any order) and written once -- no [E, F] staging buffer, no csc positions, no seg_sum_rows.
    const int32_t* blk_cut;
    int32_t blk_bin;
    int32_t blk_rows;     // LDS rows per workgroup (blk_bin + the largest gap between closed cuts - 1 fits)
    // Graph backward (dgn_agg_graph.hpp): gblk_desc [n_gblk][4] = first row, end row, first slot, end slot of blocks of WHOLE graphs (one
    // workgroup each, at most gblk_rows rows), csc_order[rank] = the CSR slot of (source, slot) rank `rank`, dst_csr[slot] = its row
    const int32_t* gblk_desc; const int32_t* csc_order; const int32_t* dst_csr;
    int32_t n_gblk, gblk_rows;
    // set per workgroup by the kernel (a copy of the parameter block):
    float* blk_lds;       // [blk_rows][F] accumulator of d x_src (+ d x_in when g_in aliases g_src)
    int32_t blk_lo, blk_hi;
};

// accumulator slot ids in the hub workspace
constexpr int SLOT_SUM = 0, SLOT_SQ = 1, SLOT_MAX = 2, SLOT_MIN = 3, SLOT_AMAX = 4, SLOT_AMIN = 5, SLOT_W0 = 6;
// coefficient slot ids (backward hub path)
constexpr int COEF_C0 = 0, COEF_CV = 1, COEF_GMAX = 2, COEF_GMIN = 3, COEF_AMAX = 4, COEF_AMIN = 5, COEF_W0 = 6;

// ODD_: the rows of x_src / x_in hold F - 1 columns at their own (odd) stride (DgnMsg.f_valid): the last lane of a row shifts its pair
template <int VEC_, int NCH_, bool STATS_, bool AV_, bool ODD_ = false>
struct Cfg {
    static constexpr bool ODD = ODD_;
    static constexpr int VEC = VEC_;
    static constexpr int NCH = NCH_;
    static constexpr int NW = NCH_ > 0 ? NCH_ : 1;   // storage size (no zero-length arrays)
    static constexpr bool STATS = STATS_;
    static constexpr bool AV = AV_;
};

template <class C, bool TRACK>
struct Acc {
    static constexpr int VEC = C::VEC, NCH = C::NCH, NW = C::NW;
    float sum[VEC];
    float sq[C::STATS ? VEC : 1];
    float mx[C::STATS ? VEC : 1];
    float mn[C::STATS ? VEC : 1];
    int amax[(C::STATS && TRACK) ? VEC : 1];
    int amin[(C::STATS && TRACK) ? VEC : 1];
    float ws[NW][VEC];
    float wa[C::AV ? NW : 1][C::AV ? VEC : 1];
    float sw[NW];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum[i] = 0.f;
        if constexpr (C::STATS) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sq[i] = 0.f;
                mx[i] = -INFINITY;
                mn[i] = INFINITY;
                if constexpr (TRACK) { amax[i] = -1; amin[i] = -1; }
            }
        }
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            sw[c] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                ws[c][i] = 0.f;
                if constexpr (C::AV) wa[c][i] = 0.f;
            }
        }
    }

    // one message; pos = CSR slot (for first-occurrence arg tracking).  No branches.
    __device__ __forceinline__ void add(const float (&m)[VEC], const float (&wk)[NW], int pos) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum[i] += m[i];
        if constexpr (C::STATS) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sq[i] = fmaf(m[i], m[i], sq[i]);
                if constexpr (TRACK) {
                    if (m[i] > mx[i]) { mx[i] = m[i]; amax[i] = pos; }
                    if (m[i] < mn[i]) { mn[i] = m[i]; amin[i] = pos; }
                } else {
                    mx[i] = fmaxf(mx[i], m[i]);
                    mn[i] = fminf(mn[i], m[i]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            sw[c] += wk[c];
#pragma unroll
            for (int i = 0; i < VEC; ++i) ws[c][i] = fmaf(wk[c], m[i], ws[c][i]);
            if constexpr (C::AV) {
                const float a = fabsf(wk[c]);
#pragma unroll
                for (int i = 0; i < VEC; ++i) wa[c][i] = fmaf(a, m[i], wa[c][i]);
            }
        }
    }

    // merge a later partial (slot order preserved: strict compare keeps the first occurrence)
    __device__ __forceinline__ void merge(const Acc& o) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum[i] += o.sum[i];
        if constexpr (C::STATS) {
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
            for (int i = 0; i < VEC; ++i) {
                ws[c][i] += o.ws[c][i];
                if constexpr (C::AV) wa[c][i] += o.wa[c][i];
            }
        }
    }
};

// message of slot e coming from node s:  x_src[s] + x_dst[row] + m_edge[e]   (rounded in this order)
template <int VEC, bool ODD = false>
__device__ __forceinline__ void load_msg(float (&m)[VEC], const AggParams& p, int s, int e, int f0, const float (&xd)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) m[i] = xd[i];
    if (p.x_src) {
        float t[VEC];
        ldx<VEC, ODD>(t, p.x_src + (int64_t)s * p.ld_src, f0, p.Fv);
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

// The gather loops do NOT use load_msg(): a value consumed in the same (conditional) block as its load makes the
// compiler put `s_waitcnt vmcnt(0)` right behind every load, i.e. the gathers of a row run one full memory round
// trip after the other (that is how these kernels ran until round 1's ISA check: 2.5 TB/s of gathers on C5,
// 0.25 ms for the ZINC-12k forward).  They first ISSUE a group of loads into a register tile and consume it
// afterwards.  MsgSrc is the gathered part when there is exactly one (x_src rows by source id, or m_edge rows by
// slot); messages with both parts use half the tile for each.
// VEC columns of a row of x_src / x_in from column f0 on.  ODD (rows of an odd width F - 1 = Fv at their own stride, 8-byte lanes): the lane
// of the last pair loads (x[F - 3], x[F - 2]) -- inside the row, nothing past it -- and hands on (x[F - 2], 0): the zero-padded row without
// the padded copy.  A compile-time flag: as a run-time select in every gather it cost the even widths 3 - 23 % of the forward sweep.
template <int VEC, bool ODD>
__device__ __forceinline__ void ldx(float (&d)[VEC], const float* row, int f0, int Fv) {
    if constexpr (ODD && VEC == 2) {
        const bool last = f0 + 1 >= Fv;
        const float2 t = *reinterpret_cast<const float2*>(row + (last ? f0 - 1 : f0));
        d[0] = last ? t.y : t.x;
        d[1] = last ? 0.f : t.y;
    } else {
        ldv<VEC>(d, row + f0);
    }
}

template <int VEC, bool ODD = false>
struct MsgSrc {
    const float* base;
    int ld, Fv;
    bool by_src, both;
    __device__ __forceinline__ explicit MsgSrc(const AggParams& p)
        : base(p.x_src ? p.x_src : p.m_edge), ld(p.x_src ? p.ld_src : p.ld_edge), Fv(p.Fv), by_src(p.x_src != nullptr),
          both(p.x_src != nullptr && p.m_edge != nullptr) {}
    __device__ __forceinline__ void load(float (&t)[VEC], int s, int e, int f0) const {
        ldx<VEC, ODD>(t, base + (int64_t)(by_src ? s : e) * ld, f0, Fv);
    }
};

// 64 CSR slots (source id + weights), one per lane
template <int NCH, int NW>
struct SlotBatch {
    int src;
    int et;        // row of the slot's m_edge term: the slot itself, or its edge type (table mode)
    float w[NW];
    __device__ __forceinline__ void load(const AggParams& p, int base, int end) {
        const int e = base + lane_id();
        const bool in = e < end;
        src = in ? p.src[e] : 0;
        et = p.edge_type ? (in ? p.edge_type[e] : 0) : e;
#pragma unroll
        for (int c = 0; c < NW; ++c) w[c] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) w[c] = in ? p.w[(int64_t)c * p.ld_w + e] : 0.f;
    }
    // prefetch form: branch-free loads clamped into the slot arrays (lanes beyond the range hold some other slot's values and are never
    // read: the consumer broadcasts lanes below its slot count only); no edge-type table
    __device__ __forceinline__ void load_raw(const AggParams& p, int base) {
        const int e = min(base + lane_id(), (int)p.n_edges - 1);
        src = p.src[e];
        et = e;
#pragma unroll
        for (int c = 0; c < NW; ++c) w[c] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) w[c] = p.w[(int64_t)c * p.ld_w + e];
    }
    __device__ __forceinline__ void weights(float (&wk)[NW], int k) const {
#pragma unroll
        for (int c = 0; c < NW; ++c) wk[c] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) wk[c] = bcast_f(w[c], k);
    }
};

// accumulate the cnt (<= 64) slots of one loaded slot batch (slots base .. base + cnt - 1); active lanes only.
// Groups of U gathers in flight; the last (partial) group is predicated, NOT a one-at-a-time loop: molecule
// rows have 2-3 slots, and a scalar remainder loop would serialise their gather latencies.  Slot k always
// exists (loop condition), so its load is unconditional: the wait for the slot batch then sits on the common
// path instead of behind every conditional load.
template <class C, bool TRACK>
__device__ __forceinline__ void accumulate_batch(Acc<C, TRACK>& acc, const AggParams& p, const MsgSrc<C::VEC, C::ODD>& src,
                                                 const SlotBatch<C::NCH, C::NW>& b, int base, int cnt, int f0,
                                                 const float (&xd)[C::VEC]) {
    constexpr int VEC = C::VEC, U = DGN_UNROLL, H = U / 2;
    float t[U][VEC];
    auto use = [&](int slot, const float (&m)[VEC]) {
        float wk[C::NW];
        b.weights(wk, slot);
        acc.add(m, wk, base + slot);
    };
    if (!src.both) {
        for (int k = 0; k < cnt; k += U) {
            // two half-groups: a short row (<= U/2 slots) pays U/2 + 1 uniform checks instead of U
            const bool second = k + H < cnt;
#pragma unroll
            for (int u = 0; u < H; ++u)
                if (u == 0 || k + u < cnt) src.load(t[u], bcast_i(b.src, k + u), base + k + u, f0);
            if (second) {
#pragma unroll
                for (int u = H; u < U; ++u)
                    if (k + u < cnt) src.load(t[u], bcast_i(b.src, k + u), base + k + u, f0);
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                if (u == 0 || k + u < cnt) {
                    float m[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) m[i] = xd[i] + t[u][i];
                    use(k + u, m);
                }
            }
            if (second) {
#pragma unroll
                for (int u = H; u < U; ++u) {
                    if (k + u < cnt) {
                        float m[VEC];
#pragma unroll
                        for (int i = 0; i < VEC; ++i) m[i] = xd[i] + t[u][i];
                        use(k + u, m);
                    }
                }
            }
        }
    } else {
        for (int k = 0; k < cnt; k += H) {      // x_src part in t[0..H), m_edge part in t[H..U)
#pragma unroll
            for (int u = 0; u < H; ++u) {
                if (u == 0 || k + u < cnt) {
                    ldx<VEC, C::ODD>(t[u], p.x_src + (int64_t)bcast_i(b.src, k + u) * p.ld_src, f0, p.Fv);
                    ldv<VEC>(t[H + u], p.m_edge + (int64_t)bcast_i(b.et, k + u) * p.ld_edge + f0);
                }
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                if (u == 0 || k + u < cnt) {
                    float m[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) m[i] = (xd[i] + t[u][i]) + t[H + u][i];
                    use(k + u, m);
                }
            }
        }
    }
}

// accumulate CSR slots [beg, end) of one destination row
template <class C, bool TRACK>
__device__ __forceinline__ void accumulate_range(Acc<C, TRACK>& acc, const AggParams& p, int beg, int end, int f0,
                                                 bool active, const float (&xd)[C::VEC]) {
    const MsgSrc<C::VEC, C::ODD> src(p);
    for (int base = beg; base < end; base += kWave) {
        SlotBatch<C::NCH, C::NW> b;
        b.load(p, base, end);
        // ONE exec-mask region for the whole batch (v_readlane ignores exec), not one per load
        if (active) accumulate_batch<C, TRACK>(acc, p, src, b, base, min(kWave, end - base), f0, xd);
    }
}

__device__ __forceinline__ int agg_op(const AggParams& p, int a) { return (int)((p.op_pack >> (4 * a)) & 15u); }
// 3-bit fields: with a 2-bit field the compiler can prove c in [0,3], folds the compare chains of
// pick_channel()/make_coef() into dynamic register-array indexing and sends the accumulators to scratch
__device__ __forceinline__ int agg_ch(const AggParams& p, int a) { return (int)((p.ch_pack >> (3 * a)) & 7u); }
__device__ __forceinline__ int scaler_kind(const AggParams& p, int s) { return (int)((p.scaler_pack >> (2 * s)) & 3u); }

// ---- aggregator-list policies -------------------------------------------------------------------
// DynOps reads the packed lists from the kernel arguments (any list).  StaticOps<...> bakes a list into the
// kernel: the per-row epilogue / coefficient code is then fully unrolled with constant ops, i.e. no decode,
// no branches (about 3x fewer instructions per row on the molecule configs).  The host uses a StaticOps
// kernel when the launched list matches one of the hot lists of the reference's configs (dgn_agg_hot.hpp).
struct DynOps {
    static constexpr bool kStatic = false;
    static constexpr int NA = DGN_MAX_AGG;
    static __device__ __forceinline__ int n_agg(const AggParams& p) { return p.n_agg; }
    static __device__ __forceinline__ int n_scalers(const AggParams& p) { return p.n_scalers; }
    static __device__ __forceinline__ int op(const AggParams& p, int a) { return agg_op(p, a); }
    static __device__ __forceinline__ int ch(const AggParams& p, int a) { return agg_ch(p, a); }
    static __device__ __forceinline__ int scaler(const AggParams& p, int s) { return scaler_kind(p, s); }
};
template <int NA_, uint64_t OPS, uint64_t CHS, int NS_, uint32_t SCS>
struct StaticOps {
    static constexpr bool kStatic = true;
    static constexpr int NA = NA_;
    static constexpr uint64_t kOps = OPS, kChs = CHS;
    static constexpr int kNS = NS_;
    static constexpr uint32_t kScs = SCS;
    static __device__ __forceinline__ constexpr int n_agg(const AggParams&) { return NA_; }
    static __device__ __forceinline__ constexpr int n_scalers(const AggParams&) { return NS_; }
    static __device__ __forceinline__ constexpr int op(const AggParams&, int a) { return (int)((OPS >> (4 * a)) & 15u); }
    static __device__ __forceinline__ constexpr int ch(const AggParams&, int a) { return (int)((CHS >> (3 * a)) & 7u); }
    static __device__ __forceinline__ constexpr int scaler(const AggParams&, int s) { return (int)((SCS >> (2 * s)) & 3u); }
};

// loop over the aggregators: a plain loop for DynOps, fully unrolled for StaticOps
template <class O, int STEP = 1, class Fn>
__device__ __forceinline__ void for_each_agg(const AggParams& p, Fn&& fn) {
    if constexpr (O::kStatic) {
#pragma unroll
        for (int a = 0; a < O::NA; a += STEP) fn(a);
    } else {
        for (int a = 0; a < O::n_agg(p); a += STEP) fn(a);
    }
}

__device__ __forceinline__ float scaler_factor(int kind, float logd, float avg) {
    if (kind == DGN_SCALE_AMPLIFICATION) return logd / avg;
    if (kind == DGN_SCALE_ATTENUATION) return avg / logd;
    return 1.f;
}

// output column of (scaler s, aggregator a, feature f) in the [T][S][A][Ft] layout, split into a
// per-lane part (tower block + feature inside the tower, computed once per wave) and a wave-uniform part
__device__ __forceinline__ int64_t lane_col(const AggParams& p, int f) {
    if (p.n_towers == 1) return f;          // (uniform) no tower blocks
    int t = 0;                              // f / Ft by compares (a per-lane integer division costs ~35 VALU ops)
    for (int q = 1; q < p.n_towers; ++q) t += (f >= q * p.Ft) ? 1 : 0;
    const int ft = f - t * p.Ft;
    return (int64_t)t * p.tower_stride + ft;
}
// 32-bit on purpose (S*A*F < 2^31 is validated on the host): one scalar multiply instead of a 64x64 one
__device__ __forceinline__ int sa_col(const AggParams& p, int s, int a) {
    return (s * p.agg_total + p.agg_offset + a) * p.Ft;
}

// r = sum_j w_j m_j - (sum_j w_j) x  with the reference's two roundings (aggregators.py:52/:59)
__device__ __forceinline__ float dx_residual(float ws, float sw, float x) {
    const float t = sw * x;
    return ws - t;
}

// Row statistics shared by several aggregators, with the reference's roundings
// (aggregators.py:8-9, :24-28, :20-21): mean = sum/d, var = relu(sq/d - mean*mean), std = sqrt(var + eps).
template <int VEC>
struct RowStats {
    float mean[VEC], rawvar[VEC], var[VEC], sd[VEC];
};

template <class C, bool TRACK>
__device__ __forceinline__ void row_stats(RowStats<C::VEC>& st, const Acc<C, TRACK>& acc, float d, const AggParams& p) {
#pragma unroll
    for (int i = 0; i < C::VEC; ++i) {
        st.mean[i] = __fdiv_rn(acc.sum[i], d);
        st.rawvar[i] = 0.f; st.var[i] = 0.f; st.sd[i] = 0.f;
    }
    if constexpr (C::STATS) {
        if (p.need & NEED_SQ) {
#pragma unroll
            for (int i = 0; i < C::VEC; ++i) {
                const float ms = __fdiv_rn(acc.sq[i], d);
                st.rawvar[i] = ms - st.mean[i] * st.mean[i];   // -ffp-contract=off: two roundings, as torch
                st.var[i] = fmaxf(st.rawvar[i], 0.f);
                st.sd[i] = __fsqrt_rn(st.var[i] + p.eps);
            }
        }
    }
}

// Channel selection with compares.  NOTE: default 0 + compare chain, NOT "start from channel 0": the
// latter is folded by the compiler into acc.ws[c] (dynamic indexing), which sends the whole accumulator
// struct to scratch memory.
template <class C, bool TRACK>
__device__ __forceinline__ void pick_channel(float (&wsv)[C::VEC], float (&wav)[C::VEC], float& swv, const Acc<C, TRACK>& acc, int c) {
    swv = 0.f;
#pragma unroll
    for (int i = 0; i < C::VEC; ++i) { wsv[i] = 0.f; wav[i] = 0.f; }
#pragma unroll
    for (int cc = 0; cc < C::NCH; ++cc) {
        if (cc == c) {
            swv = acc.sw[cc];
#pragma unroll
            for (int i = 0; i < C::VEC; ++i) {
                wsv[i] = acc.ws[cc][i];
                if constexpr (C::AV) wav[i] = acc.wa[cc][i];
            }
        }
    }
}

// value of one aggregator from the row's accumulators
// (uniform if-chain on purpose: a switch over plain values is turned into a scratch lookup table)
template <class C, bool TRACK>
__device__ __forceinline__ void agg_value(float (&val)[C::VEC], int op, int c, const Acc<C, TRACK>& acc,
                                          const RowStats<C::VEC>& st, const float (&xin)[C::VEC]) {
    constexpr int VEC = C::VEC;
    if (op == DGN_AGG_MEAN) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) val[i] = st.mean[i];
        return;
    }
    if (op == DGN_AGG_SUM) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) val[i] = acc.sum[i];
        return;
    }
    if (op == DGN_AGG_VAR) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) val[i] = st.var[i];
        return;
    }
    if (op == DGN_AGG_STD) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) val[i] = st.sd[i];
        return;
    }
    if (op == DGN_AGG_MAX || op == DGN_AGG_MIN) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            val[i] = 0.f;
            if constexpr (C::STATS) val[i] = op == DGN_AGG_MAX ? acc.mx[i] : acc.mn[i];
        }
        return;
    }
    if (op == DGN_AGG_X_IN) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) val[i] = xin[i];
        return;
    }
    float wsv[VEC], wav[VEC], swv;
    pick_channel<C, TRACK>(wsv, wav, swv, acc, c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float v;
        if (op == DGN_AGG_DIR_AV) v = wav[i];
        else if (op == DGN_AGG_DIR_WSUM) v = wsv[i];
        else {
            v = dx_residual(wsv[i], swv, xin[i]);
            if (op == DGN_AGG_DIR_DX) v = fabsf(v);
        }
        val[i] = v;
    }
}

// write one finished row: aggregator values x scalers in the reference concat order.
// orow already includes the lane's column part; xin / logd were loaded by the caller (early).
template <class C, class O = DynOps, bool TRACK = false>
__device__ __forceinline__ void write_row(const Acc<C, TRACK>& acc, const AggParams& p, float* orow, int deg,
                                          const float (&xin)[C::VEC], float logd, bool to_global = false) {      // to_global: streaming stores
    constexpr int VEC = C::VEC;
    if (deg == 0) {      // no messages: zeros (DGL fills such rows from the zero initializer); the x_in block is still x_in
        float z[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) z[i] = 0.f;
        for (int s = 0; s < O::n_scalers(p); ++s)
            for (int a = 0; a < O::n_agg(p); ++a)
                if (to_global) stv_stream<VEC>(orow + sa_col(p, s, a), (O::op(p, a) == DGN_AGG_X_IN && O::scaler(p, s) == DGN_SCALE_IDENTITY) ? xin : z);
                else stv<VEC>(orow + sa_col(p, s, a), (O::op(p, a) == DGN_AGG_X_IN && O::scaler(p, s) == DGN_SCALE_IDENTITY) ? xin : z);
        return;
    }
    const float d = (float)deg;
    RowStats<VEC> st;
    row_stats<C, TRACK>(st, acc, d, p);
    float fac[DGN_MAX_SCALERS];
#pragma unroll
    for (int s = 0; s < DGN_MAX_SCALERS; ++s) fac[s] = s < O::n_scalers(p) ? scaler_factor(O::scaler(p, s), logd, p.avg_log) : 1.f;
    for_each_agg<O>(p, [&](int a) {
        float val[VEC];
        agg_value<C, TRACK>(val, O::op(p, a), O::ch(p, a), acc, st, xin);
#pragma unroll
        for (int s = 0; s < DGN_MAX_SCALERS; ++s) {
            if (s < O::n_scalers(p)) {
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = O::scaler(p, s) == DGN_SCALE_IDENTITY ? val[i] : val[i] * fac[s];
                if (to_global) stv_stream<VEC>(orow + sa_col(p, s, a), o);
                else stv<VEC>(orow + sa_col(p, s, a), o);      // (orow may point into the wave's LDS staging slice, see agg_fwd_short)
            }
        }
    });
}

// AggParams.aux: the byte of one (row, feature) from the row's tracked accumulators (arg max / min tracked as positions within the row)
template <class C, bool TRACK>
__device__ __forceinline__ unsigned aux_byte(const Acc<C, TRACK>& acc, int i, float xin_i) {
    unsigned b = 0;
    if constexpr (C::STATS && TRACK) b = ((unsigned)acc.amax[i] & 3u) | (((unsigned)acc.amin[i] & 3u) << 2);      // (tracked as positions within the row)
#pragma unroll
    for (int c = 0; c < (C::NCH < 2 ? C::NCH : 2); ++c) {
        const float r = dx_residual(acc.ws[c][i], acc.sw[c], xin_i);
        b |= (r > 0.f ? 1u : (r < 0.f ? 2u : 0u)) << (4 + 2 * c);
    }
    return b;
}
template <int VEC>
__device__ __forceinline__ void store_aux_row(unsigned char* at, const unsigned (&b)[VEC]) {
    if constexpr (VEC == 1) *at = (unsigned char)b[0];
    else if constexpr (VEC == 2) *reinterpret_cast<unsigned short*>(at) = (unsigned short)(b[0] | (b[1] << 8));
    else *reinterpret_cast<unsigned*>(at) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
}
template <int VEC>
__device__ __forceinline__ unsigned load_aux_row(const unsigned char* at) {
    if constexpr (VEC == 1) return *at;
    else if constexpr (VEC == 2) return *reinterpret_cast<const unsigned short*>(at);
    else return *reinterpret_cast<const unsigned*>(at);
}
// the fields of an accumulator the coefficient code reads for a dx aggregator, from the aux byte(s) `w` (byte i = feature i): a stand-in
// of sum_j w_jc m_j with the residual's sign (x_in then reads as zero)
template <class C, bool TRACK>
__device__ __forceinline__ void acc_signs_from_aux(Acc<C, TRACK>& acc, unsigned w) {
#pragma unroll
    for (int i = 0; i < C::VEC; ++i) {
        const unsigned ab = (w >> (8 * i)) & 0xffu;
#pragma unroll
        for (int c = 0; c < (C::NCH < 2 ? C::NCH : 2); ++c) {
            const unsigned code = (ab >> (4 + 2 * c)) & 3u;
            acc.ws[c][i] = code == 1u ? 1.f : (code == 2u ? -1.f : 0.f);
        }
    }
}
// The table is laid out per group of four rows and per lane: a lane's bytes of the four rows are adjacent (4 VEC bytes at
// group * 4 F + 4 f0), so a group is ONE store / load instruction of consecutive 8-byte (VEC = 2) lanes.
__device__ __forceinline__ int64_t aux_offset(const AggParams& p, int row0, int f0) { return (int64_t)(row0 >> 2) * 4 * p.F + 4 * f0; }
template <int VEC>
__device__ __forceinline__ void store_aux_group(unsigned char* at, const unsigned (&b)[4][VEC]) {
    unsigned w[4];                           // VEC bytes per row, packed
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w[r] = 0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) w[r] |= b[r][i] << (8 * i);
    }
    if constexpr (VEC == 1) *reinterpret_cast<unsigned*>(at) = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
    else if constexpr (VEC == 2) *reinterpret_cast<uint2*>(at) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
    else *reinterpret_cast<uint4*>(at) = make_uint4(w[0], w[1], w[2], w[3]);
}
// w[r] = the VEC bytes of row r of the group, byte i = feature f0 + i
template <int VEC>
__device__ __forceinline__ void load_aux_group(unsigned (&w)[4], const unsigned char* at) {
    if constexpr (VEC == 1) {
        const unsigned v = *reinterpret_cast<const unsigned*>(at);
        w[0] = v & 0xffu; w[1] = (v >> 8) & 0xffu; w[2] = (v >> 16) & 0xffu; w[3] = v >> 24;
    } else if constexpr (VEC == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(at);
        w[0] = v.x & 0xffffu; w[1] = v.x >> 16; w[2] = v.y & 0xffffu; w[3] = v.y >> 16;
    } else {
        const uint4 v = *reinterpret_cast<const uint4*>(at);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
}

// ---- hub workspace I/O ----------------------------------------------------------------------

template <class C, bool TRACK>
__device__ __forceinline__ void store_partial(const Acc<C, TRACK>& acc, const AggParams& p, int64_t chunk, int f0, bool active) {
    constexpr int VEC = C::VEC;
    float* base = p.part + chunk * (int64_t)p.n_slots * p.F;
    if (active) {
        stv<VEC>(base + (int64_t)SLOT_SUM * p.F + f0, acc.sum);
        if constexpr (C::STATS) {
            stv<VEC>(base + (int64_t)SLOT_SQ * p.F + f0, acc.sq);
            stv<VEC>(base + (int64_t)SLOT_MAX * p.F + f0, acc.mx);
            stv<VEC>(base + (int64_t)SLOT_MIN * p.F + f0, acc.mn);
            if constexpr (TRACK) {
                stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)SLOT_AMAX * p.F + f0), acc.amax);
                stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)SLOT_AMIN * p.F + f0), acc.amin);
            }
        }
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) {
            stv<VEC>(base + (int64_t)(SLOT_W0 + 2 * c) * p.F + f0, acc.ws[c]);
            if constexpr (C::AV) stv<VEC>(base + (int64_t)(SLOT_W0 + 2 * c + 1) * p.F + f0, acc.wa[c]);
        }
    }
    if (lane_id() == 0 && blockIdx.y == 0) {
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) p.part_sw[chunk * DGN_MAX_CH + c] = acc.sw[c];
    }
}

template <class C, bool TRACK>
__device__ __forceinline__ void load_partial(Acc<C, TRACK>& acc, const AggParams& p, int64_t chunk, int f0, bool active) {
    constexpr int VEC = C::VEC;
    const float* base = p.part + chunk * (int64_t)p.n_slots * p.F;
    acc.init();
    if (active) {
        ldv<VEC>(acc.sum, base + (int64_t)SLOT_SUM * p.F + f0);
        if constexpr (C::STATS) {
            ldv<VEC>(acc.sq, base + (int64_t)SLOT_SQ * p.F + f0);
            ldv<VEC>(acc.mx, base + (int64_t)SLOT_MAX * p.F + f0);
            ldv<VEC>(acc.mn, base + (int64_t)SLOT_MIN * p.F + f0);
            if constexpr (TRACK) {
                ldvi<VEC>(acc.amax, reinterpret_cast<const int*>(base + (int64_t)SLOT_AMAX * p.F + f0));
                ldvi<VEC>(acc.amin, reinterpret_cast<const int*>(base + (int64_t)SLOT_AMIN * p.F + f0));
            }
        }
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) {
            ldv<VEC>(acc.ws[c], base + (int64_t)(SLOT_W0 + 2 * c) * p.F + f0);
            if constexpr (C::AV) ldv<VEC>(acc.wa[c], base + (int64_t)(SLOT_W0 + 2 * c + 1) * p.F + f0);
        }
    }
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) acc.sw[c] = p.part_sw[chunk * DGN_MAX_CH + c];
}

// ---- forward kernels --------------------------------------------------------------------------

// one destination row, start to finish (row pointers -> slot batches -> gathers -> epilogue)
template <class C, class O>
__device__ __forceinline__ void fwd_one_row(const AggParams& p, int row, int f0, bool active, float* orow_override = nullptr) {
    constexpr int VEC = C::VEC;
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    const int deg = end - beg;
    if (deg > p.hub_threshold) return;  // hub row: slice + combine kernels own it
    // everything the epilogue needs is requested before the gather loop, so its latency hides there
    float xd[VEC], xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xd[i] = 0.f; xin[i] = 0.f; }
    const float logd = p.log_deg ? p.log_deg[row] : 0.f;
    if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
    if (active && (p.need & NEED_XIN)) ldx<VEC, C::ODD>(xin, p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
    Acc<C, false> acc;
    acc.init();
    accumulate_range<C, false>(acc, p, beg, end, f0, active, xd);
    if (active) write_row<C, O>(acc, p, orow_override ? orow_override : p.out + (int64_t)row * p.ld_out + lane_col(p, f0), deg, xin, logd, orow_override == nullptr);
    if constexpr (!C::STATS && C::NCH >= 1 && C::NCH <= 2) {
        if (active && p.aux_rows) {          // the dx signs for the backward (AggParams.aux_rows)
            unsigned ab[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) ab[i] = aux_byte<C, false>(acc, i, xin[i]);
            store_aux_row<VEC>(p.aux + (int64_t)row * p.F + f0, ab);
        }
    }
}

template <class C, class O = DynOps>
__global__ __launch_bounds__(256) void agg_fwd_rows(const AggParams p) {
    const int wpb = blockDim.x >> 6;   // 1 (long rows: a finished row frees its slot at once) or 4 (short rows: dispatch-rate bound)
    const int64_t n_blocks = (p.n_nodes + wpb - 1) / wpb;
    const int64_t lb = xcd_remap(blockIdx.x, n_blocks);
    if (lb < 0) return;
    const int64_t row64 = lb * wpb + (threadIdx.x >> 6);
    if (row64 >= p.n_nodes) return;
    const int f0 = (blockIdx.y * kWave + lane_id()) * C::VEC;
    fwd_one_row<C, O>(p, uniform_i((int)row64), f0, f0 < p.F);
}

// ---- short-row kernels: kShortRows consecutive destination rows per wavefront -------------------------------
// A molecule row has 2-3 in-edges: one wave per row is a chain of four dependent memory round trips (row
// pointers -> slot batch -> gathers -> store acknowledgement) around ~1 000 cycles of work, and with 8 waves per
// SIMD the chip mostly waits (measured on ZINC-12k: 0.25 ms, of which 0.11 ms remain without the stores and
// 0.05 ms with neither gathers nor epilogue; a 128-float row takes barely longer than a 64-float one).  Here a
// wave takes kShortRows rows whose slots -- contiguous in the CSR -- fit ONE slot batch: one row-pointer load, one
// batch load, all gathers of all rows in flight together, then the rows are finished one after the other from
// registers while the next row's side inputs are already on their way and the previous row's stores drain.
// Groups with a row of more than kShortDeg slots fall back to the row-at-a-time routine, so the kernels
// are correct on any graph; the host picks them when the average in-degree is small (short_rows()).
constexpr int kShortRows = 4, kShortDeg = 4;

// side inputs of one row: x_dst (message part), x_in (epilogue), log-degree
template <int VEC, bool ODD = false>
struct RowSide {
    float xd[VEC], xin[VEC], logd;
    __device__ __forceinline__ void load(const AggParams& p, int row, int f0, bool active) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { xd[i] = 0.f; xin[i] = 0.f; }
        logd = p.log_deg ? p.log_deg[row] : 0.f;
        if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
        if (active && (p.need & NEED_XIN)) ldx<VEC, ODD>(xin, p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
    }
};

// group of the wave: first row, number of rows, row pointers (lane l: indptr[row0 + min(l, nrows)])
struct ShortGroup {
    int row0, nrows, ipv;
    __device__ __forceinline__ bool init(const AggParams& p) {
        const int wpb = blockDim.x >> 6;
        const int64_t n_groups = (p.n_nodes + kShortRows - 1) / kShortRows;
        const int64_t n_blocks = (n_groups + wpb - 1) / wpb;
        const int64_t lb = xcd_remap(blockIdx.x, n_blocks);
        if (lb < 0) return false;
        const int64_t g = lb * wpb + (threadIdx.x >> 6);
        if (g >= n_groups) return false;
        row0 = uniform_i((int)(g * kShortRows));
        nrows = (int)min((int64_t)kShortRows, p.n_nodes - row0);
        ipv = p.indptr[row0 + min(lane_id(), nrows)];
        return true;
    }
    // explicit group (the fused layer kernel walks its own row ranges)
    __device__ __forceinline__ bool init_at(const AggParams& p, int64_t first_row) {
        if (first_row >= p.n_nodes) return false;
        row0 = uniform_i((int)first_row);
        nrows = (int)min((int64_t)kShortRows, p.n_nodes - row0);
        ipv = p.indptr[row0 + min(lane_id(), nrows)];
        return true;
    }
    __device__ __forceinline__ int ptr(int r) const { return bcast_i(ipv, r); }
};

// kShortRows rows of one wave with every row LEFT in LDS: row r of the group goes to lds_rows + r * row_stride, laid out per
// tower as it would lie in a tower-major output ([tower][aggregator][Ft], tower blocks K floats apart).  The sweep half of the
// fused layer kernel (layer_fwd_fused); one feature tile, no scalers in the row.
template <class C, class O>
__device__ __forceinline__ void short_group_to_lds(const AggParams& p, const ShortGroup& grp, int f0, bool active, float* lds_rows,
                                                   int row_stride) {
    constexpr int VEC = C::VEC, R = kShortRows, J = kShortDeg;
    const int K = p.agg_total * p.Ft;
    int t_of = 0;
    for (int q = 1; q < p.n_towers; ++q) t_of += (f0 >= q * p.Ft) ? 1 : 0;
    const int lane_off = t_of * K + (f0 - t_of * p.Ft);
    const int beg0 = grp.ptr(0);
    int lo[R], deg[R], max_deg = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        lo[r] = grp.ptr(min(r, grp.nrows)) - beg0;
        deg[r] = grp.ptr(min(r + 1, grp.nrows)) - beg0 - lo[r];
        max_deg = max(max_deg, deg[r]);
    }
    if (max_deg > J || (p.x_src && p.m_edge)) {   // a longer row, or two gathered parts per message: row at a time
        for (int r = 0; r < grp.nrows; ++r) fwd_one_row<C, O>(p, grp.row0 + r, f0, active, lds_rows + r * row_stride + lane_off);
        return;
    }
    RowSide<VEC, C::ODD> side[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r == 0 || r < grp.nrows) side[r].load(p, grp.row0 + r, f0, active);
    SlotBatch<C::NCH, C::NW> b;
    b.load(p, beg0, beg0 + lo[R - 1] + deg[R - 1]);
    if (!active) return;
    const MsgSrc<VEC, C::ODD> src(p);
    float t[R][J][VEC];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (j < deg[r]) src.load(t[r][j], bcast_i(b.src, lo[r] + j), beg0 + lo[r] + j, f0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r != 0 && r >= grp.nrows) break;
        Acc<C, false> acc;
        acc.init();
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (j < deg[r]) {
                float mm[VEC], wk[C::NW];
#pragma unroll
                for (int i = 0; i < VEC; ++i) mm[i] = side[r].xd[i] + t[r][j][i];
                b.weights(wk, lo[r] + j);
                acc.add(mm, wk, beg0 + lo[r] + j);
            }
        }
        write_row<C, O>(acc, p, lds_rows + r * row_stride + lane_off, deg[r], side[r].xin, side[r].logd);
    }
}

// MEDGE: the message has a third, per-edge term m_edge[slot] (dense edge features: rows in slot order, so a group's rows are one
// contiguous block): a second tile, issued with the gathers.
template <class C, class O = DynOps, bool AUX = false, bool MEDGE = false>
__global__ __launch_bounds__(256) void agg_fwd_short(const AggParams p) {
    constexpr int VEC = C::VEC, R = kShortRows, J = kShortDeg;
    extern __shared__ float lds_rows[];
    // table mode of the m_edge term (one row per edge TYPE): every workgroup keeps the table in LDS, a slot's term is an LDS read
    const float* tab = nullptr;
    if (p.edge_type) {
        float* dst = lds_rows + p.tab_off;
        for (int i = threadIdx.x; i < p.n_edge_types * p.F; i += blockDim.x) {
            const int k = i / p.F;
            dst[i] = p.m_edge[(int64_t)k * p.ld_edge + (i - k * p.F)];
        }
        __syncthreads();
        tab = dst;
    }
    ShortGroup grp;
    if (!grp.init(p)) return;
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    const int beg0 = grp.ptr(0);
    // in-degree of each row of the group (0 for rows past the end of the graph)
    int lo[R], deg[R], max_deg = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        lo[r] = grp.ptr(min(r, grp.nrows)) - beg0;
        deg[r] = grp.ptr(min(r + 1, grp.nrows)) - beg0 - lo[r];
        max_deg = max(max_deg, deg[r]);
    }
    if (max_deg > J || (!MEDGE && p.x_src && p.m_edge && !tab)) {   // a longer row, or an untiled second part per message: row at a time
        for (int r = 0; r < grp.nrows; ++r) fwd_one_row<C, O>(p, grp.row0 + r, f0, active);
        return;
    }
    // every load of the group is issued before the first store: a load waited for after a store drains the store
    // first (loads and stores share one in-order counter on gfx950), so the rows are finished from registers only
    RowSide<VEC, C::ODD> side[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r == 0 || r < grp.nrows) side[r].load(p, grp.row0 + r, f0, active);
    SlotBatch<C::NCH, C::NW> b;
    b.load(p, beg0, beg0 + lo[R - 1] + deg[R - 1]);
    // Tower-major output: a lane owns one feature pair of ONE tower, so a direct store of an aggregator block is
    // n_towers pieces of Ft floats (56 bytes on ZINC).  With p.stage_out the wave first lays the row out in its LDS
    // slice exactly as it lies in memory per tower ([tower][aggregator][Ft]) and then stores every tower's row --
    // agg_total * Ft contiguous floats (336 bytes) -- with consecutive lanes.  All lanes stay for that second step.
    const bool staged = p.stage_out != 0;
    if (!active && !staged) return;
    const int K = p.agg_total * p.Ft;                                     // floats of one row inside one tower
    // stage_out == 2: ALL rows of the group are laid out first ([tower][row][K]) and each tower's nrows * K contiguous floats
    // (1344 bytes on ZINC) leave as one run of 16-byte lanes -- full 128-byte lines apart from the run's two ends, which the
    // neighbouring waves of the workgroup complete (tools/microbench/rowwrite.hip: 4.2 -> 4.9 TB/s for the same bytes)
    const bool grouped = p.stage_out == 2;
    float* slice = lds_rows + (size_t)(threadIdx.x >> 6) * (grouped ? R : 1) * p.n_towers * K;
    const bool wide = staged && (K & 3) == 0 && (p.tower_stride & 3) == 0 && (p.ld_out & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
    int t_of = 0;                                                        // tower and in-tower feature of this lane
    if (staged) for (int q = 1; q < p.n_towers; ++q) t_of += (f0 >= q * p.Ft) ? 1 : 0;
    float* lds_row = slice + t_of * (grouped ? R : 1) * K + (f0 - t_of * p.Ft);
    const MsgSrc<VEC, C::ODD> src(p);
    // tile [row][j-th slot of the row]: the register index is static, the slot (= lane of the batch) is not --
    // one compare per tile instead of a range check of every slot against every row
    float t[R][J][VEC], t2[MEDGE ? R : 1][MEDGE ? J : 1][VEC];
    unsigned auxv[AUX ? R : 1][VEC] = {};
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (j < deg[r]) {
                    if constexpr (MEDGE) {
                        ldx<VEC, C::ODD>(t[r][j], p.x_src + (int64_t)bcast_i(b.src, lo[r] + j) * p.ld_src, f0, p.Fv);
                        ldv<VEC>(t2[r][j], p.m_edge + (int64_t)(beg0 + lo[r] + j) * p.ld_edge + f0);
                    } else {
                        src.load(t[r][j], bcast_i(b.src, lo[r] + j), beg0 + lo[r] + j, f0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r != 0 && r >= grp.nrows) break;
        float* orow = p.out + (int64_t)(grp.row0 + r) * p.ld_out;
        if (active) {
            Acc<C, AUX> acc;
            acc.init();
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (j < deg[r]) {
                    float mm[VEC], wk[C::NW];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) mm[i] = side[r].xd[i] + t[r][j][i];
                    if constexpr (MEDGE) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) mm[i] += t2[r][j][i];            // (load_msg's order: (x_dst + x_src) + m_edge)
                    }
                    if (tab) {
                        const float* tr = tab + bcast_i(b.et, lo[r] + j) * p.F + f0;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) mm[i] += tr[i];
                    }
                    b.weights(wk, lo[r] + j);
                    acc.add(mm, wk, AUX ? j : beg0 + lo[r] + j);          // (AUX: the tracked "slot" is the position within the row)
                }
            }
            write_row<C, O, AUX>(acc, p, staged ? lds_row + (grouped ? r * K : 0) : orow + lane_col(p, f0), deg[r], side[r].xin, side[r].logd, !staged);
            if constexpr (AUX) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) auxv[r][i] = aux_byte<C, true>(acc, i, side[r].xin[i]);
            }
        }
        if (staged && !grouped) {
            // (same wave: LDS operations complete in program order, no barrier needed)
            if (wide) {
                // 16-byte lanes over ALL towers' rows at once: n_towers * K / 4 pieces (105 on ZINC: two store instructions
                // per row instead of five 8-byte ones)
                const int K4 = K >> 2, total4 = p.n_towers * K4;
                for (int i4 = lane_id(); i4 < total4; i4 += kWave) {
                    const int q = i4 / K4, c4 = i4 - q * K4;
                    st_stream(reinterpret_cast<float4*>(orow + (int64_t)q * p.tower_stride + 4 * c4), reinterpret_cast<const float4*>(slice)[i4]);
                }
            } else {
                const int c = lane_id() * VEC;
                if (c < K) {
                    for (int q = 0; q < p.n_towers; ++q) {
                        float v[VEC];
                        ldv<VEC>(v, slice + q * K + c);
                        stv_stream<VEC>(orow + (int64_t)q * p.tower_stride + c, v);
                    }
                }
            }
        }
    }
    if constexpr (AUX) {
        static_assert(R == 4, "the aux table is laid out per group of four rows");
        if (active) store_aux_group<VEC>(p.aux + aux_offset(p, grp.row0, f0), auxv);      // (rows past the graph's end: unread)
    }
    if (grouped) {
        // (same wave: LDS operations complete in program order, no barrier needed)
        const int run = grp.nrows * K;                   // (K is even; a full group's 4 K floats are a whole number of 16-byte pieces)
        float* obase = p.out + (int64_t)grp.row0 * p.ld_out;
        for (int q = 0; q < p.n_towers; ++q) {
            const float* from = slice + q * R * K;
            float* to = obase + (int64_t)q * p.tower_stride;
            if ((run & 3) == 0) {
                for (int i4 = lane_id(); i4 < (run >> 2); i4 += kWave) st_stream(reinterpret_cast<float4*>(to) + i4, reinterpret_cast<const float4*>(from)[i4]);
            } else {
                for (int i2 = lane_id(); i2 < (run >> 1); i2 += kWave) st_stream(reinterpret_cast<float2*>(to) + i2, reinterpret_cast<const float2*>(from)[i2]);
            }
        }
    }
}

__device__ __forceinline__ void slice_bounds(const AggParams& p, int chunk, int& hub, int& row, int& beg, int& end) {
    hub = p.chunk_hub[chunk];
    row = p.hub_rows[hub];
    const int rbeg = p.indptr[row], rend = p.indptr[row + 1];
    beg = rbeg + (chunk - p.hub_chunk_ptr[hub]) * p.hub_chunk;
    end = min(beg + p.hub_chunk, rend);
}

// one wave per hub slice: partial accumulators -> workspace
template <class C, bool TRACK>
__global__ __launch_bounds__(kBlock) void agg_hub_slices(const AggParams p) {
    constexpr int VEC = C::VEC;
    const int64_t chunk64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk64 >= p.n_chunks) return;
    const int chunk = uniform_i((int)chunk64);
    int hub, row, beg, end;
    slice_bounds(p, chunk, hub, row, beg, end);
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<C, TRACK> acc;
    acc.init();
    float xd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xd[i] = 0.f;
    if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
    accumulate_range<C, TRACK>(acc, p, beg, end, f0, active, xd);
    store_partial<C, TRACK>(acc, p, chunk, f0, active);
}

// one wave per hub row: merge its slices in slot order, then the normal epilogue
template <class C>
__global__ __launch_bounds__(kBlock) void agg_fwd_hub_combine(const AggParams p) {
    constexpr int VEC = C::VEC;
    const int64_t hub64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (hub64 >= p.n_hub) return;
    const int hub = uniform_i((int)hub64);
    const int row = p.hub_rows[hub];
    const int deg = p.indptr[row + 1] - p.indptr[row];
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<C, false> acc, part;
    acc.init();
    for (int c = p.hub_chunk_ptr[hub]; c < p.hub_chunk_ptr[hub + 1]; ++c) {
        load_partial<C, false>(part, p, c, f0, active);
        acc.merge(part);
    }
    if (!active) return;
    float xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) xin[i] = 0.f;
    if (p.need & NEED_XIN) ldx<VEC, C::ODD>(xin, p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
    write_row<C>(acc, p, p.out + (int64_t)row * p.ld_out + lane_col(p, f0), deg, xin, p.log_deg ? p.log_deg[row] : 0.f, true);
}

// ---- backward -----------------------------------------------------------------------------------

template <class C>
struct Coef {
    static constexpr int VEC = C::VEC, NW = C::NW;
    float c0[VEC], cv[VEC], gmax[VEC], gmin[VEC];
    int amax[VEC], amin[VEC];
    float cs[NW][VEC];
    float ca[C::AV ? NW : 1][C::AV ? VEC : 1];
};

// per-row coefficient vectors from the upstream gradient and the (recomputed) accumulators;
// also returns d x_in for this row.  grow already includes the lane's column part.
// `load_g(a, s, g)` delivers the upstream gradient block of (aggregator a, scaler s): from memory, or from a
// register tile the caller loaded earlier (agg_bwd_rows, static lists).
template <class C, class O = DynOps, class LoadG>
__device__ __forceinline__ void make_coef_from(Coef<C>& k, float (&gxin)[C::VEC], const Acc<C, true>& acc, const AggParams& p,
                                               LoadG&& load_g, int deg, const float (&xin)[C::VEC], float logd) {
    constexpr int VEC = C::VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        k.c0[i] = 0.f; k.cv[i] = 0.f; k.gmax[i] = 0.f; k.gmin[i] = 0.f; gxin[i] = 0.f;
        k.amax[i] = -1; k.amin[i] = -1;
    }
    if constexpr (C::STATS) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { k.amax[i] = acc.amax[i]; k.amin[i] = acc.amin[i]; }
    }
#pragma unroll
    for (int c = 0; c < C::NW; ++c) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            k.cs[c][i] = 0.f;
            if constexpr (C::AV) k.ca[c][i] = 0.f;
        }
    }
    const float d = (float)deg;
    RowStats<VEC> st;
    row_stats<C, true>(st, acc, d, p);
    float fac[DGN_MAX_SCALERS];
#pragma unroll
    for (int s = 0; s < DGN_MAX_SCALERS; ++s) fac[s] = s < O::n_scalers(p) ? scaler_factor(O::scaler(p, s), logd, p.avg_log) : 0.f;
    // one aggregator's share of the coefficients, given its upstream gradient g (scalers already summed in)
    auto apply = [&](int a, const float (&g)[VEC]) {
        const int op = O::op(p, a);
        const int c = O::ch(p, a);
        if (op == DGN_AGG_X_IN) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) gxin[i] += g[i];
            return;
        }
        if (op < DGN_AGG_DIR_AV) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if (op == DGN_AGG_MEAN) k.c0[i] += g[i] / d;
                else if (op == DGN_AGG_SUM) k.c0[i] += g[i];
                else if (op == DGN_AGG_MAX) k.gmax[i] += g[i];
                else if (op == DGN_AGG_MIN) k.gmin[i] += g[i];
                else if (st.rawvar[i] > 0.f) {       // VAR / STD; relu'(x) = [x > 0]
                    float gg = g[i];
                    if (op == DGN_AGG_STD) gg = gg / (2.f * st.sd[i]);
                    const float t = gg * 2.f / d;
                    k.cv[i] += t;
                    k.c0[i] -= t * st.mean[i];
                }
            }
            return;
        }
        float wsv[VEC], wav[VEC], swv, dcs[VEC], dca[VEC];
        pick_channel<C, true>(wsv, wav, swv, acc, c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            dcs[i] = 0.f; dca[i] = 0.f;
            if (op == DGN_AGG_DIR_AV) {
                dca[i] = g[i];
            } else if (op == DGN_AGG_DIR_WSUM) {
                dcs[i] = g[i];
            } else if (op == DGN_AGG_DIR_DX_NO_ABS) {
                dcs[i] = g[i];
                gxin[i] -= swv * g[i];
            } else {
                const float r = dx_residual(wsv[i], swv, xin[i]);
                const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);  // d|r|/dr with sign(0) = 0
                dcs[i] = sg * g[i];
                gxin[i] -= sg * swv * g[i];
            }
        }
#pragma unroll
        for (int cc = 0; cc < C::NCH; ++cc) {
            if (cc == c) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    k.cs[cc][i] += dcs[i];
                    if constexpr (C::AV) k.ca[cc][i] += dca[i];
                }
            }
        }
    };
    // The upstream-gradient loads of several aggregators are issued together (tiles of 4 aggregators when
    // there is one scaler, 2 otherwise): a one-at-a-time loop would be a chain of n_agg dependent latencies.
    if (O::n_scalers(p) == 1) {
        constexpr int AT = 4;
        for_each_agg<O, AT>(p, [&](int a0) {
            float t[AT][VEC];
#pragma unroll
            for (int j = 0; j < AT; ++j) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) t[j][i] = 0.f;
                if (a0 + j < O::n_agg(p)) load_g(a0 + j, 0, t[j]);
            }
#pragma unroll
            for (int j = 0; j < AT; ++j) {
                if (a0 + j < O::n_agg(p)) {
                    if (O::scaler(p, 0) != DGN_SCALE_IDENTITY) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) t[j][i] *= fac[0];
                    }
                    apply(a0 + j, t[j]);
                }
            }
        });
    } else {
        constexpr int AT = 2;
        for_each_agg<O, AT>(p, [&](int a0) {
            float t[AT][DGN_MAX_SCALERS][VEC];
#pragma unroll
            for (int j = 0; j < AT; ++j) {
#pragma unroll
                for (int s = 0; s < DGN_MAX_SCALERS; ++s) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) t[j][s][i] = 0.f;
                    if (a0 + j < O::n_agg(p) && s < O::n_scalers(p)) load_g(a0 + j, s, t[j][s]);
                }
            }
#pragma unroll
            for (int j = 0; j < AT; ++j) {
                if (a0 + j < O::n_agg(p)) {
                    float g[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
#pragma unroll
                    for (int s = 0; s < DGN_MAX_SCALERS; ++s) {
                        if (s < O::n_scalers(p)) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i)
                                g[i] += O::scaler(p, s) == DGN_SCALE_IDENTITY ? t[j][s][i] : t[j][s][i] * fac[s];
                        }
                    }
                    apply(a0 + j, g);
                }
            }
        });
    }
}

template <class C, class O = DynOps>
__device__ __forceinline__ void make_coef(Coef<C>& k, float (&gxin)[C::VEC], const Acc<C, true>& acc, const AggParams& p,
                                          const float* grow, int deg, const float (&xin)[C::VEC], float logd) {
    make_coef_from<C, O>(k, gxin, acc, p, [&](int a, int s, float (&g)[C::VEC]) { ldv<C::VEC>(g, grow + sa_col(p, s, a)); },
                         deg, xin, logd);
}

// emit dm_j for the cnt slots of one loaded slot batch (my_tpos: the lane's csc position, two-phase scatter);
// adds them to the row-sum rsum.  Active lanes only.
// block backward: one per-edge gradient row added to its source's accumulator row in LDS.  The rows belong to ONE wave (a workgroup
// of agg_bwd_block is a single wave that owns whole graphs) and a lane owns its features: a plain read-add-write, in program order.
// (ds_add_f32 from several waves was measured first: ~2 cycles per LANE on this part -- 75 us of a 250 us backward on ZINC-12k.)
template <int VEC>
__device__ __forceinline__ void blk_add(const AggParams& p, int node, int f0, const float (&v)[VEC]) {
    float* at = p.blk_lds + (node - p.blk_lo) * p.F + f0;
    float cur[VEC];
    ldv<VEC>(cur, at);
#pragma unroll
    for (int i = 0; i < VEC; ++i) cur[i] += v[i];
    stv<VEC>(at, cur);
}

template <class C, bool NEED_M, bool BLK = false>
__device__ __forceinline__ void emit_batch(const Coef<C>& k, float (&rsum)[C::VEC], const AggParams& p,
                                           const SlotBatch<C::NCH, C::NW>& b, int my_tpos, int base, int cnt, int f0,
                                           const float (&xd)[C::VEC]) {
    constexpr int VEC = C::VEC, U = DGN_UNROLL;
    const MsgSrc<VEC, C::ODD> src(p);
    for (int k0 = 0; k0 < cnt; k0 += U) {
        // var/std need the message again: the group's gathers are issued together, BEFORE the group's stores (a
        // load waited for after a store drains the store first: loads and stores share one in-order counter)
        float t[NEED_M ? U : 1][VEC], t2[NEED_M ? U : 1][VEC];
        if constexpr (NEED_M) {
            if (!src.both) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u == 0 || k0 + u < cnt) src.load(t[u], bcast_i(b.src, k0 + u), base + k0 + u, f0);
            } else {                                   // both gathered parts of the group's messages: x_src rows, m_edge rows
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u == 0 || k0 + u < cnt) {
                        ldx<VEC, C::ODD>(t[u], p.x_src + (int64_t)bcast_i(b.src, k0 + u) * p.ld_src, f0, p.Fv);
                        ldv<VEC>(t2[u], p.m_edge + (int64_t)bcast_i(b.et, k0 + u) * p.ld_edge + f0);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k0 + u;
            if (u != 0 && kk >= cnt) break;
            const int s = bcast_i(b.src, kk);
            const int tp = bcast_i(my_tpos, kk);
            float wk[C::NW];
            b.weights(wk, kk);
            const int pos = base + kk;
            float gm[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) gm[i] = k.c0[i];
            if constexpr (NEED_M) {
                float m[VEC];
                if (src.both) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) m[i] = (xd[i] + t[u][i]) + t2[u][i];
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) m[i] = xd[i] + t[u][i];
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) gm[i] = fmaf(k.cv[i], m[i], gm[i]);
            }
#pragma unroll
            for (int c = 0; c < C::NCH; ++c) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) gm[i] = fmaf(wk[c], k.cs[c][i], gm[i]);
                if constexpr (C::AV) {
                    const float a = fabsf(wk[c]);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gm[i] = fmaf(a, k.ca[c][i], gm[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if constexpr (C::STATS) {
                    if (k.amax[i] == pos) gm[i] += k.gmax[i];
                    if (k.amin[i] == pos) gm[i] += k.gmin[i];
                }
                rsum[i] += gm[i];
            }
            if constexpr (BLK) {
                blk_add<VEC>(p, s, f0, gm);
            } else if (p.g_src) {
                if (p.stage) {
                    // atomic-free path: park the row at its csc position; seg_sum_rows adds each source's rows
                    stv<VEC>(p.stage + (int64_t)tp * p.F + f0, gm);
                } else {
                    float* dst = p.g_src + (int64_t)s * p.ldg_src + f0;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, gm[i]);
                }
            }
            if (p.g_edge) stv<VEC>(p.g_edge + (int64_t)pos * p.ldg_edge + f0, gm);
        }
    }
}

// emit dm_j for slots [beg, end) of a row; returns the row-sum of dm_j in rsum
template <class C, bool NEED_M, bool BLK = false>
__device__ __forceinline__ void emit_range(const Coef<C>& k, float (&rsum)[C::VEC], const AggParams& p, int beg, int end,
                                           int f0, bool active, const float (&xd)[C::VEC]) {
    for (int base = beg; base < end; base += kWave) {
        SlotBatch<C::NCH, C::NW> b;
        b.load(p, base, end);
        const int my_tpos = (!BLK && p.stage && base + lane_id() < end) ? p.csc_pos[base + lane_id()] : 0;
        if (active) emit_batch<C, NEED_M, BLK>(k, rsum, p, b, my_tpos, base, min(kWave, end - base), f0, xd);   // (lanes beyond F only help loading the slot batch)
    }
}

template <class C, bool BLK = false>
__device__ __forceinline__ void emit_dispatch(const Coef<C>& k, float (&rsum)[C::VEC], const AggParams& p, int beg, int end,
                                              int f0, bool active, const float (&xd)[C::VEC]) {
    if constexpr (C::STATS) {
        if (p.need & NEED_M_EMIT) {
            emit_range<C, true, BLK>(k, rsum, p, beg, end, f0, active, xd);
            return;
        }
    }
    emit_range<C, false, BLK>(k, rsum, p, beg, end, f0, active, xd);
}

template <class C, bool BLK = false>
__device__ __forceinline__ void emit_batch_dispatch(const Coef<C>& k, float (&rsum)[C::VEC], const AggParams& p,
                                                    const SlotBatch<C::NCH, C::NW>& b, int my_tpos, int base, int cnt, int f0,
                                                    const float (&xd)[C::VEC]) {
    if constexpr (C::STATS) {
        if (p.need & NEED_M_EMIT) {
            emit_batch<C, true, BLK>(k, rsum, p, b, my_tpos, base, cnt, f0, xd);
            return;
        }
    }
    emit_batch<C, false, BLK>(k, rsum, p, b, my_tpos, base, cnt, f0, xd);
}

// per-row gradients d x_dst (= row sum of dm_j) and d x_in.  `plain`: the caller owns the row (row kernel in
// fresh mode) and stores; otherwise hardware atomics into initialised buffers (accumulate mode, hub slices).
template <int VEC, bool BLK = false>
__device__ __forceinline__ void add_row_grads(const AggParams& p, int row, int f0, const float (&rsum)[VEC],
                                              const float (&gxin)[VEC], bool with_xin, bool plain = false) {
    if constexpr (BLK) {
        // the workgroup owns the row: d x_dst stored; d x_in stored -- or, where it aliases d x_src (simple layers: x_src = x_in = h),
        // added to the row's accumulator in LDS, which the workgroup writes once
        if (p.g_dst) stv<VEC>(p.g_dst + (int64_t)row * p.ldg_dst + f0, rsum);
        if (with_xin && p.g_in) {
            if (p.g_in == p.g_src) blk_add<VEC>(p, row, f0, gxin);
            else stv<VEC>(p.g_in + (int64_t)row * p.ldg_in + f0, gxin);
        }
        return;
    }
    if (p.g_dst) {
        float* dst = p.g_dst + (int64_t)row * p.ldg_dst + f0;
        if (plain) {
            stv<VEC>(dst, rsum);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, rsum[i]);
        }
    }
    if (with_xin && p.g_in && (plain || (p.need & NEED_XIN))) {
        float* dst = p.g_in + (int64_t)row * p.ldg_in + f0;
        if (plain) {
            stv<VEC>(dst, gxin);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, gxin[i]);
        }
    }
}

// Backward of one row whose slots fit ONE slot batch (every row of a molecule / kNN graph): one batch load serves
// recompute and emit, and every load of the row -- slot batch, csc positions, side inputs, upstream gradient (static
// lists), gathers -- is issued before the first store; separate passes would re-load the batch after the recompute and
// fetch the gradient after the gather wait (two more dependent round trips per row).
template <class C, class O, bool BLK = false>
__device__ __forceinline__ void bwd_row_one_batch(const AggParams& p, int row, int beg, int end, int f0, bool active) {
    constexpr int VEC = C::VEC;
    const int deg = end - beg;
    float xd[VEC], xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xd[i] = 0.f; xin[i] = 0.f; }
    const float logd = p.log_deg ? p.log_deg[row] : 0.f;
    const float* grow = p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0);
    Acc<C, true> acc;
    acc.init();
    Coef<C> k;
    float gxin[VEC], rsum[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) rsum[i] = 0.f;
    SlotBatch<C::NCH, C::NW> b;
    b.load(p, beg, end);
    const int my_tpos = (!BLK && p.stage && beg + lane_id() < end) ? p.csc_pos[beg + lane_id()] : 0;
    bool recomp = (p.need & NEED_RECOMP) != 0;
    // aux_rows: the forward left the dx signs (the only thing such a list recomputes for): no gathers, no x_dst / x_in rows
    bool signs = false;
    unsigned auxw = 0;
    if constexpr (!C::STATS && C::NCH >= 1 && C::NCH <= 2) {
        signs = recomp && p.aux_rows != 0;
        if (signs && active) auxw = load_aux_row<VEC>(p.aux + (int64_t)row * p.F + f0);
    }
    if (signs) recomp = false;
    if constexpr (C::NCH > 0) {
        if (!recomp && !signs) {       // only sum_j w_jc is needed (d x_in of dx-no-abs): the batch's weights alone, no gathers
#pragma unroll
            for (int c = 0; c < C::NCH; ++c) acc.sw[c] = wave_sum(b.w[c]);       // (all lanes still here)
        }
    }
    if (!active) return;
    if (!signs) {
        if (p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
        if (p.need & NEED_XIN) ldx<VEC, C::ODD>(xin, p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
    }
    constexpr bool PRE = O::kStatic && O::NA <= 8;
    float gpre[PRE ? O::NA : 1][VEC];
    const bool pre = PRE && O::n_scalers(p) == 1;
    if constexpr (PRE) {
        if (pre) {
#pragma unroll
            for (int a = 0; a < O::NA; ++a) ldv<VEC>(gpre[a], grow + sa_col(p, 0, a));
        }
    }
    const MsgSrc<VEC, C::ODD> src(p);
    if (recomp) accumulate_batch<C, true>(acc, p, src, b, beg, deg, f0, xd);
    if constexpr (!C::STATS && C::NCH >= 1 && C::NCH <= 2) {
        if (signs) {                   // sum_j w_jc in acc.add()'s order, the residual's sign from the table
            for (int k_ = 0; k_ < deg; ++k_) {
#pragma unroll
                for (int c = 0; c < C::NCH; ++c) acc.sw[c] += bcast_f(b.w[c], k_);
            }
            acc_signs_from_aux<C, true>(acc, auxw);
        }
    }
    if constexpr (PRE) {
        if (pre) {
            make_coef_from<C, O>(k, gxin, acc, p, [&](int a, int, float (&g)[VEC]) {
#pragma unroll
                for (int aa = 0; aa < O::NA; ++aa) {
                    if (aa == a) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) g[i] = gpre[aa][i];
                    }
                }
            }, deg, xin, logd);
        } else {
            make_coef<C, O>(k, gxin, acc, p, grow, deg, xin, logd);
        }
    } else {
        make_coef<C, O>(k, gxin, acc, p, grow, deg, xin, logd);
    }
    emit_batch_dispatch<C, BLK>(k, rsum, p, b, my_tpos, beg, deg, f0, xd);
    add_row_grads<VEC, BLK>(p, row, f0, rsum, gxin, true, p.fresh);
}

// the backward of ONE destination row, any in-degree (what a wave of agg_bwd_rows does)
template <class C, class O, bool BLK = false>
__device__ __forceinline__ void bwd_any_row(const AggParams& p, int row, int f0, bool active) {
    constexpr int VEC = C::VEC;
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    const int deg = end - beg;
    if (deg > p.hub_threshold || deg == 0) {
        // hub row: the slice kernels own it (they add with atomics: in fresh mode this kernel zeroes the row first).
        // Row without messages: no gradient -- except through the x_in pass-through block.
        if (!active) return;
        float gx[VEC], zero[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { gx[i] = 0.f; zero[i] = 0.f; }
        if (deg == 0 && (p.need & NEED_XPASS) && p.g_in) {
            const float* grow = p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0);
            for (int a = 0; a < O::n_agg(p); ++a) {
                if (O::op(p, a) == DGN_AGG_X_IN) {
                    float g[VEC];
                    ldv<VEC>(g, grow + sa_col(p, 0, a));
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gx[i] += g[i];
                }
            }
        }
        if (BLK || p.fresh) {
            add_row_grads<VEC, BLK>(p, row, f0, zero, gx, true, true);
        } else if (deg == 0 && (p.need & NEED_XPASS) && p.g_in) {
            float* dst = p.g_in + (int64_t)row * p.ldg_in + f0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(dst + i, gx[i]);
        }
        return;
    }
    float xd[VEC], xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xd[i] = 0.f; xin[i] = 0.f; }
    const float logd = p.log_deg ? p.log_deg[row] : 0.f;
    const float* grow = p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0);
    Acc<C, true> acc;
    acc.init();
    Coef<C> k;
    float gxin[VEC], rsum[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) rsum[i] = 0.f;
    if (deg <= kWave) {
        bwd_row_one_batch<C, O, BLK>(p, row, beg, end, f0, active);
        return;
    }
    bool signs = false;
    if constexpr (!C::STATS && C::NCH >= 1 && C::NCH <= 2) signs = (p.need & NEED_RECOMP) && p.aux_rows != 0;
    if (!signs) {
        if (active && p.x_dst) ldv<VEC>(xd, p.x_dst + (int64_t)row * p.ld_dst + f0);
        if (active && (p.need & NEED_XIN)) ldx<VEC, C::ODD>(xin, p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
    }
    if (signs) {
        if constexpr (!C::STATS && C::NCH >= 1 && C::NCH <= 2) {
            // (aux_rows, see bwd_row_one_batch) the weights of the row's slots summed in slot order, batch after batch
            const unsigned auxw = active ? load_aux_row<VEC>(p.aux + (int64_t)row * p.F + f0) : 0u;
            for (int base = beg; base < end; base += kWave) {
                const int e = base + lane_id(), cnt = min(kWave, end - base);
                float wl[C::NW];
#pragma unroll
                for (int c = 0; c < C::NW; ++c) wl[c] = 0.f;
#pragma unroll
                for (int c = 0; c < C::NCH; ++c) wl[c] = e < end ? p.w[(int64_t)c * p.ld_w + e] : 0.f;
                for (int k_ = 0; k_ < cnt; ++k_) {
#pragma unroll
                    for (int c = 0; c < C::NCH; ++c) acc.sw[c] += bcast_f(wl[c], k_);
                }
            }
            acc_signs_from_aux<C, true>(acc, auxw);
        }
    } else if (p.need & NEED_RECOMP) {
        accumulate_range<C, true>(acc, p, beg, end, f0, active, xd);
    } else if constexpr (C::NCH > 0) {
        // only sum_j w_jc is needed (d x_in of dx-no-abs): weights alone, no gathers
        const int lane = lane_id();
        float part[C::NW];
#pragma unroll
        for (int c = 0; c < C::NW; ++c) part[c] = 0.f;
        for (int e = beg + lane; e < end; e += kWave) {
#pragma unroll
            for (int c = 0; c < C::NCH; ++c) part[c] += p.w[(int64_t)c * p.ld_w + e];
        }
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) acc.sw[c] = wave_sum(part[c]);
    }
    if (active) make_coef<C, O>(k, gxin, acc, p, grow, deg, xin, logd);
    emit_dispatch<C, BLK>(k, rsum, p, beg, end, f0, active, xd);
    if (active) add_row_grads<VEC, BLK>(p, row, f0, rsum, gxin, true, p.fresh);
}

template <class C, class O = DynOps>
__global__ __launch_bounds__(256) void agg_bwd_rows(const AggParams p) {
    const int wpb = blockDim.x >> 6;   // 1 (long rows: a finished row frees its slot at once) or 4 (short rows: dispatch-rate bound)
    constexpr int VEC = C::VEC;
    const int64_t n_blocks = (p.n_nodes + wpb - 1) / wpb;
    const int64_t lb = xcd_remap(blockIdx.x, n_blocks);
    if (lb < 0) return;
    const int64_t row64 = lb * wpb + (threadIdx.x >> 6);
    if (row64 >= p.n_nodes) return;
    const int row = uniform_i((int)row64);
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    bwd_any_row<C, O>(p, row, f0, f0 < p.F);
}

// RB consecutive rows per wave when every one of them has 1..kShortDeg in-edges (molecule batches): ONE row-pointer load and ONE
// slot batch serve the group, and the upstream-gradient blocks, side rows and source gathers of ALL its rows are in flight together
// before the first row is worked on.  agg_bwd_rows keeps one row per wave in flight: three dependent round trips (row pointer ->
// slot batch -> gathers) per row at four waves per SIMD, which is what bounds it on such graphs.  Same
// arithmetic in the same order as bwd_row_one_batch: bit-identical gradients.  Groups with a longer / empty / hub row, partial
// groups, and launches outside the fast case (dynamic lists, several scalers, accumulate mode, an m_edge term, the atomic scatter)
// take the per-row routine.
// EDGE: the message has an edge-TYPE table term (DgnMsg.edge_type): a second tile of (L1-resident) table rows.
// The backward of RB consecutive rows [row0, row0 + nrows) by one wave (what a wave of agg_bwd_short does; also the sweep half of the
// fused backward kernel, layer_bwd_fused, whose p.g_out points at upstream-gradient rows it has just formed in LDS).
// GLDS: p.g_out points into LDS -- the upstream-gradient blocks are then read where they are used instead of being requested up front
// (48 registers on the ZINC list: with them the fused kernel's 16-wave workgroups would spill).
// AUX: p.aux holds, for every (row, feature) of such a group, what the recompute would find (aux_byte): no message is formed again --
// no source gathers, no x_dst / x_in rows -- and the coefficient code runs on accumulators that carry just those facts.
// The row-indexed operands of a group of RB rows (what bwd_short_group requests up front), as a value the block kernel can request
// one group AHEAD: upstream-gradient blocks, log-degrees, the aux words or the x_dst / x_in rows.
template <class C, int RB, int NG, bool AUX>
struct GroupRows {
    float xd[RB][C::VEC], xin[RB][C::VEC], logd[RB], gpre[RB][NG][C::VEC];
    unsigned auxw[AUX ? 4 : 1];
};
template <class O>
constexpr int n_gout_blocks() { if constexpr (O::kStatic) return O::NA * O::kNS; else return 1; }
// branch-free, clamped into the arrays (rows beyond the graph and lanes beyond the row re-read the last valid one: prefetches must not
// sit under a branch or a select -- the compiler would wait for them on the spot)
template <class C, class O, int RB, bool AUX>
__device__ __forceinline__ void load_group_rows(GroupRows<C, RB, n_gout_blocks<O>(), AUX>& R, const AggParams& p, int row0, int f0) {
    constexpr int VEC = C::VEC;
    const int last = (int)p.n_nodes - 1;
    row0 = min(row0, last & ~(RB - 1));
    const int f0c = min(f0, p.F - VEC);
    if constexpr (AUX) load_aux_group<VEC>(R.auxw, p.aux + aux_offset(p, row0, f0c));
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = min(row0 + r, last);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { R.xd[r][i] = 0.f; R.xin[r][i] = 0.f; }
        R.logd[r] = p.log_deg ? p.log_deg[row] : 0.f;
        if constexpr (!AUX) {
            if (p.x_dst) ldv<VEC>(R.xd[r], p.x_dst + (int64_t)row * p.ld_dst + f0c);
            if (p.need & NEED_XIN) ldx<VEC, C::ODD>(R.xin[r], p.x_in + (int64_t)row * p.ld_in, f0c, p.Fv);
        }
        const float* grow = p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0c);
#pragma unroll
        for (int sc = 0; sc < O::kNS; ++sc)
#pragma unroll
            for (int a = 0; a < O::NA; ++a) ldv<VEC>(R.gpre[r][sc * O::NA + a], grow + sa_col(p, sc, a));
    }
}

// BLK (agg_bwd_block): d x_src goes to the wave's accumulator rows in LDS; the group's row pointers (ipv_in: lane r holds
// indptr[row0 + r], r <= RB), its slot batch (b_in) and its row operands (rows_in: a GroupRows) were requested an iteration
// earlier by the caller.
template <class C, class O, int RB, bool EDGE = false, bool GLDS = false, bool AUX = false, bool BLK = false, bool PREROWS = false>
__device__ __forceinline__ void bwd_short_group(const AggParams& p, int row0, int nrows, int f0, bool active, int ipv_in = 0,
                                                const SlotBatch<C::NCH, C::NW>* b_in = nullptr, const void* rows_in = nullptr) {
    constexpr int VEC = C::VEC, J = kShortDeg;
    constexpr int NG = []() { if constexpr (O::kStatic) return O::NA * O::kNS; else return 99; }();     // upstream-gradient blocks per row
    constexpr bool PRE = O::kStatic && NG <= 8;
    bool fast = PRE && nrows == RB && (BLK || (p.stage && p.fresh)) && p.g_src && p.x_src && (EDGE ? p.m_edge != nullptr : (!p.m_edge && !p.g_edge)) &&
                !(p.need & NEED_M_EMIT);      // (no static list carries std / var: their emit term stays with the per-row routine)
    const bool recomp = (p.need & NEED_RECOMP) != 0;     // (otherwise only sum_j w_jc is needed: no gathers at all)
    int lo[RB], deg[RB], beg0 = 0;
    if (fast) {
        int ipv;
        if constexpr (BLK) ipv = ipv_in;
        else ipv = p.indptr[row0 + min(lane_id(), RB)];
        beg0 = bcast_i(ipv, 0);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            lo[r] = bcast_i(ipv, r) - beg0;
            deg[r] = bcast_i(ipv, r + 1) - beg0 - lo[r];
            fast = fast && deg[r] >= 1 && deg[r] <= J;
        }
    }
    if (!fast) {
        for (int r = 0; r < nrows; ++r) {
            if (BLK && (row0 + r < p.blk_lo || row0 + r >= p.blk_hi)) continue;      // (a group across a block boundary: the neighbour's rows)
            bwd_any_row<C, O, BLK>(p, row0 + r, f0, active);
        }
        return;
    }
    if constexpr (PRE) {
        const int total = lo[RB - 1] + deg[RB - 1];
        SlotBatch<C::NCH, C::NW> b;
        if constexpr (BLK) b = *b_in;
        else b.load(p, beg0, beg0 + total);
        const int my_tpos = (!BLK && lane_id() < total) ? p.csc_pos[beg0 + lane_id()] : 0;
        if (!active) return;
        // every load of the group, issued before anything is consumed
        float xd[RB][VEC], xin[RB][VEC], logd[RB], gpre[GLDS ? 1 : RB][GLDS ? 1 : NG][VEC], t[AUX ? 1 : RB][AUX ? 1 : J][VEC],
              t2[(EDGE && !AUX) ? RB : 1][(EDGE && !AUX) ? J : 1][VEC];
        unsigned auxw[AUX ? 4 : 1];
        if constexpr (PREROWS) {
            static_assert(BLK && !GLDS && !EDGE, "the block kernel takes the plain message forms");
            const auto& R = *static_cast<const GroupRows<C, RB, n_gout_blocks<O>(), AUX>*>(rows_in);
#pragma unroll
            for (int q = 0; q < (AUX ? 4 : 1); ++q) auxw[q] = R.auxw[q];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                logd[r] = R.logd[r];
#pragma unroll
                for (int i = 0; i < VEC; ++i) { xd[r][i] = R.xd[r][i]; xin[r][i] = R.xin[r][i]; }
#pragma unroll
                for (int q = 0; q < NG; ++q)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gpre[r][q][i] = R.gpre[r][q][i];
            }
        }
        if constexpr (AUX && !PREROWS) {
            static_assert(RB == 4, "the aux table is laid out per group of four rows");
            load_aux_group<VEC>(auxw, p.aux + aux_offset(p, row0, f0));
        }
#pragma unroll
        for (int r = 0; r < (PREROWS ? 0 : RB); ++r) {
            const int row = row0 + r;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { xd[r][i] = 0.f; xin[r][i] = 0.f; }
            logd[r] = p.log_deg ? p.log_deg[row] : 0.f;
            if constexpr (!AUX) {
                if (p.x_dst) ldv<VEC>(xd[r], p.x_dst + (int64_t)row * p.ld_dst + f0);
                if (p.need & NEED_XIN) ldx<VEC, C::ODD>(xin[r], p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
            }
            if constexpr (!GLDS) {
                const float* grow = p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0);
#pragma unroll
                for (int sc = 0; sc < O::kNS; ++sc)
#pragma unroll
                    for (int a = 0; a < O::NA; ++a) ldv<VEC>(gpre[r][sc * O::NA + a], grow + sa_col(p, sc, a));
            }
        }
        if constexpr (!AUX) {
            if (recomp) {
#pragma unroll
                for (int r = 0; r < RB; ++r) {
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        if (j < deg[r]) {
                            ldx<VEC, C::ODD>(t[r][j], p.x_src + (int64_t)bcast_i(b.src, lo[r] + j) * p.ld_src, f0, p.Fv);
                            if constexpr (EDGE) ldv<VEC>(t2[r][j], p.m_edge + (int64_t)bcast_i(b.et, lo[r] + j) * p.ld_edge + f0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (BLK && (row0 + r < p.blk_lo || row0 + r >= p.blk_hi)) continue;      // (a group across a block boundary: the neighbour's rows)
            Acc<C, true> acc;
            acc.init();
            if constexpr (AUX) {
                // what acc.add() over the row's messages would leave in the fields the coefficient code reads: sum_j w_jc in slot
                // order, the slots of the first maximum / minimum, and a stand-in of the dx residual with its sign (x_in reads as 0)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (j < deg[r]) {
#pragma unroll
                        for (int c = 0; c < C::NCH; ++c) acc.sw[c] += bcast_f(b.w[c], lo[r] + j);
                    }
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const unsigned ab = (auxw[r] >> (8 * i)) & 0xffu;
                    if constexpr (C::STATS) {
                        acc.amax[i] = beg0 + lo[r] + (int)(ab & 3u);
                        acc.amin[i] = beg0 + lo[r] + (int)((ab >> 2) & 3u);
                    }
#pragma unroll
                    for (int c = 0; c < (C::NCH < 2 ? C::NCH : 2); ++c) {
                        const unsigned code = (ab >> (4 + 2 * c)) & 3u;
                        acc.ws[c][i] = code == 1u ? 1.f : (code == 2u ? -1.f : 0.f);
                    }
                }
            } else if (recomp) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (j < deg[r]) {
                        float mm[VEC], wk[C::NW];
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            mm[i] = xd[r][i] + t[r][j][i];
                            if constexpr (EDGE) mm[i] += t2[r][j][i];
                        }
                        b.weights(wk, lo[r] + j);
                        acc.add(mm, wk, beg0 + lo[r] + j);
                    }
                }
            } else {
                // sum_j w_jc in the order of wave_sum over a batch that holds this row alone: (w0 + w2) + (w1 + w3)
                static_assert(J == 4, "the summation tree below is wave_sum's for four slots");
#pragma unroll
                for (int c = 0; c < C::NCH; ++c) {
                    float wj[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) wj[j] = j < deg[r] ? bcast_f(b.w[c], lo[r] + j) : 0.f;
                    acc.sw[c] = (wj[0] + wj[2]) + (wj[1] + wj[3]);
                }
            }
            Coef<C> k;
            float gxin[VEC], rsum[VEC];
            make_coef_from<C, O>(k, gxin, acc, p, [&](int a, int sc, float (&g)[VEC]) {
                if constexpr (GLDS) {
                    ldv<VEC>(g, p.g_out + (int64_t)(row0 + r) * p.ld_gout + lane_col(p, f0) + sa_col(p, sc, a));
                } else {
#pragma unroll
                    for (int q = 0; q < NG; ++q) {
                        if (q == sc * O::NA + a) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) g[i] = gpre[r][q][i];
                        }
                    }
                }
            }, deg[r], xin[r], logd[r]);
#pragma unroll
            for (int i = 0; i < VEC; ++i) rsum[i] = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (j < deg[r]) {
                    const int l = lo[r] + j, pos = beg0 + l;
                    float wk[C::NW], gm[VEC];
                    b.weights(wk, l);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gm[i] = k.c0[i];
#pragma unroll
                    for (int c = 0; c < C::NCH; ++c) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gm[i] = fmaf(wk[c], k.cs[c][i], gm[i]);
                        if constexpr (C::AV) {
                            const float a = fabsf(wk[c]);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) gm[i] = fmaf(a, k.ca[c][i], gm[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        if constexpr (C::STATS) {
                            if (k.amax[i] == pos) gm[i] += k.gmax[i];
                            if (k.amin[i] == pos) gm[i] += k.gmin[i];
                        }
                        rsum[i] += gm[i];
                    }
                    if constexpr (BLK) blk_add<VEC>(p, bcast_i(b.src, l), f0, gm);
                    else stv<VEC>(p.stage + (int64_t)bcast_i(my_tpos, l) * p.F + f0, gm);
                    if constexpr (EDGE) {          // dense edge term: its gradient is dm_j itself, rows in slot order
                        if (p.g_edge) stv<VEC>(p.g_edge + (int64_t)pos * p.ldg_edge + f0, gm);
                    }
                }
            }
            add_row_grads<VEC, BLK>(p, row0 + r, f0, rsum, gxin, true, true);
        }
    }
}

template <class C, class O, int RB, bool EDGE = false, bool AUX = false>
__global__ __launch_bounds__(256) void agg_bwd_short(const AggParams p) {
    const int wpb = blockDim.x >> 6;
    const int64_t n_groups = (p.n_nodes + RB - 1) / RB;
    const int64_t n_blocks = (n_groups + wpb - 1) / wpb;
    const int64_t lb = xcd_remap(blockIdx.x, n_blocks);
    if (lb < 0) return;
    const int64_t g64 = lb * wpb + (threadIdx.x >> 6);
    if (g64 >= n_groups) return;
    const int row0 = uniform_i((int)(g64 * RB));
    const int nrows = (int)min((int64_t)RB, p.n_nodes - row0);
    const int f0 = (blockIdx.y * kWave + lane_id()) * C::VEC;
    bwd_short_group<C, O, RB, EDGE, false, AUX>(p, row0, nrows, f0, f0 < p.F);
}

// hub backward, phase 2: merge slice partials, build the row's coefficient vectors, park them
template <class C>
__global__ __launch_bounds__(kBlock) void agg_bwd_hub_coef(const AggParams p) {
    constexpr int VEC = C::VEC;
    const int64_t hub64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (hub64 >= p.n_hub) return;
    const int hub = uniform_i((int)hub64);
    const int row = p.hub_rows[hub];
    const int deg = p.indptr[row + 1] - p.indptr[row];
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    Acc<C, true> acc, part;
    acc.init();
    for (int c = p.hub_chunk_ptr[hub]; c < p.hub_chunk_ptr[hub + 1]; ++c) {
        load_partial<C, true>(part, p, c, f0, active);
        acc.merge(part);
    }
    if (!active) return;
    Coef<C> k;
    float gxin[VEC], zero[VEC], xin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { zero[i] = 0.f; xin[i] = 0.f; }
    if (p.need & NEED_XIN) ldx<VEC, C::ODD>(xin, p.x_in + (int64_t)row * p.ld_in, f0, p.Fv);
    make_coef<C>(k, gxin, acc, p, p.g_out + (int64_t)row * p.ld_gout + lane_col(p, f0), deg, xin,
                 p.log_deg ? p.log_deg[row] : 0.f);
    float* base = p.coef + (int64_t)hub * p.n_coef * p.F;
    stv<VEC>(base + (int64_t)COEF_C0 * p.F + f0, k.c0);
    stv<VEC>(base + (int64_t)COEF_CV * p.F + f0, k.cv);
    stv<VEC>(base + (int64_t)COEF_GMAX * p.F + f0, k.gmax);
    stv<VEC>(base + (int64_t)COEF_GMIN * p.F + f0, k.gmin);
    stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)COEF_AMAX * p.F + f0), k.amax);
    stvi<VEC>(reinterpret_cast<int*>(base + (int64_t)COEF_AMIN * p.F + f0), k.amin);
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) {
        stv<VEC>(base + (int64_t)(COEF_W0 + 2 * c) * p.F + f0, k.cs[c]);
        if constexpr (C::AV) stv<VEC>(base + (int64_t)(COEF_W0 + 2 * c + 1) * p.F + f0, k.ca[c]);
    }
    add_row_grads<VEC>(p, row, f0, zero, gxin, true);  // d x_in only (rsum = 0)
}

// hub backward, phase 3: one wave per slice emits with the parked coefficients
template <class C>
__global__ __launch_bounds__(kBlock) void agg_bwd_hub_emit(const AggParams p) {
    constexpr int VEC = C::VEC;
    const int64_t chunk64 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk64 >= p.n_chunks) return;
    const int chunk = uniform_i((int)chunk64);
    int hub, row, beg, end;
    slice_bounds(p, chunk, hub, row, beg, end);
    const int f0 = (blockIdx.y * kWave + lane_id()) * VEC;
    const bool active = f0 < p.F;
    float xd[VEC], rsum[VEC], gxin[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xd[i] = 0.f; rsum[i] = 0.f; gxin[i] = 0.f; }
    Coef<C> k;
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
        for (int c = 0; c < C::NCH; ++c) {
            ldv<VEC>(k.cs[c], base + (int64_t)(COEF_W0 + 2 * c) * p.F + f0);
            if constexpr (C::AV) ldv<VEC>(k.ca[c], base + (int64_t)(COEF_W0 + 2 * c + 1) * p.F + f0);
        }
    }
    emit_dispatch<C>(k, rsum, p, beg, end, f0, active, xd);
    if (active) add_row_grads<VEC>(p, row, f0, rsum, gxin, false);
}

// second phase of the atomic-free backward: g_src[u] (+)= sum of the staged rows of source u (contiguous in csc
// order; p.seg_add selects += over =).  Flat mapping: one thread per (node, VEC-chunk), so short out-neighbourhoods do not cost a wave each.
template <int VEC>
__global__ __launch_bounds__(256) void seg_sum_rows(const AggParams p) {
    constexpr int PER = VEC == 1 ? 4 : (VEC == 2 ? 2 : 1);    // 4 floats per thread whatever the vector width
    const int nchunk = (p.F + VEC * PER - 1) / (VEC * PER);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_src * nchunk) return;
    const int u = (int)(t / nchunk);
    const int f0 = (int)(t - (int64_t)u * nchunk) * VEC * PER;
    const int* ptr = p.csc_ptr;
    const int beg = ptr[u], end = ptr[u + 1];
    const bool add = p.seg_add;
    if (beg == end && add) return;
    float acc[PER][VEC];
#pragma unroll
    for (int q = 0; q < PER; ++q)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[q][i] = 0.f;
    // four staged rows are requested before the first add (row index clamped, adds predicated and in csc order):
    // a load consumed right away would make every out-edge of the node a separate memory round trip
    constexpr int KU = 4;
    for (int k0 = beg; k0 < end; k0 += KU) {
        float r[KU][PER][VEC];
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const int kk = min(k0 + j, end - 1);
            const float* row = p.stage + (int64_t)kk * p.F + f0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[j][q][i] = 0.f;
                if (f0 + q * VEC < p.F) ldv<VEC>(r[j][q], row + q * VEC);
            }
        }
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            if (k0 + j < end) {
#pragma unroll
                for (int q = 0; q < PER; ++q)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[q][i] += r[j][q][i];
            }
        }
    }
    float* dst = p.g_src + (int64_t)u * p.ldg_src + f0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (f0 + q * VEC < p.F) {
            if (add) {
                float cur[VEC];
                ldv<VEC>(cur, dst + q * VEC);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[q][i] += cur[i];
            }
            stv<VEC>(dst + q * VEC, acc[q]);
        }
    }
}

// ---- launchers (one translation unit per VEC: dgn_agg_v{1,2,4}.hip) ----------------------------

// Workgroup shape of the row kernels.  The dispatcher starts ~4.6 workgroups per ns whatever their size
// (measured: tools/microbench), so one-wave workgroups cap a sweep at ~4.6 rows/ns: fine for power-law graphs
// (tens of KB per row, and a finished long row frees its slot immediately: +9 % on C5), but molecule-like
// batches (2-3 edges per row) need 4 rows per workgroup.
inline int row_waves_per_block(const AggParams& p) {
    static const char* env = getenv("DGN_ROW_WPB");
    if (env) { const int v = atoi(env); return v == 1 ? 1 : (v == 2 ? 2 : 4); }
    return (p.n_edges >= 8 * p.n_nodes) ? 1 : 4;
}

constexpr int kBwdShortRows = 4;     // rows per wave of agg_bwd_short (two rows measured 0.293 ms on c2, four 0.26; DGN_BWD_ROWS_PER_WAVE=1 selects agg_bwd_rows)

// kShortRows rows per wave when a group's slots usually fit one gather group (average in-degree <= 3)
inline bool short_rows(const AggParams& p) {
    static const char* env = getenv("DGN_SHORT_ROWS");
    if (env) return atoi(env) != 0;
    return p.n_edges <= 3 * p.n_nodes;
}

template <class C, class O = DynOps>
int launch_forward_cfg(const AggParams& p, unsigned tiles, hipStream_t stream) {
    const int wpb = row_waves_per_block(p);
    if (short_rows(p)) {
        const int64_t n_groups = (p.n_nodes + kShortRows - 1) / kShortRows;
        dim3 grid((unsigned)xcd_grid((n_groups + wpb - 1) / wpb), tiles);
        // LDS-staged stores: tower-major layout, one feature tile, no scalers in the row, a row per tower <= 64 lanes wide
        AggParams q = p;
        const int K = p.agg_total * p.Ft;
        static const bool no_stage = getenv("DGN_NO_STAGE_OUT") != nullptr;
        const bool sa_in_tower = p.agg_offset == 0 && p.n_agg == p.agg_total;     // the launch writes the whole row of every tower
        q.stage_out = (!no_stage && p.n_towers > 1 && tiles == 1 && p.n_scalers == 1 && sa_in_tower && K <= kWave * C::VEC &&
                       (size_t)wpb * p.n_towers * K * sizeof(float) <= 32768) ? 1 : 0;
        // ... and the whole group at once when the rows of a tower plane are contiguous and 16-byte pieces line up: also for ONE
        // tower (simple / complex layers: the group's four rows are one run of 4 A F floats)
        static const bool no_group = getenv("DGN_NO_STAGE_GROUP") != nullptr;
        const size_t group_lds = (size_t)wpb * kShortRows * p.n_towers * K * sizeof(float);
        if (!no_stage && !no_group && tiles == 1 && p.n_scalers == 1 && sa_in_tower && p.ld_out == K && (K & 1) == 0 &&
            (p.n_towers == 1 || ((p.tower_stride & 3) == 0 && (K & 3) == 0)) && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && group_lds <= 32768)
            q.stage_out = 2;
        size_t lds = q.stage_out == 2 ? group_lds : q.stage_out ? (size_t)wpb * p.n_towers * K * sizeof(float) : 0;
        if (p.edge_type) {
            q.tab_off = (int32_t)(lds / sizeof(float));
            lds += (size_t)p.n_edge_types * p.F * sizeof(float);
        }
        bool launched = false;
        const bool medge = p.x_src && p.m_edge && !p.edge_type;        // dense per-slot rows next to the gathered part
        if constexpr (O::kStatic && C::NCH <= 2) {
            if (q.aux) {
                if (medge) hipLaunchKernelGGL((agg_fwd_short<C, O, true, true>), grid, dim3(kWave * wpb), lds, stream, q);
                else hipLaunchKernelGGL((agg_fwd_short<C, O, true>), grid, dim3(kWave * wpb), lds, stream, q);
                launched = true;
            }
        }
        if constexpr (O::kStatic) {
            if (!launched && medge) { hipLaunchKernelGGL((agg_fwd_short<C, O, false, true>), grid, dim3(kWave * wpb), lds, stream, q); launched = true; }
        }
        if (!launched) hipLaunchKernelGGL((agg_fwd_short<C, O>), grid, dim3(kWave * wpb), lds, stream, q);
    } else {
        const int64_t n_blocks = (p.n_nodes + wpb - 1) / wpb;
        dim3 grid((unsigned)xcd_grid(n_blocks), tiles);
        hipLaunchKernelGGL((agg_fwd_rows<C, O>), grid, dim3(kWave * wpb), 0, stream, p);
    }
    if (p.n_hub > 0) {
        dim3 gs((unsigned)((p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
        hipLaunchKernelGGL((agg_hub_slices<C, false>), gs, dim3(kBlock), 0, stream, p);
        dim3 gc((unsigned)((p.n_hub + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
        hipLaunchKernelGGL((agg_fwd_hub_combine<C>), gc, dim3(kBlock), 0, stream, p);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

template <class C, class O = DynOps>
int launch_backward_cfg(const AggParams& p, unsigned tiles, hipStream_t stream) {
    const int wpb = row_waves_per_block(p);
    const int rb = option(OPT_BWD_ROWS_PER_WAVE) <= 1 ? 1 : kBwdShortRows;
    if (O::kStatic && short_rows(p) && p.stage && p.fresh && rb > 1) {       // molecule-like batches, static lists: rows in groups per wave
        const int64_t n_groups = (p.n_nodes + rb - 1) / rb;
        dim3 grid((unsigned)xcd_grid((n_groups + wpb - 1) / wpb), tiles);
        bool launched = false;
        if constexpr (O::kStatic && C::NCH <= 2) {
            if (p.aux) {           // (the host sets it only for lists and graphs agg_aux_supported() accepts)
                if (p.m_edge) hipLaunchKernelGGL((agg_bwd_short<C, O, kBwdShortRows, true, true>), grid, dim3(kWave * wpb), 0, stream, p);
                else hipLaunchKernelGGL((agg_bwd_short<C, O, kBwdShortRows, false, true>), grid, dim3(kWave * wpb), 0, stream, p);
                launched = true;
            }
        }
        if (!launched) {
            if (p.m_edge) hipLaunchKernelGGL((agg_bwd_short<C, O, kBwdShortRows, true>), grid, dim3(kWave * wpb), 0, stream, p);
            else hipLaunchKernelGGL((agg_bwd_short<C, O, kBwdShortRows, false>), grid, dim3(kWave * wpb), 0, stream, p);
        }
    } else {
        const int64_t n_blocks = (p.n_nodes + wpb - 1) / wpb;
        dim3 grid((unsigned)xcd_grid(n_blocks), tiles);
        hipLaunchKernelGGL((agg_bwd_rows<C, O>), grid, dim3(kWave * wpb), 0, stream, p);
    }
    if (p.n_hub > 0) {
        dim3 gs((unsigned)((p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
        dim3 gc((unsigned)((p.n_hub + kWavesPerBlock - 1) / kWavesPerBlock), tiles);
        hipLaunchKernelGGL((agg_hub_slices<C, true>), gs, dim3(kBlock), 0, stream, p);
        hipLaunchKernelGGL((agg_bwd_hub_coef<C>), gc, dim3(kBlock), 0, stream, p);
        hipLaunchKernelGGL((agg_bwd_hub_emit<C>), gs, dim3(kBlock), 0, stream, p);
    }
    if (p.stage && p.g_src) {
        constexpr int per = C::VEC == 1 ? 4 : (C::VEC == 2 ? 2 : 1);
        const int64_t n_threads = p.n_src * ((p.F + C::VEC * per - 1) / (C::VEC * per));
        hipLaunchKernelGGL((seg_sum_rows<C::VEC>), dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, stream, p);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

// the baked-in lists that have kernels for rows of an odd width (Cfg::ODD): the simple layers of the reference's configs at odd hidden
// sizes -- mean dir1-dx-no-abs (ZINC simple, hidden 75) and mean dir1-dx dir2-dx (CIFAR10 / MNIST, hidden 65)
constexpr bool odd_width_list(int na, uint64_t ops, uint64_t chs) { return (na == 2 && ops == 0x90ull && chs == 0x0ull) || (na == 3 && ops == 0x880ull && chs == 0x40ull); }

// runtime (n_ch, stats, av) -> compile-time configuration
template <int VEC, bool BWD>
int launch_vec(const AggParams& p, unsigned tiles, hipStream_t stream) {
    const bool stats = (p.need & (NEED_SQ | NEED_MAX | NEED_MIN)) != 0;
    const bool av = p.any_av;
    // hot aggregator lists of the reference's configs: kernels with the list baked in (dgn_agg_hot.hpp)
    static const bool no_hot = getenv("DGN_NO_HOT") != nullptr;
#define DGN_HOT(NA, OPS, CHS, NS, SCS, N, S, A)                                                                  \
    if (!no_hot && p.n_agg == NA && p.op_pack == OPS && p.ch_pack == CHS && p.n_scalers == NS && p.scaler_pack == SCS && \
        p.agg_total == NA && p.agg_offset == 0 && p.n_ch == N) {                                                 \
        using O = StaticOps<NA, OPS, CHS, NS, SCS>;                                                              \
        if (p.Fv != p.F) {      /* rows of an odd width (DgnMsg.f_valid): the two lists that meet them, 8-byte lanes */ \
            if constexpr (VEC == 2 && odd_width_list(NA, OPS, CHS)) {                                            \
                if constexpr (BWD) return launch_backward_cfg<Cfg<VEC, N, S, A, true>, O>(p, tiles, stream);     \
                else return launch_forward_cfg<Cfg<VEC, N, S, A, true>, O>(p, tiles, stream);                    \
            }                                                                                                    \
            set_error("f_valid: no kernel for this aggregator list (dgn_agg_f_valid_supported)");               \
            return DGN_ERR_INVALID;                                                                              \
        }                                                                                                        \
        if constexpr (BWD) return launch_backward_cfg<Cfg<VEC, N, S, A>, O>(p, tiles, stream);                   \
        else return launch_forward_cfg<Cfg<VEC, N, S, A>, O>(p, tiles, stream);                                  \
    }
#include "dgn_agg_hot.hpp"
#undef DGN_HOT
    if (p.Fv != p.F) { set_error("f_valid: no kernel for this aggregator list (dgn_agg_f_valid_supported)"); return DGN_ERR_INVALID; }
#define DGN_GO(N, S, A)                                                                              \
    if (p.n_ch == N && stats == S && av == A) {                                                      \
        if constexpr (BWD) return launch_backward_cfg<Cfg<VEC, N, S, A>>(p, tiles, stream);          \
        else return launch_forward_cfg<Cfg<VEC, N, S, A>>(p, tiles, stream);                         \
    }
    DGN_GO(0, false, false) DGN_GO(0, true, false)
    DGN_GO(1, false, false) DGN_GO(1, true, false) DGN_GO(1, false, true) DGN_GO(1, true, true)
    DGN_GO(2, false, false) DGN_GO(2, true, false) DGN_GO(2, false, true) DGN_GO(2, true, true)
    DGN_GO(3, false, false) DGN_GO(3, true, false) DGN_GO(3, false, true) DGN_GO(3, true, true)
    DGN_GO(4, false, false) DGN_GO(4, true, false) DGN_GO(4, false, true) DGN_GO(4, true, true)
#undef DGN_GO
    set_error("no kernel for vec=%d n_ch=%d stats=%d av=%d", VEC, p.n_ch, (int)stats, (int)av);
    return DGN_ERR_INVALID;
}

// is the launch's (list, scalers, channels) one of the baked-in hot lists (launch_vec would pick a StaticOps kernel)?
inline bool is_hot_list(const AggParams& p) {
    static const bool no_hot = getenv("DGN_NO_HOT") != nullptr;
    if (no_hot) return false;
#define DGN_HOT(NA, OPS, CHS, NS, SCS, N, S, A)                                                                  \
    if (p.n_agg == NA && p.op_pack == OPS && p.ch_pack == CHS && p.n_scalers == NS && p.scaler_pack == SCS && \
        p.agg_total == NA && p.agg_offset == 0 && p.n_ch == N) return true;
#include "dgn_agg_hot.hpp"
#undef DGN_HOT
    return false;
}

// defined in dgn_agg_v1.hip / _v2.hip / _v4.hip
int launch_agg_v1(const AggParams& p, unsigned tiles, hipStream_t stream, bool backward);
int launch_agg_v2(const AggParams& p, unsigned tiles, hipStream_t stream, bool backward);
int launch_agg_v4(const AggParams& p, unsigned tiles, hipStream_t stream, bool backward);
// defined in dgn_agg_blk_v2.hip (dgn_agg_block.hpp): DGN_OK, an error, or 1 = no block kernel for this launch.  8-byte lanes only: with
// 16-byte lanes (F > 128) a wave's 13 KB of LDS hold fewer rows than one molecule.
int launch_agg_block_v2(const AggParams& p, int gap, hipStream_t stream);
int launch_agg_graph_v2(const AggParams& p, hipStream_t stream);

}  // namespace dgn
