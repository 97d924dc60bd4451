// Degree-class posttrans: the post-aggregation Linear of a DGN layer WITH its degree scalers, at a third of the folded product's flops.
//
// Reference: realworld_benchmark/nets/dgn_layer.py:116-119 / :187-190 -- posttrans(cat(h, agg x scalers)) -- with the scalers of
// nets/scalers.py:7-18, each of which multiplies a node's whole aggregate row by a factor that depends on the node's IN-DEGREE only
// (1, log(d + 1) / avg, avg / log(d + 1)).  Rounds 1-2 folded the scalers behind the Linear: z = agg W_f^T with S f_out output columns,
// y = sum_s scale_s(node) z_s.  All nodes of one in-degree d share their factors, so for them
//     y = agg (sum_s scale_s(d) W_f[s])^T = agg W_d^T
// is ONE product with f_out columns: S (= 3) times fewer MFMA flops in the forward, the input gradient and the weight gradient, and no
// [N, S f_out] intermediate.  Molecules have in-degrees 1..4 (6 classes at most), k-NN graphs one class.
//
// Rows are visited in a VIRTUAL ROW SPACE: nodes stably sorted by in-degree class (0..31), every class segment padded to a multiple of
// 64 rows ("units"); vperm[v] = node of virtual row v (-1: padding), unit_class[u] = class of unit u (-1: empty).  The aggregate rows
// stay where the sweep wrote them -- the kernels gather / scatter whole rows through vperm.
//   dc_fold        W_c[c] = sum_s scale[c][s] W_f[s]  (and its transpose), for the classes present
//   dc_gemm<NQ>    C[node] = row_scale[node] (bias + A[node] W_c^T): the 256 x 16 NQ register-tile kernel of dgn_gemm_kernels.hpp
//                  (both operands through LDS per 16-k chunk) over runs of up to four units of one class
//   dc_wgrad<NTN, KT>  per class G_c = sum_{node in c} g[node]^T x[node]: contiguous ranges of 16-row strips per workgroup, the
//                  accumulators flushed to a partial block at every class change (run id = workgroup + class: unique, see below)
//   dc_wgrad_finalize  g_wf[s n + o][k] = sum_runs scale[class(run)][s] part[run][o][k], fixed order (bitwise reproducible)
#pragma once
#include "dgn_common.hpp"
#include "dgn_load4.hpp"

#include <type_traits>

namespace dgn {
namespace dc {

using gemm::f4;
using gemm::f4u;
using gemm::load4_raw;
using gemm::load4_window;
using gemm::Raw4;

constexpr int kClasses = DGN_DC_CLASSES;     // in-degrees 0 .. 31
constexpr int kUnit = DGN_DC_UNIT;           // rows per unit of the virtual row space
constexpr int kTileM = 256, kTKS = 20;

// The posttrans weight in the reference's layout (DgnDcLayout) seen as the scaler-major folded matrix wf[(s n + o)][kk]:
// kk = a f_pad + f; aggregator blocks a < n_agg at W[o][h_off + (s n_agg + a) f_in + f], the h block (a == n_agg, complex layer) at W[o][f]
// in the identity scaler's rows only, padded feature columns (f >= f_in) zero.  Returns the element's offset in W, or -1.
__device__ __forceinline__ int64_t dc_ref_offset(const DgnDcLayout& L, int s, int o, int kk) {
    const int a = kk / L.f_pad, f = kk - a * L.f_pad;
    if (f >= L.f_in) return -1;
    if (a < L.n_agg) return (int64_t)o * L.ld + L.h_off + (s * L.n_agg + a) * L.f_in + f;
    return s == L.id_slot ? (int64_t)o * L.ld + f : -1;
}

// ---- class weights ---------------------------------------------------------------------------------------------------------------
// wc[c][o][kk] = sum_s scale[c][s] wf[(s n + o)][kk],  wct[c][kk][o] the same transposed; classes without rows are skipped
static __global__ __launch_bounds__(256) void dc_fold(int S, int n, int k, int towers, const int32_t* __restrict__ present, const float* __restrict__ scale,
                                                      const float* __restrict__ wf, float* __restrict__ wc, float* __restrict__ wct, const DgnDcLayout lay,
                                                      int has_layout) {
    // blockIdx.y = class * towers + tower; tower t: wf + t S n k, class c / tower t: wc + (c towers + t) n k
    const int c = blockIdx.y / towers, t = blockIdx.y - c * towers;
    if (present[c] <= 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * k) return;
    const int o = i / k, kk = i - o * k;
    const float* w = wf + (int64_t)t * S * n * k;
    float v = 0.f;
    for (int s = 0; s < S; ++s) {
        float ws;
        if (has_layout) {
            const int64_t off = dc_ref_offset(lay, s, o, kk);
            ws = off >= 0 ? wf[off] : 0.f;
        } else {
            ws = w[(int64_t)(s * n + o) * k + kk];
        }
        v += scale[c * S + s] * ws;
    }
    const int64_t base = ((int64_t)c * towers + t) * n * k;
    wc[base + i] = v;
    wct[base + (int64_t)kk * n + o] = v;
}

// ---- forward / input gradient ----------------------------------------------------------------------------------------------------
struct DcGemmParams {
    int64_t n_units;
    const int32_t* vperm;            // [64 n_units]
    const int32_t* unit_class;       // [n_units]
    int k, n;
    const float* A; int64_t lda;     // [N, k] node rows
    const float* W; int64_t ldw;     // class c: W + c * class_stride, [n, k] (nn.Linear layout)
    int64_t class_stride;
    const float* bias;               // [n] or NULL
    const float* row_scale;          // [N] or NULL
    float* C; int64_t ldc;           // [N, n] node rows
    int n_slice;                     // 16 NQ
    int64_t units_per_block;
    int stream_out;                  // nontemporal result stores
    int col_tiles;                   // > 1: 1-D grid of 8-aligned row ranges x column tiles (XCD-aware dealing), else blockIdx.y = 0
    int64_t a_tower, w_tower, c_tower, bias_tower;      // blockIdx.z = tower: element offsets of its A rows / weights / C columns / bias
    // Round 6 (one tower): BatchNorm's training statistics of C ride in the epilogue -- a wave folds its rows' values (and their fp32 squares) over
    // the sixteen lanes of a column group in fp64 and adds them to fp64 cells of its own in LDS; the workgroup leaves
    // bn_part[(q * bn_F + column) * bn_G + slot], q = 0 (sum) / 1 (sum of squares), slot = its row range (dc_gemm) or unit (dc_gemm_small): what
    // bn_stats would compute in a pass of its own over C, in the layout bn_finalize reads.  NULL: nothing.
    double* bn_part; int bn_F, bn_G;
};

// sum over the sixteen lanes of a DPP row (every lane gets it): quad butterflies, then the two mirrors
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false)); };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1, 0, 3, 2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2, 3, 0, 1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

// One tile of 64 RT virtual rows (units u0 .. u0 + RT, all of one class) x 16 NQ columns.  Wave w owns rows 16 RT w .. of the tile.
// TM: rows the A staging buffers were sized for (dc_gemm: 256; dc_gemm_small: 64, RT = 1 only).
template <int NQ, int RT, bool DEEP = (NQ <= 6), int TM = kTileM>      // (NQ = 7: no registers left for the second set)
__device__ __forceinline__ void dc_tile(const DcGemmParams& p, float* As, float* Bs, int64_t u0, const float* __restrict__ W, int n0, double* St = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int KB = (p.k + 15) >> 4;
    const int lr = tid >> 2, c4 = (tid & 3) * 4;
    constexpr int NBJ = (NQ * 16 + 63) / 64;
    constexpr int kABuf = TM * kTKS, kBBuf = NQ * 16 * kTKS;
    static_assert(64 * RT <= TM, "tile taller than its staging buffer");
    int node[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) node[j] = p.vperm[(u0 + j) * kUnit + lr];
    Raw4 ra0[RT], rb0[NBJ];
    auto fetch = [&](Raw4 (&ra)[RT], Raw4 (&rb)[NBJ], int kc) {
        const int k0 = 16 * kc + c4;
        // (padding rows and the weight rows past n are loaded from a clamped address and NOT zeroed: each row of A and of W only feeds
        //  its own output row / column, which is never stored -- so `sh` stays 0 in every lane except in a ragged last k chunk)
#pragma unroll
        for (int j = 0; j < RT; ++j) ra[j] = load4_raw(p.A + (int64_t)max(node[j], 0) * p.lda, k0, p.k);
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int c = n0 + lr + 64 * j;
            rb[j] = load4_raw(W + (int64_t)min(c, p.n - 1) * p.ldw, k0, p.k);
        }
    };
    auto commit = [&](const Raw4 (&ra)[RT], const Raw4 (&rb)[NBJ], int buf) {
#pragma unroll
        for (int j = 0; j < RT; ++j) *reinterpret_cast<f4*>(As + buf * kABuf + (lr + 64 * j) * kTKS + c4) = load4_window(ra[j]);
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
            if (lr + 64 * j < NQ * 16) *reinterpret_cast<f4*>(Bs + buf * kBBuf + (lr + 64 * j) * kTKS + c4) = load4_window(rb[j]);
    };
    f4 acc[RT][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = n0 + 16 * q + 4 * g + r;
            const float b = (p.bias && col < p.n) ? p.bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][q][r] = b;
        }
    }
    auto mma = [&](int buf) {
        const float* al = As + buf * kABuf + (16 * RT * wave + i16) * kTKS + 4 * g;
        const float* bl = Bs + buf * kBBuf + i16 * kTKS + 4 * g;
        f4 xa[RT], wb[NQ];
#pragma unroll
        for (int t = 0; t < RT; ++t) xa[t] = *reinterpret_cast<const f4*>(al + 16 * t * kTKS);
#pragma unroll
        for (int q = 0; q < NQ; ++q) wb[q] = *reinterpret_cast<const f4*>(bl + 16 * q * kTKS);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[q][s], xa[rt][s], acc[rt][q], 0, 0, 0);
    };
    fetch(ra0, rb0, 0);
    if constexpr (DEEP) {
        // two chunks in flight: chunk kc + 2 is requested before the MFMAs of chunk kc, chunk kc + 1 (requested one round earlier) is
        // committed behind them -- a load has two rounds to arrive (ZINC-280k: forward 0.0965 -> 0.0903 ms, input gradient 0.1019 ->
        // 0.0944; neutral on HIV's 52 k rows).  Two register sets that swap roles by NAME -- rotating one into the other would read a
        // load still in flight -- and fetches without a branch around them: the last rounds re-request the last chunk.
        Raw4 ra1[RT], rb1[NBJ];
        fetch(ra1, rb1, min(1, KB - 1));
        commit(ra0, rb0, 0);
        __syncthreads();
        for (int kc = 0; kc < KB; kc += 2) {
            fetch(ra0, rb0, min(kc + 2, KB - 1));
            mma(0);
            if (kc + 1 < KB) commit(ra1, rb1, 1);
            __syncthreads();
            if (kc + 1 >= KB) break;
            fetch(ra1, rb1, min(kc + 3, KB - 1));
            mma(1);
            if (kc + 2 < KB) commit(ra0, rb0, 0);
            __syncthreads();
        }
    } else {
        commit(ra0, rb0, 0);
        __syncthreads();
        for (int kc = 0; kc < KB; ++kc) {
            if (kc + 1 < KB) fetch(ra0, rb0, kc + 1);                      // next chunk in flight during the MFMAs
            mma(kc & 1);
            if (kc + 1 < KB) commit(ra0, rb0, (kc + 1) & 1);
            __syncthreads();
        }
    }
    if constexpr (NQ <= 6) if (p.bn_part) {           // (uniform) column sums of what is stored below: St[(2 wave + q) * 16 NQ + column]
        int nds[RT];
        float rss[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            nds[rt] = p.vperm[u0 * kUnit + 16 * RT * wave + 16 * rt + i16];
            rss[rt] = (p.row_scale && nds[rt] >= 0) ? p.row_scale[nds[rt]] : 1.f;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // (the <= 64 rows a column group of this wave holds: fp32 over the lane's rows and the sixteen lanes -- DPP adds, no LDS traffic --, fp64 from there)
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (nds[rt] >= 0) {                   // (padding rows of a unit hold copies of row 0: not part of the batch)
                    f4 v = acc[rt][q];
                    if (p.row_scale) v = v * rss[rt];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
            if (i16 == 0) {
                double* c1 = St + (2 * wave) * (16 * NQ) + 16 * q + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) { lds_add_f64(c1 + r, (double)s1[r]); lds_add_f64(c1 + 16 * NQ + r, (double)s2[r]); }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int nd = p.vperm[u0 * kUnit + 16 * RT * wave + 16 * rt + i16];
        if (nd >= 0) {
            const float rs = p.row_scale ? p.row_scale[nd] : 1.f;
            float* crow = p.C + (int64_t)nd * p.ldc;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = n0 + 16 * q + 4 * g;
                f4 v = acc[rt][q];
                if (p.row_scale) v = v * rs;
                if (col + 3 < p.n) {
                    if (p.stream_out) __builtin_nontemporal_store(v, reinterpret_cast<f4u*>(crow + col));
                    else *reinterpret_cast<f4u*>(crow + col) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (col + r < p.n) crow[col + r] = v[r];
                }
            }
        }
    }
}

// A workgroup owns the units [blockIdx.x * units_per_block, + units_per_block) and walks them in tiles of up to four units of one class,
// the range cut into the fewest tiles of nearly equal height (dgn_gemm_kernels.hpp: tile_gemm).
// DcGemmParams.bn_part: the workgroup's four waves' cells, column by column, to slot `slot`
template <int NQ>
__device__ __forceinline__ void dc_stats_flush(const DcGemmParams& p, const double* St, int n0, int64_t slot) {
    __syncthreads();
    const int tid = threadIdx.x;
    if (NQ <= 6 && tid < 2 * 16 * NQ) {
        const int q = tid / (16 * NQ), c = tid - q * (16 * NQ);
        if (n0 + c < p.n) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += St[(2 * w + q) * (16 * NQ) + c];
            p.bn_part[((int64_t)q * p.bn_F + n0 + c) * p.bn_G + slot] = a;
        }
    }
}

template <int NQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dc_gemm(const DcGemmParams p_in) {
    __shared__ float As[2 * kTileM * kTKS];
    __shared__ float Bs[2 * NQ * 16 * kTKS];
    __shared__ double St[NQ <= 6 ? 8 * 16 * NQ : 1];      // (the statistics ride in the tiles of up to 96 columns only: wider ones have no registers to spare)
    DcGemmParams p = p_in;
    if constexpr (NQ > 6) p.bn_part = nullptr;
    if (p.bn_part) {
        for (int i = threadIdx.x; i < 8 * 16 * NQ; i += 256) St[i] = 0.0;
        __syncthreads();
    }
    // several column tiles: a 1-D grid dealt so that the column tiles of ONE row range are consecutive workgroups of ONE XCD (the
    // dispatcher places workgroup b on XCD b % 8): they run together and share the range's A rows in that XCD's L2
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.col_tiles > 1) {
        const int xcd = bx % kXcds, j = bx / kXcds;
        by = j % p.col_tiles;
        bx = (j / p.col_tiles) * kXcds + xcd;
    }
    const int n0 = by * p.n_slice;
    int64_t u = (int64_t)bx * p.units_per_block;
    const int64_t end = min(p.n_units, u + p.units_per_block);
    p.A += blockIdx.z * p.a_tower; p.W += blockIdx.z * p.w_tower; p.C += blockIdx.z * p.c_tower;
    if (p.bias) p.bias += blockIdx.z * p.bias_tower;
    while (u < end) {
        const int c = uniform_i(p.unit_class[u]);
        if (c < 0) { ++u; continue; }
        const int left = (int)min((int64_t)64, end - u), tiles = (left + 3) >> 2;
        const int h = min(4, (left + tiles - 1) / tiles);
        int run = 1;
        while (run < h && uniform_i(p.unit_class[u + run]) == c) ++run;
        const float* W = p.W + (int64_t)c * p.class_stride;
        switch (run) {
            case 4: dc_tile<NQ, 4>(p, As, Bs, u, W, n0, St); break;
            case 3: dc_tile<NQ, 3>(p, As, Bs, u, W, n0, St); break;
            case 2: dc_tile<NQ, 2>(p, As, Bs, u, W, n0, St); break;
            default: dc_tile<NQ, 1>(p, As, Bs, u, W, n0, St); break;
        }
        u += run;
    }
    if (p.bn_part) dc_stats_flush<NQ>(p, St, n0, bx);
}

// Few units (HIV batch 2048: 816 units of 64 rows): dc_gemm's two 256-row workgroups per CU leave the chip a single, latency-bound round of
// ~1.6 workgroups per CU, each walking its k chunks with two loads in flight.  Here a workgroup owns ONE unit (64 rows x 16 NQ columns,
// 23 KB of LDS, <= 128 registers): four workgroups per CU, every unit resident at once, four times the loads in flight per CU.
template <int NQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void dc_gemm_small(const DcGemmParams p_in) {
    __shared__ float As[2 * 64 * kTKS];
    __shared__ float Bs[2 * NQ * 16 * kTKS];
    __shared__ double St[NQ <= 6 ? 8 * 16 * NQ : 1];
    DcGemmParams p = p_in;
    if constexpr (NQ > 6) p.bn_part = nullptr;
    const int64_t u = blockIdx.x;
    const int n0 = blockIdx.y * p.n_slice;
    if (p.bn_part) {
        for (int i = threadIdx.x; i < 8 * 16 * NQ; i += 256) St[i] = 0.0;
        __syncthreads();
    }
    p.A += blockIdx.z * p.a_tower; p.W += blockIdx.z * p.w_tower; p.C += blockIdx.z * p.c_tower;
    if (p.bias) p.bias += blockIdx.z * p.bias_tower;
    const int c = uniform_i(p.unit_class[u]);
    if (c >= 0) dc_tile<NQ, 1, true, 64>(p, As, Bs, u, p.W + (int64_t)c * p.class_stride, n0, St);
    if (p.bn_part) dc_stats_flush<NQ>(p, St, n0, u);      // (an unused unit leaves a slot of zeros)
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
constexpr int kWgWaves = 8;       // 512 threads: one workgroup per CU may use 256 registers per lane (NTN x KT accumulator tiles)

struct DcWgradParams {
    int64_t n_units;
    const int32_t* vperm;
    const int32_t* unit_class;
    int n, k;                        // G [N, n], X [N, k]
    const float* G; int64_t ldg;
    const float* X; int64_t ldx;
    float* part;                     // [k_slices][slots + kClasses][NTN * 16][kpad]
    uint32_t* run_mask;              // [slots] classes flushed by each workgroup (bit c: partial block x + c is valid)
    int k_slice, kpad, slots;        // k_slice = columns per blockIdx.y = 128 KT
    int64_t units_per_block;
};

// 8 waves; wave w owns all NTN n-tiles x the k-tiles w + 8 b (b < KT <= 4) of the workgroup's k slice.  A run (workgroup x, class c) gets the
// partial block x + c: the workgroups' unit ranges are contiguous and ascending and the classes ascend along the virtual rows, so
// (x, c) -> x + c is strictly increasing along the sequence of runs, hence unique and ascending in class for the finalize.
// KT: k tiles per wave the LDS strips are laid out for; KTW (KT or KT - 1): the k tiles THIS wave accumulates (w + 8 b, b < KTW) -- with
// k = 420 (27 tiles) three waves own four tiles and five own three: the wave-uniform count is a template argument (MFMAs under run-time
// guards make the compiler copy the accumulators), every wave runs the same barriers whatever its instantiation.
template <int NTN, int KT, int KTW>
__device__ __forceinline__ void dc_wgrad_run(const DcWgradParams& p, float* lds_dc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.y * p.k_slice, k_here = min(p.k_slice, p.k - k0);
    constexpr int gs = NTN * 16 + 4, xs = KT * 128 + 4;          // LDS row strides
    constexpr int half = 16 * (gs + xs);
    const int gq = (p.n + 3) >> 2, xq = (k_here + 3) >> 2, per_row = gq + xq, total = 16 * per_row;
    constexpr int kMaxItems = (16 * (NTN * 4 + KT * 32) + kWave * kWgWaves - 1) / (kWave * kWgWaves);
    auto fetch = [&](Raw4 (&reg)[kMaxItems], int64_t strip) {
#pragma unroll
        for (int j = 0; j < kMaxItems; ++j) {
            const int it = min(tid + j * kWave * kWgWaves, total - 1);
            const int r = it / per_row, c = it - r * per_row;
            const int nd = p.vperm[strip * 16 + r];
            const int64_t row = max(nd, 0);
            const bool is_g = c < gq;
            reg[j] = load4_raw(is_g ? p.G + row * p.ldg : p.X + row * p.ldx, is_g ? 4 * c : k0 + 4 * (c - gq), is_g ? p.n : p.k, nd >= 0);
        }
    };
    auto commit = [&](const Raw4 (&reg)[kMaxItems], int buf) {
        float* Gb = lds_dc + buf * half;
        float* Xb = Gb + 16 * gs;
#pragma unroll
        for (int j = 0; j < kMaxItems; ++j) {
            const int it = tid + j * kWave * kWgWaves;
            if (it < total) {
                const int r = it / per_row, c = it - r * per_row;
                const f4 v = load4_window(reg[j]);
                if (c < gq) *reinterpret_cast<f4*>(Gb + r * gs + 4 * c) = v;
                else *reinterpret_cast<f4*>(Xb + r * xs + 4 * (c - gq)) = v;
            }
        }
    };
    const int ids = p.slots + kClasses;
    const int64_t s_begin = (int64_t)blockIdx.x * p.units_per_block * (kUnit / 16);
    const int64_t s_end = min(p.n_units, ((int64_t)blockIdx.x + 1) * p.units_per_block) * (kUnit / 16);
    if (s_begin >= s_end) {
        if (tid == 0 && blockIdx.y == 0) p.run_mask[blockIdx.x] = 0u;
        return;
    }
    uint32_t flushed = 0u;
    Raw4 sreg[kMaxItems];
    fetch(sreg, s_begin);
    commit(sreg, 0);
    __syncthreads();
    int buf = 0;
    int64_t strip = s_begin;
    // one pass of the outer loop = one run: the strips of one class (+ empty units behind it).  The accumulators are zeroed and flushed
    // BETWEEN the MFMA loops -- a flush under a condition inside the loop made the compiler keep copies of every accumulator.
    while (strip < s_end) {
        const int cls = uniform_i(p.unit_class[strip / (kUnit / 16)]);
        if (cls < 0) break;                                        // (empty units lie behind the last class only)
        int64_t run_end = strip;
        while (run_end < s_end) {
            const int c = uniform_i(p.unit_class[run_end / (kUnit / 16)]);
            if (c >= 0 && c != cls) break;
            run_end += kUnit / 16;
        }
        run_end = min(run_end, s_end);
        f4 acc[NTN][KTW];
#pragma unroll
        for (int a = 0; a < NTN; ++a)
#pragma unroll
            for (int b = 0; b < KTW; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        for (; strip < run_end; ++strip, buf ^= 1) {
            const bool more = strip + 1 < s_end;
            if (more) fetch(sreg, strip + 1);
            const float* Gb = lds_dc + buf * half;
            const float* Xb = Gb + 16 * gs;
            // D[n][k] += G[m][n] X[m][k], the strip's rows are the reduction index: m = 4 s + g in the s-th instruction
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float gv[NTN], xv[KTW];
#pragma unroll
                for (int a = 0; a < NTN; ++a) gv[a] = Gb[(4 * s + g) * gs + 16 * a + i16];
#pragma unroll
                for (int b = 0; b < KTW; ++b) xv[b] = Xb[(4 * s + g) * xs + 16 * (wave + kWgWaves * b) + i16];
#pragma unroll
                for (int a = 0; a < NTN; ++a)
#pragma unroll
                    for (int b = 0; b < KTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[a], xv[b], acc[a][b], 0, 0, 0);
            }
            if (more) commit(sreg, buf ^ 1);
            __syncthreads();
        }
        float* out = p.part + ((int64_t)blockIdx.y * ids + blockIdx.x + cls) * (NTN * 16) * p.kpad;
#pragma unroll
        for (int b = 0; b < KTW; ++b) {
            const int col = 16 * (wave + kWgWaves * b) + i16;
            if (col < p.kpad) {
#pragma unroll
                for (int a = 0; a < NTN; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[(16 * a + 4 * g + r) * p.kpad + col] = acc[a][b][r];
            }
        }
        flushed |= 1u << cls;
    }
    if (tid == 0 && blockIdx.y == 0) p.run_mask[blockIdx.x] = flushed;
}

template <int NTN, int KT>
__global__ __launch_bounds__(kWave * kWgWaves) void dc_wgrad(const DcWgradParams p) {
    extern __shared__ float lds_dc[];
    if constexpr (KT > 1) {
        const int wave = uniform_i((int)threadIdx.x >> 6);
        const int tiles = (min(p.k_slice, p.k - (int)blockIdx.y * p.k_slice) + 15) >> 4;
        const int mine = wave < tiles ? (tiles - wave + kWgWaves - 1) / kWgWaves : 0;      // tiles w, w + 8, ... below `tiles`
        if (mine < KT) { dc_wgrad_run<NTN, KT, KT - 1>(p, lds_dc); return; }
    }
    dc_wgrad_run<NTN, KT, KT>(p, lds_dc);
}

// g_wf[(s n + o)][kk] = sum over the runs, in (workgroup, class) order, of scale[class][s] * part[run][o][kk]; a block covers 64
// consecutive elements, its sixteen waves take every sixteenth workgroup, LDS joins them in wave order (bitwise reproducible).
// has_layout: the sums go straight into the reference's weight-gradient layout (DgnDcLayout) instead.
template <int S>
static __global__ __launch_bounds__(64 * 16) void dc_wgrad_finalize(int n, int k, int k_slice, int kpad, int npad, int slots, const uint32_t* __restrict__ run_mask,
                                                                    const float* __restrict__ scale, const float* __restrict__ part,
                                                                    float* __restrict__ g_wf, int64_t ldw, const DgnDcLayout lay, int has_layout) {
    __shared__ float red[S][16][64];
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const bool live = e < (int64_t)n * k;
    const int o = live ? (int)(e / k) : 0, kk = live ? (int)(e - (int64_t)o * k) : 0;
    const int sl = kk / k_slice, cc = kk - sl * k_slice;
    const int ids = slots + kClasses;
    const float* src = part + (int64_t)sl * ids * npad * kpad + (int64_t)o * kpad + cc;
    float out[S];
#pragma unroll
    for (int s = 0; s < S; ++s) out[s] = 0.f;
    for (int x = sg; x < slots; x += 16) {
        uint32_t m = run_mask[x];
        while (m) {
            const int c = __builtin_ctz(m);
            m &= m - 1;
            const float v = live ? src[(int64_t)(x + c) * npad * kpad] : 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) out[s] += scale[c * S + s] * v;
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) red[s][sg][lane] = out[s];
    __syncthreads();
    if (live && sg < S) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[sg][w][lane];
        if (has_layout) {
            const int64_t off = dc_ref_offset(lay, sg, o, kk);
            if (off >= 0) g_wf[off] = v;
        } else {
            g_wf[(int64_t)(sg * n + o) * ldw + kk] = v;
        }
    }
}

}  // namespace dc
}  // namespace dgn
