// Wide tall-skinny fp32 GEMMs for the layers whose post-aggregation Linear does not fit dgn_linear_kernels.hpp's "whole weight
// matrix in LDS" scheme: the simple / complex layers' posttrans (reference nets/dgn_layer.py:148,187-190 and :69,116-119 via
// nets/layers.py:101-112): A [M, k] with M = number of nodes (1e4 .. 1e6), k = aggregators x features = 152 .. 420 after scaler
// folding, n = scalers x f_out = 65 .. 225 (and the transposed shapes of the input gradient).  The library GEMM the round-1 layers
// called for these runs at 40-45 TFLOP/s on such shapes and only with a shape-keyed TunableOp selection (a training run with
// varying node counts never hits it); these kernels are shape independent.
//   ts_gemm<NT, WKN>     C[M, n-slice] = A . W^T (+ bias)  (WKN: W given as [k, n]: the input gradient)
//                        512 threads = 8 waves x 16 rows; each wave keeps its 16 x (NT*16) output tile in NT accumulators and walks k
//                        in 16-wide chunks: A straight from memory in the MFMA lane layout (16 bytes per lane, unaligned-safe), the
//                        W chunk staged in LDS by the whole workgroup, double buffered.  Exact fp32: v_mfma_f32_16x16x4_f32.
//   ts_gemm_wgrad<KT>    dW[n, k-slice] = G^T . X over the rows: 16 waves, wave a owns n-tile a and KT k-tiles; 16-row strips of
//                        G and X staged in LDS (double buffered), per-workgroup partials, fixed-order finalize (bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dgn_common.hpp"
#include "dgn_load4.hpp"

namespace dgn {
namespace gemm {

constexpr int kWaves = 8;            // ts_gemm: rows per workgroup pass = 16 * kWaves
#ifndef DGN_GEMM_KC
#define DGN_GEMM_KC 16     // 32 (half the barriers) measured 5-15 % slower: more registers, one workgroup per CU either way
#endif
constexpr int kKC = DGN_GEMM_KC;
#ifndef DGN_GEMM_ABL
#define DGN_GEMM_ABL 0      // timing ablations (tools only, results wrong): 1 no A loads after the first chunk, 2 no W fetch / commit after the first, 3 no barrier in the k loop, 4 = 1 + 2
#endif              // reduction columns per LDS chunk (one barrier per chunk)
constexpr int kKS = kKC + 4;         // LDS row stride of a weight chunk (floats): spreads the rows over the banks
constexpr int kMaxNT = 16;           // n-slice of at most 256 columns per workgroup column (128: measured 5-15 % slower)

struct GemmParams {
    int64_t M;
    int k, n;                        // full reduction width, full output width
    const float* A; int64_t lda;
    const float* W; int64_t ldw;     // WKN == 0: [n, k];  WKN == 1: [k, n]
    const float* bias;               // [n] or NULL
    float* C; int64_t ldc;
    int n_slice;                     // columns per blockIdx.y (a multiple of 16)
    int64_t rows_per_block;          // tile_gemm: rows of one workgroup's range (a multiple of 64)
};

#ifdef DGN_GEMM_WPE
#define DGN_GEMM_ATTR __attribute__((amdgpu_waves_per_eu(DGN_GEMM_WPE, DGN_GEMM_WPE)))
#else
#define DGN_GEMM_ATTR
#endif
template <int NT, int WKN>
__global__ __launch_bounds__(kWave * kWaves) DGN_GEMM_ATTR void ts_gemm(const GemmParams p) {
    extern __shared__ float Wc_dyn[];
    constexpr int kBuf = NT * 16 * kKS;                        // floats per weight-chunk buffer (two of them)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * p.n_slice;                    // first output column of this workgroup column
    const int n_here = min(NT * 16, p.n - n0);
    const int KB = (p.k + kKC - 1) / kKC;
    const int64_t n_blocks = (p.M + 16 * kWaves - 1) / (16 * kWaves);

    // chunk kc of the weights, element (r, c) = W(n0 + r, 16 kc + c), zero outside: FETCHED into registers (global -> VGPR, issued
    // before the chunk's MFMAs) and COMMITTED to Wc[buf] after them -- a load consumed by its LDS store right away would make the
    // wave sit out the L2 latency in front of every chunk's MFMAs
    constexpr int kC4 = kKC / 4;                                                        // float4's per weight row and chunk
    constexpr int kItems = (NT * 16 * kC4 + kWave * kWaves - 1) / (kWave * kWaves);      // float4's per thread and chunk
    auto fetch = [&](Raw4 (&reg)[kItems], int kc) {
        const int k0 = kKC * kc;
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const int it = min(tid + j * kWave * kWaves, NT * 16 * kC4 - 1);      // (surplus threads repeat the last piece; commit skips them)
            if (WKN == 0) {
                const int r = it / kC4, c4 = (it % kC4) * 4;
                reg[j] = load4_raw(p.W + (int64_t)(n0 + min(r, n_here - 1)) * p.ldw, k0 + c4, p.k, r < n_here);
            } else {
                const int c = it / (NT * 4), r4 = (it - c * (NT * 4)) * 4;
                reg[j] = load4_raw(p.W + (int64_t)min(k0 + c, p.k - 1) * p.ldw, n0 + r4, p.n, k0 + c < p.k);      // (columns of the next slice land in rows >= n_here: never stored)
            }
        }
    };
    auto commit = [&](const Raw4 (&reg)[kItems], int buf) {
        float* dst = Wc_dyn + buf * kBuf;
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const int it = tid + j * kWave * kWaves;
            if (it < NT * 16 * kC4) {
                const f4 v = load4_window(reg[j]);
                if (WKN == 0) {
                    const int r = it / kC4, c4 = (it % kC4) * 4;
                    *reinterpret_cast<f4*>(dst + r * kKS + c4) = v;
                } else {
                    const int c = it / (NT * 4), r4 = (it - c * (NT * 4)) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[(r4 + e) * kKS + c] = v[e];
                }
            }
        }
    };
    // this lane's four A values of chunk kc: A[row][16 kc + 4 g .. + 3]
    constexpr int SB = kKC / 16;                                // 16-column sub-blocks (one MFMA k-group each) per chunk
    struct AChunk { f4 v[SB]; };
    struct ARaw { Raw4 r[SB]; };
    auto load_a = [&](const float* arow, int kc) {
        ARaw a;
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) a.r[sb] = load4_raw(arow, kKC * kc + 16 * sb + 4 * g, p.k);
        return a;
    };
    auto window_a = [&](const ARaw& a) {
        AChunk x;
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) x.v[sb] = load4_window(a.r[sb]);
        return x;
    };

    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t row = blk * (16 * kWaves) + wave * 16 + i16;
        const float* arow = p.A + min(row, p.M - 1) * p.lda;
        f4 acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * q + 4 * g + r;
                acc[q][r] = (p.bias && col < n_here) ? p.bias[n0 + col] : 0.f;
            }
        }
        __syncthreads();                                       // (the previous row block is done with both buffers)
        Raw4 wreg[kItems];
        fetch(wreg, 0);
        commit(wreg, 0);
        AChunk xv = window_a(load_a(arow, 0));
        __syncthreads();
        for (int kc = 0; kc < KB; ++kc) {
            ARaw xn = ARaw{};
            if (kc + 1 < KB) { if (DGN_GEMM_ABL != 1 && DGN_GEMM_ABL != 4) xn = load_a(arow, kc + 1); if (DGN_GEMM_ABL != 2 && DGN_GEMM_ABL != 4) fetch(wreg, kc + 1); }              // next chunk's A and W in flight during the MFMAs
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
            const float* wl = Wc_dyn + (kc & 1) * kBuf + i16 * kKS + 16 * sb + 4 * g;
            // groups of four n-tiles, s outer: consecutive MFMAs go to different accumulators (back-to-back MFMAs on one accumulator
            // wait for each other), and only four weight operands are live at a time
#pragma unroll
            for (int q0 = 0; q0 < NT; q0 += 4) {
                f4 wv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q0 + j < NT) wv[j] = *reinterpret_cast<const f4*>(wl + 16 * (q0 + j) * kKS);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q0 + j < NT) acc[q0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][s], xv.v[sb][s], acc[q0 + j], 0, 0, 0);
            }
            }
            if (kc + 1 < KB) xv = window_a(xn);
            if (kc + 1 < KB && DGN_GEMM_ABL != 2 && DGN_GEMM_ABL != 4) commit(wreg, (kc + 1) & 1);
            if (DGN_GEMM_ABL != 3) __syncthreads();            // chunk kc+1 is staged; everyone is done reading chunk kc
        }
        if (row < p.M) {
            float* crow = p.C + row * p.ldc + n0;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int col = 16 * q + 4 * g;
                if (col + 3 < n_here) *reinterpret_cast<f4u*>(crow + col) = acc[q];
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (col + r < n_here) crow[col + r] = acc[q][r];
                }
            }
        }
    }
}

// ---- 256 x (16 NQ) tile GEMM (C = A W^T + bias, W [n, k]) -------------------------------------------------------------------
// Both operands staged through LDS per 16-k chunk (16-byte copies, fetched into registers before the chunk's MFMAs and committed after
// them); four waves stacked over the rows, each with a 64-row x 16 NQ-column register tile (4 x NQ MFMA tiles, NQ <= 7): the 4 + NQ
// ds_read_b128 of a chunk feed 16 NQ MFMAs -- one operand read per ~10 MFMAs where ts_gemm pays one per four (tools/microbench/
// mfma_peak.hip: that ratio is what caps the fp32 MFMA stream).  The host picks NQ and the number of column tiles with the least padding
// (n = 210: two tiles of 112; 420: four of 112; 225: three of 80).
constexpr int kTileM = 256, kTKS = 20;

// One tile of 64 RT rows (RT <= 4) x 16 NQ columns: rows [row0, row0 + 64 RT) of which only those below row_end are written (the
// rest belong to the next workgroup's range or lie past M).  Wave w owns rows 16 RT w .. 16 RT (w + 1) of the tile.
template <int NQ, int RT>
__device__ __forceinline__ void tile_pass(const GemmParams& p, float* As, float* Bs, int64_t row0, int64_t row_end, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int KB = (p.k + 15) >> 4;
    // this thread's 16-byte pieces of the operand chunks: A rows lr + 64 j (j < RT), W rows lr + 64 j (j < NBJ), k offset c4
    const int lr = tid >> 2, c4 = (tid & 3) * 4;
    constexpr int NBJ = (NQ * 16 + 63) / 64;
    constexpr int kABuf = kTileM * kTKS, kBBuf = NQ * 16 * kTKS;
    Raw4 ra[RT], rb[NBJ];
    auto fetch = [&](int kc) {
        const int k0 = 16 * kc + c4;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            const int64_t r = row0 + lr + 64 * j;
            ra[j] = load4_raw(p.A + min(r, p.M - 1) * p.lda, k0, p.k, r < p.M);
        }
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int rl = lr + 64 * j, c = n0 + rl;
            rb[j] = load4_raw(p.W + (int64_t)min(c, p.n - 1) * p.ldw, k0, p.k, rl < NQ * 16 && c < p.n);
        }
    };
    auto commit = [&](int buf) {
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
    fetch(0);
    commit(0);
    __syncthreads();
    for (int kc = 0; kc < KB; ++kc) {
        if (kc + 1 < KB) fetch(kc + 1);                            // next chunk in flight during the MFMAs
        const float* al = As + (kc & 1) * kABuf + (16 * RT * wave + i16) * kTKS + 4 * g;
        const float* bl = Bs + (kc & 1) * kBBuf + i16 * kTKS + 4 * g;
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
        if (kc + 1 < KB) commit((kc + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t row = row0 + 16 * RT * wave + 16 * rt + i16;
        if (row < row_end) {
            float* crow = p.C + row * p.ldc;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = n0 + 16 * q + 4 * g;
                if (col + 3 < p.n) *reinterpret_cast<f4u*>(crow + col) = acc[rt][q];
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (col + r < p.n) crow[col + r] = acc[rt][q][r];
                }
            }
        }
    }
}

// A workgroup owns the rows [blockIdx.x * rows_per_block, + rows_per_block) (a multiple of 64; the host sizes it so that ONE
// workgroup per resident slot covers all rows: no partial last round of tiles) and walks them in tiles of 256, 192, 128 or 64 rows,
// the range cut into the fewest tiles of nearly equal height.
template <int NQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tile_gemm(const GemmParams p) {   // two workgroups per CU: 256 registers

    __shared__ float As[2 * kTileM * kTKS];
    __shared__ float Bs[2 * NQ * 16 * kTKS];
    const int n0 = blockIdx.y * p.n_slice;                    // n_slice = 16 NQ
    int64_t r = (int64_t)blockIdx.x * p.rows_per_block;
    const int64_t end = min(p.M, r + p.rows_per_block);
    while (r < end) {
        const int left = (int)(end - r), tiles = (left + kTileM - 1) / kTileM;
        const int h = min(4, ((left + tiles - 1) / tiles + 63) >> 6);         // 64-row units of this tile
        switch (h) {
            case 4: tile_pass<NQ, 4>(p, As, Bs, r, end, n0); break;
            case 3: tile_pass<NQ, 3>(p, As, Bs, r, end, n0); break;
            case 2: tile_pass<NQ, 2>(p, As, Bs, r, end, n0); break;
            default: tile_pass<NQ, 1>(p, As, Bs, r, end, n0); break;
        }
        r += 64 * h;
    }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------
constexpr int kWgWaves = 16;
constexpr int kMaxKT = 16;           // k-tiles one wave accumulates (64 registers): k-slice of at most 256 columns

struct WgradParams {
    int64_t M;
    int n, k, kk;                    // n <= 16 * kWgWaves; kk = k + 1: column k of X := 1 (the bias gradient rides along), else kk = k
    const float* G; int64_t ldg;     // [M, n]
    const float* X; int64_t ldx;     // [M, k]
    float* part;                     // [k-slices][slots][NT*16][KT*16] per-workgroup partials
    int k_slice;                     // columns of X per blockIdx.y (a multiple of 16, <= 16 * kMaxKT)
    int slots;
};

template <int KT>
__global__ __launch_bounds__(kWave * kWgWaves) void ts_gemm_wgrad(const WgradParams p) {
    extern __shared__ float lds_g[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int NTn = (p.n + 15) >> 4;
    const int k0 = blockIdx.y * p.k_slice, k_here = min(KT * 16, p.kk - k0);
    const int gs = NTn * 16 + 4, xs = KT * 16 + 4;              // LDS row strides (== 4 mod 8... spreads the rows over the banks)
    const int half = 16 * (gs + xs);                              // floats of one (G strip, X strip) buffer
    auto Gbuf = [&](int b) { return lds_g + b * half; };         // (pointer arithmetic, not a pointer table: a dynamically indexed
    auto Xbuf = [&](int b) { return lds_g + b * half + 16 * gs; };   //  array of pointers would live in scratch memory)
    const int64_t n_strips = (p.M + 15) / 16;
    // a 16-row strip of G (all n columns) and of X (this k slice), zero beyond the matrices: fetched into registers before the
    // current strip's MFMAs, committed to the other LDS buffer after them (see ts_gemm)
    constexpr int kMaxItems = (16 * (kWgWaves * 4 + KT * 4) + kWave * kWgWaves - 1) / (kWave * kWgWaves);
    const int gq = NTn * 4, xq = KT * 4;                         // float4's per staged row
    auto fetch = [&](Raw4 (&reg)[kMaxItems], int64_t strip) {
        const int64_t r0 = strip * 16;
#pragma unroll
        for (int j = 0; j < kMaxItems; ++j) {
            const int it = min(tid + j * kWave * kWgWaves, 16 * (gq + xq) - 1);      // (surplus threads repeat the last piece; commit skips them)
            const int r = it / (gq + xq), c = it - r * (gq + xq);
            const int64_t row = min(r0 + r, p.M - 1);
            const bool is_g = c < gq;
            reg[j] = load4_raw(is_g ? p.G + row * p.ldg : p.X + row * p.ldx, is_g ? 4 * c : k0 + 4 * (c - gq), is_g ? p.n : p.k, r0 + r < p.M);
            if (r0 + r >= p.M) reg[j].sh = 8;                              // (a row past the matrix: zeros, and no ones column either)
        }
    };
    auto commit = [&](const Raw4 (&reg)[kMaxItems], int buf) {
#pragma unroll
        for (int j = 0; j < kMaxItems; ++j) {
            const int it = tid + j * kWave * kWgWaves;
            if (it < 16 * (gq + xq)) {
                const int r = it / (gq + xq), c = it - r * (gq + xq);
                f4 v = load4_window(reg[j]);
                if (c < gq) *reinterpret_cast<f4*>(Gbuf(buf) + r * gs + 4 * c) = v;
                else {
                    if (p.kk > p.k && reg[j].sh < 8) {                  // (sh >= 8 marks a row past the matrix, see fetch)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = k0 + 4 * (c - gq) + e == p.k ? 1.f : v[e];      // column k of X := 1
                    }
                    *reinterpret_cast<f4*>(Xbuf(buf) + r * xs + 4 * (c - gq)) = v;
                }
            }
        }
    };
    f4 acc[KT];
#pragma unroll
    for (int b = 0; b < KT; ++b) acc[b] = f4{0.f, 0.f, 0.f, 0.f};
    const bool has_tile = wave < NTn;
    int64_t strip = blockIdx.x;
    int buf = 0;
    Raw4 sreg[kMaxItems];
    if (strip < n_strips) {
        fetch(sreg, strip);
        commit(sreg, 0);
    }
    __syncthreads();
    for (; strip < n_strips; strip += gridDim.x, buf ^= 1) {
        const bool more = strip + gridDim.x < n_strips;
        if (more) fetch(sreg, strip + gridDim.x);
        if (has_tile) {
            // D[n][k] += G[m][n] X[m][k], the strip's rows are the reduction index: m = 4 s + g in the s-th instruction
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float gv = Gbuf(buf)[(4 * s + g) * gs + 16 * wave + i16];
                const float* xr = Xbuf(buf) + (4 * s + g) * xs + i16;
#pragma unroll
                for (int b = 0; b < KT; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv, xr[16 * b], acc[b], 0, 0, 0);
            }
        }
        if (more) commit(sreg, buf ^ 1);
        __syncthreads();
    }
    // lane holds D[n = 16 wave + 4 g + r][k = 16 b + i16]
    if (has_tile) {
        float* out = p.part + ((int64_t)blockIdx.y * p.slots + blockIdx.x) * (NTn * 16) * (KT * 16);
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * wave + 4 * g + r) * (KT * 16) + 16 * b + i16] = acc[b][r];
    }
}

// dW[n][k] = sum over the workgroup slots, in slot order (bitwise reproducible); thread per element, 4 slots in flight
static __global__ __launch_bounds__(256) void ts_gemm_wgrad_finalize(int n, int k, int kk, int k_slice, int slots, int npad, int kpad,
                                                                      const float* __restrict__ part, float* __restrict__ dW, int64_t lddw,
                                                                      float* __restrict__ dbias) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * kk) return;
    const int r = (int)(e / kk), c = (int)(e - (int64_t)r * kk);
    const int sl = c / k_slice, cc = c - sl * k_slice;
    const float* src = part + (int64_t)sl * slots * npad * kpad + (int64_t)r * kpad + cc;
    // (sixteen partial blocks in flight, four running sums in slot order -- the same association as with four in flight: s_j takes the
    //  slots q = j mod 4 --: with four loads per round trip the 256 slots of a 15 k-row batch made this kernel longer than the product)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int64_t st = (int64_t)npad * kpad;
    int q = 0;
    for (; q + 15 < slots; q += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[(q + j) * st];
#pragma unroll
        for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; q + 3 < slots; q += 4) {
        s0 += src[(int64_t)q * st];
        s1 += src[(int64_t)(q + 1) * st];
        s2 += src[(int64_t)(q + 2) * st];
        s3 += src[(int64_t)(q + 3) * st];
    }
    for (; q < slots; ++q) s0 += src[(int64_t)q * st];
    const float v = (s0 + s1) + (s2 + s3);
    if (c < k) dW[(int64_t)r * lddw + c] = v;
    else if (dbias) dbias[r] = v;
}

// ---- tile weight gradient (32 x 32 x 2 MFMA, the whole [n-block, k-block] accumulator in one workgroup's registers) -----------------
// dW [n, k] = G^T X over M rows: the outputs are small and the reduction index is the long one.  A workgroup of 8 waves (2 over n x 4 over
// k) holds the accumulators of a whole block of up to 256 x 256 outputs -- wave (wn, wk) owns up to 4 x 2 tiles of 32 x 32, 16 registers
// each -- and streams ITS share of the rows through LDS once, 16 rows at a time (both operand strips fetched into registers before the
// strip's MFMAs and committed to the other LDS buffer after them).  Per pair of rows a wave reads 4 + 2 operand values for 8
// v_mfma_f32_32x32x2_f32 (ts_gemm_wgrad: one read per MFMA and G re-read per 256-column k slice).  Tiles are dealt to the waves at run
// time (wave-uniform guards), so n = 210, k = 420 costs 7 x 14 tiles, not 8 x 16.  The bias gradient rides as column k of X := 1
// (`ones`).  Per-workgroup partial blocks, fixed-order finalize: bitwise reproducible.  Any n, k (blocks of up to 256 columns each way).
using f16v = __attribute__((ext_vector_type(16))) float;
constexpr int kTwRows = 16, kTwWaves = 8, kTwWN = 2, kTwWK = 4, kTwNT = 4, kTwKT = 2;

struct TileWgParams {
    int64_t M;
    int n, k, kk;                    // kk = k + 1 with the ones column, else k
    const float* G; int64_t ldg;     // [M, n]
    const float* X; int64_t ldx;     // [M, k]
    float* part;                     // [n_blocks * k_blocks][slots][nb_cols][kb_cols]
    int slots, n_blocks, k_blocks, nb_tiles, kb_tiles;      // tiles of 32 per block (<= 8 each)
};

__host__ __device__ inline int tw_stride(int cols) { return ((cols + 63) / 64) * 64 + 32; }      // == 32 mod 64: the two half-waves of an operand read hit disjoint banks

constexpr int kTwItems = (kTwRows * 128 + kWave * kTwWaves - 1) / (kWave * kTwWaves);          // 256 + 256 columns: 4 float4's per thread and strip

// block geometry of one workgroup (uniform)
struct TwBlock {
    int n0, k0, n_here, k_here, gs, xs, half, gq, per_row, total;
};

// a 16-row strip of G's and X's column blocks, zero beyond the matrices, column k of X := 1 (the bias gradient's ones column)
__device__ __forceinline__ void tw_fetch(Raw4 (&reg)[kTwItems], const TileWgParams& p, const TwBlock& B, int64_t strip, int tid) {
    const int64_t r0 = strip * kTwRows;
#pragma unroll
    for (int j = 0; j < kTwItems; ++j) {
        const int it = min(tid + j * kWave * kTwWaves, B.total - 1);          // (surplus threads repeat the last piece; commit skips them)
        const int r = it / B.per_row, c = it - r * B.per_row;
        const int64_t row = min(r0 + r, p.M - 1);
        const bool is_g = c < B.gq;
        // (absolute columns against the whole row's width: always >= 4 columns to clamp into; X: its real columns, the ones column is set at commit)
        reg[j] = load4_raw(is_g ? p.G + row * p.ldg : p.X + row * p.ldx, (is_g ? B.n0 : B.k0) + 4 * (is_g ? c : c - B.gq), is_g ? p.n : p.k, r0 + r < p.M);
    }
}
__device__ __forceinline__ void tw_commit(const Raw4 (&reg)[kTwItems], float* Gb, const TileWgParams& p, const TwBlock& B, int64_t strip, int tid) {
    float* Xb = Gb + kTwRows * B.gs;
#pragma unroll
    for (int j = 0; j < kTwItems; ++j) {
        const int it = tid + j * kWave * kTwWaves;
        if (it < B.total) {
            const int r = it / B.per_row, c = it - r * B.per_row;
            f4 v = load4_window(reg[j]);
            if (c < B.gq) *reinterpret_cast<f4*>(Gb + r * B.gs + 4 * c) = v;
            else {
                const int gcol = B.k0 + 4 * (c - B.gq);
                if (p.kk > p.k && strip * kTwRows + r < p.M) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gcol + e == p.k ? 1.f : v[e];        // column k of X := 1 on the rows that exist
                }
                *reinterpret_cast<f4*>(Xb + r * B.xs + 4 * (c - B.gq)) = v;
            }
        }
    }
}

// The strip loop of a wave that owns NTc x KTc tiles (n tiles from a0, k tiles from b0): counts as template arguments -- MFMAs under
// run-time guards made the compiler keep copies of the accumulators across the branches (spills).  Every wave of the workgroup runs the
// same number of iterations and barriers whatever its instantiation.
template <int NTc, int KTc>
__device__ __forceinline__ void tw_run(const TileWgParams& p, const TwBlock& B, float* lds, int a0, int b0, int tid) {
    const int lane = tid & 63, i32 = lane & 31, hi = lane >> 5;
    f16v acc[NTc > 0 ? NTc : 1][KTc > 0 ? KTc : 1];
#pragma unroll
    for (int a = 0; a < NTc; ++a)
#pragma unroll
        for (int b = 0; b < KTc; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int64_t n_strips = (p.M + kTwRows - 1) / kTwRows;
    Raw4 sreg[kTwItems];
    if ((int64_t)blockIdx.x < n_strips) {
        tw_fetch(sreg, p, B, blockIdx.x, tid);
        tw_commit(sreg, lds, p, B, blockIdx.x, tid);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t strip = blockIdx.x; strip < n_strips; strip += gridDim.x, buf ^= 1) {
        const bool more = strip + gridDim.x < n_strips;
        if (more) tw_fetch(sreg, p, B, strip + gridDim.x, tid);
        const float* Gb = lds + buf * B.half + 32 * a0 + i32;
        const float* Xb = lds + buf * B.half + kTwRows * B.gs + 32 * b0 + i32;
        // D[n][k] += G[m][n] X[m][k]: the pair of rows m = 2 s + hi is the reduction index of the s-th instruction
#pragma unroll 2
        for (int s = 0; s < kTwRows / 2; ++s) {
            const int r = 2 * s + hi;
            float gv[NTc > 0 ? NTc : 1], xv[KTc > 0 ? KTc : 1];
#pragma unroll
            for (int a = 0; a < NTc; ++a) gv[a] = Gb[r * B.gs + 32 * a];
#pragma unroll
            for (int b = 0; b < KTc; ++b) xv[b] = Xb[r * B.xs + 32 * b];
#pragma unroll
            for (int a = 0; a < NTc; ++a)
#pragma unroll
                for (int b = 0; b < KTc; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[a], xv[b], acc[a][b], 0, 0, 0);
        }
        if (more) tw_commit(sreg, lds + (buf ^ 1) * B.half, p, B, strip + gridDim.x, tid);
        __syncthreads();
    }
    // lane holds D[32 a + (r & 3) + 8 (r >> 2) + 4 hi][32 b + i32] of its tiles: the workgroup's partial block goes to its slot
    const int kbc = p.kb_tiles * 32;
    float* out = p.part + ((int64_t)blockIdx.y * p.slots + blockIdx.x) * (p.nb_tiles * 32) * kbc;
#pragma unroll
    for (int a = 0; a < NTc; ++a)
#pragma unroll
        for (int b = 0; b < KTc; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(int64_t)(32 * (a0 + a) + (r & 3) + 8 * (r >> 2) + 4 * hi) * kbc + 32 * (b0 + b) + i32] = acc[a][b][r];
}

__global__ __launch_bounds__(kWave * kTwWaves) void tile_wgrad(const TileWgParams p) {
    extern __shared__ float lds_tw[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int bn = blockIdx.y / p.k_blocks, bk = blockIdx.y - bn * p.k_blocks;
    TwBlock B;
    B.n0 = bn * p.nb_tiles * 32; B.k0 = bk * p.kb_tiles * 32;
    B.n_here = min(p.nb_tiles * 32, p.n - B.n0); B.k_here = min(p.kb_tiles * 32, p.kk - B.k0);      // columns of this block that exist
    B.gs = tw_stride(p.nb_tiles * 32); B.xs = tw_stride(p.kb_tiles * 32);
    B.half = kTwRows * (B.gs + B.xs);
    B.gq = (B.n_here + 3) >> 2;
    B.per_row = B.gq + ((B.k_here + 3) >> 2);
    B.total = kTwRows * B.per_row;
    // this wave's tiles: n tiles [a0, a0 + nt_w), k tiles [b0, b0 + kt_w) of the block, dealt as evenly as possible
    const int nt_blk = (B.n_here + 31) >> 5, kt_blk = (B.k_here + 31) >> 5;
    const int wn = wave / kTwWK, wk = wave - wn * kTwWK;
    const int a0 = (nt_blk * wn) / kTwWN, nt_w = (nt_blk * (wn + 1)) / kTwWN - a0;
    const int b0 = (kt_blk * wk) / kTwWK, kt_w = (kt_blk * (wk + 1)) / kTwWK - b0;
    switch ((nt_w > 0 && kt_w > 0) ? nt_w * 4 + kt_w : 0) {
        case 4 + 1: tw_run<1, 1>(p, B, lds_tw, a0, b0, tid); break;
        case 4 + 2: tw_run<1, 2>(p, B, lds_tw, a0, b0, tid); break;
        case 8 + 1: tw_run<2, 1>(p, B, lds_tw, a0, b0, tid); break;
        case 8 + 2: tw_run<2, 2>(p, B, lds_tw, a0, b0, tid); break;
        case 12 + 1: tw_run<3, 1>(p, B, lds_tw, a0, b0, tid); break;
        case 12 + 2: tw_run<3, 2>(p, B, lds_tw, a0, b0, tid); break;
        case 16 + 1: tw_run<4, 1>(p, B, lds_tw, a0, b0, tid); break;
        case 16 + 2: tw_run<4, 2>(p, B, lds_tw, a0, b0, tid); break;
        default: tw_run<0, 0>(p, B, lds_tw, a0, b0, tid); break;       // a wave without tiles still stages and syncs
    }
}

// dW[n][k] (and dbias[n] = column k of the padded product) = the slot sums in slot order; a block covers 64 consecutive elements, its
// sixteen waves take every sixteenth slot, LDS joins them (as dgn_linear's finalize)
static __global__ __launch_bounds__(64 * 16) void tile_wgrad_finalize(int n, int k, int kk, int slots, int k_blocks, int nbc, int kbc,
                                                                      const float* __restrict__ part, float* __restrict__ dW, int64_t lddw,
                                                                      float* __restrict__ dbias) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const bool live = e < (int64_t)n * kk;
    int r = 0, c = 0;
    float s0 = 0.f, s1 = 0.f;
    if (live) {
        r = (int)(e / kk);
        c = (int)(e - (int64_t)r * kk);
        const int bn = r / nbc, bk = c / kbc;
        const float* src = part + (int64_t)(bn * k_blocks + bk) * slots * nbc * kbc + (int64_t)(r - bn * nbc) * kbc + (c - bk * kbc);
        int q = sg;
        for (; q + 16 < slots; q += 32) {
            s0 += src[(int64_t)q * nbc * kbc];
            s1 += src[(int64_t)(q + 16) * nbc * kbc];
        }
        if (q < slots) s0 += src[(int64_t)q * nbc * kbc];
    }
    red[sg][lane] = s0 + s1;
    __syncthreads();
    if (live && sg == 0) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w][lane];
        if (c < k) dW[(int64_t)r * lddw + c] = v;
        else if (dbias) dbias[r] = v;
    }
}

}  // namespace gemm
}  // namespace dgn
