// The posttrans product INSIDE the sweep (reference: realworld_benchmark/nets/dgn_layer.py:237-249 ->
// :266-271, i.e. reduce_func's [n, A*S*F] row feeding posttrans): one persistent workgroup per CU takes 64 destination rows per
// iteration,
//   1. its 16 waves run the short-row sweep on 4 rows each (the code of agg_fwd_short) but LEAVE the finished aggregate rows
//      in LDS, tower-major per row ([tower][aggregator][Ft]);
//   2. wave u owns one 16 x K tile of the posttrans weights, W[t][16q .. 16q+15][:], in REGISTERS for the life of the kernel
//      (K / 4 <= 24 floats per lane) and multiplies the four 16-row strips with v_mfma_f32_16x16x4_f32 (exact fp32), the
//      strip read from LDS in the k-order of dgn_linear_kernels.hpp (one ds_read_b128 per 16 k);
//   3. the [16][T][S*fo] products meet in a small LDS buffer and leave as y = snorm * (b + sum_s scale_s * z_s) rows.
// The [N, A*F] aggregate rows (1 680 B per node on ZINC, 0.46 GB per pass) never reach memory.  Molecule-like graphs only
// (short rows, no hub rows), one feature tile, up to 16 (tower, n-tile) units.
#pragma once
#include "dgn_agg_kernels.hpp"

namespace dgn {

using f4 = __attribute__((ext_vector_type(4))) float;

#ifndef DGN_FUSED_WAVES
#define DGN_FUSED_WAVES 16
#endif
constexpr int kFusedWaves = DGN_FUSED_WAVES;                 // 8: two workgroups per CU alternate between their sweep and MFMA phases
constexpr int kFusedRows = kFusedWaves * kShortRows;         // destination rows per iteration
constexpr int kFusedUnits = 16 / kFusedWaves;                // (tower, n-tile) weight tiles per wave: 16 units in all
constexpr int kFusedKB = 6;          // 16-k blocks a wave keeps of its weight tile: K <= 96

struct FusedParams {
    AggParams a;                      // the sweep (a.out unused)
    const float* W; int64_t ldw, sW;  // posttrans weights [T][S*fo][K]
    const float* sc;                  // [N, S] or NULL (S == 1)
    const float* rs;                  // [N] or NULL
    const float* cb;                  // [T*fo] or NULL
    int S, fo, nq;                    // scalers, per-tower output width, n-tiles per tower
    float* Y; int64_t ldy;
    int64_t n_iters;
    // wgrad twin (layer_wgrad_fused): G = gy expanded by the scalers, dW partial slots
    const float* gy; int64_t s_gy;    // [T][N][fo]
    float* part;                      // [T][slots][nq*16][kFusedKB*16]
    int dbg;                          // experiments (DGN_FUSED_DBG): 1 = sweep only, 2 = product + combine only
};

__host__ __device__ inline int fused_tk(const AggParams& a) { return a.n_towers * a.agg_total * a.Ft; }
__host__ __device__ inline size_t fused_lds_floats(const AggParams& a, int nq) {
    const int tk = fused_tk(a), zw = a.n_towers * nq * 16;
    return (size_t)kFusedRows * (tk > zw ? tk : zw) + kFusedRows * 4;
}

template <class C, class O>
__global__ __launch_bounds__(kWave * kFusedWaves) void layer_fwd_fused(const FusedParams p) {
    constexpr int VEC = C::VEC;
    extern __shared__ float lds_f[];
    const AggParams& a = p.a;
    const int T = a.n_towers, K = a.agg_total * a.Ft, TK = T * K, NQ = p.nq, ZW = T * NQ * 16;
    float* AX = lds_f;                                   // [64][T][K]
    float* FAC = AX + kFusedRows * max(TK, ZW);          // [rows][4]: scale_0..2, row_scale  (the products [rows][T][NQ*16] reuse AX)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    // this wave's weight tiles: unit u = wave + j * kFusedWaves = (tower, n-tile); lane (i16, g) keeps W[t][16q + i16][16b + 4g + s]
    int t_u[kFusedUnits], q_u[kFusedUnits];
    bool has_unit[kFusedUnits], w_live[kFusedUnits];
    const float* wrow[kFusedUnits];
    const bool w16 = (p.ldw & 3) == 0 && (p.sW & 3) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0;
#pragma unroll
    for (int j = 0; j < kFusedUnits; ++j) {
        const int u = wave + j * kFusedWaves;
        has_unit[j] = u < T * NQ;
        t_u[j] = has_unit[j] ? u / NQ : 0;
        q_u[j] = has_unit[j] ? u - t_u[j] * NQ : 0;
        const int n_u = 16 * q_u[j] + i16;
        w_live[j] = has_unit[j] && n_u < p.S * p.fo;
        wrow[j] = p.W + (int64_t)t_u[j] * p.sW + (int64_t)(w_live[j] ? n_u : 0) * p.ldw + 4 * g;
    }
    const int f0 = lane * VEC;
    const bool active = f0 < a.F;
    const int fo2 = p.fo >> 1;
    // contiguous range of iterations per workgroup: neighbouring rows (the molecules of a batch) share sources in one L2
    const int64_t per = (p.n_iters + gridDim.x - 1) / gridDim.x;
    const int64_t it0 = (int64_t)blockIdx.x * per, it1 = min(p.n_iters, it0 + per);
    for (int64_t it = it0; it < it1; ++it) {
        const int64_t row_base = it * kFusedRows;
        if (tid < kFusedRows) {                           // the rows' combine factors ride along with the sweep's loads
            const int64_t row = min(row_base + tid, a.n_nodes - 1);
            f4 f = f4{1.f, 1.f, 1.f, 1.f};
            if (p.sc) {
                f[0] = p.sc[row * p.S];
                f[1] = p.sc[row * p.S + min(1, p.S - 1)];
                f[2] = p.sc[row * p.S + min(2, p.S - 1)];
            }
            if (p.rs) f[3] = p.rs[row];
            *reinterpret_cast<f4*>(FAC + 4 * tid) = f;
        }
        ShortGroup grp;
        if (p.dbg != 2 && grp.init_at(a, row_base + (int64_t)wave * kShortRows))
            short_group_to_lds<C, O>(a, grp, f0, active, AX + (wave * kShortRows) * TK, TK);
        if (p.dbg == 1) { __syncthreads(); continue; }
        // The 16 x K weight tile of a unit (70 KB for all units together: L2 resident) is fetched per iteration, AFTER the sweep
        // (kept in registers across the sweep it pushed every configuration past 128 VGPRs) and, for the first unit, BEFORE the
        // barrier, so that its latency overlaps the other waves' last rows.
        auto load_w = [&](float (&wreg)[kFusedKB][4], int j) {
#pragma unroll
            for (int b = 0; b < kFusedKB; ++b) {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) wreg[b][s2] = 0.f;
                if (w_live[j] && 16 * b + 4 * g < K) {
                    if (w16) {
                        const f4 v = *reinterpret_cast<const f4*>(wrow[j] + 16 * b);      // (K % 4 == 0: the four k's are all inside the row)
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) wreg[b][s2] = v[s2];
                    } else {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) wreg[b][s2] = wrow[j][16 * b + s2];
                    }
                }
            }
        };
        float wreg0[kFusedKB][4];
        load_w(wreg0, 0);
        __syncthreads();
        // all strips back to back (independent accumulators keep the MFMA pipe busy); the products then take the place of the
        // aggregate rows, which are dead once every wave is through
        f4 acc[kFusedUnits][kFusedRows / 16];
#pragma unroll
        for (int j = 0; j < kFusedUnits; ++j) {
            float wreg[kFusedKB][4];
            if (j == 0) {
#pragma unroll
                for (int b = 0; b < kFusedKB; ++b)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) wreg[b][s2] = wreg0[b][s2];
            } else {
                load_w(wreg, j);
            }
#pragma unroll
            for (int strip = 0; strip < kFusedRows / 16; ++strip) acc[j][strip] = f4{0.f, 0.f, 0.f, 0.f};
            if (has_unit[j]) {
#pragma unroll
                for (int b = 0; b < kFusedKB; ++b) {
                    if (16 * b < K) {
                        f4 xv[kFusedRows / 16];
#pragma unroll
                        for (int strip = 0; strip < kFusedRows / 16; ++strip)
                            xv[strip] = *reinterpret_cast<const f4*>(AX + (strip * 16 + i16) * TK + t_u[j] * K + 4 * g + 16 * b);
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) {
                            const bool live = 16 * b + 4 * g + s2 < K;      // (past K: another row's data, and 0 * inf is not 0)
#pragma unroll
                            for (int strip = 0; strip < kFusedRows / 16; ++strip)
                                acc[j][strip] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[b][s2], live ? xv[strip][s2] : 0.f, acc[j][strip], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();                                  // AX is dead: Z (64 rows) goes over it
#pragma unroll
        for (int j = 0; j < kFusedUnits; ++j) {
            if (has_unit[j]) {
#pragma unroll
                for (int strip = 0; strip < kFusedRows / 16; ++strip)      // lane (m = i16, g) holds z[m][16 q + 4 g .. + 3]
                    *reinterpret_cast<f4*>(AX + (strip * 16 + i16) * ZW + t_u[j] * NQ * 16 + 16 * q_u[j] + 4 * g) = acc[j][strip];
            }
        }
        __syncthreads();
        // y[m][t*fo + o] = rs * (cb + sum_s sc_s * z[s*fo + o]): a thread per (row, tower, output pair)
        for (int idx = tid; idx < kFusedRows * T * fo2; idx += blockDim.x) {
            const int m = idx / (T * fo2), rem = idx - m * (T * fo2), t = rem / fo2, o = 2 * (rem - t * fo2);
            const int64_t row = row_base + m;
            if (row < a.n_nodes) {
                const f4 f = *reinterpret_cast<const f4*>(FAC + 4 * m);
                float2 v = p.cb ? *reinterpret_cast<const float2*>(p.cb + t * p.fo + o) : make_float2(0.f, 0.f);
                const float* z = AX + m * ZW + t * NQ * 16 + o;
                for (int s = 0; s < p.S; ++s) {
                    const float2 zz = *reinterpret_cast<const float2*>(z + s * p.fo);
                    v.x += f[s] * zz.x;
                    v.y += f[s] * zz.y;
                }
                *reinterpret_cast<float2*>(p.Y + row * p.ldy + t * p.fo + o) = make_float2(v.x * f[3], v.y * f[3]);
            }
        }
        __syncthreads();                                  // the next iteration's sweep overwrites AX / FAC
    }
}

}  // namespace dgn
