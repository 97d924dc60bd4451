// Block-diagonal pretrans Linear of the towers layer (include/dgn_hip.h, dgn_linear_bd_*).
//
// With divide_input every tower's pretrans acts on its own f_in-column slice of h (nets/dgn_layer.py:226-231 called from :309-316), so
// the fused P|Q product  pq = h [W_s | W_d]^T + [0 | b]  has a block-diagonal weight matrix: T blocks of [f_in, f_in] in each half,
// 1/T of the dense [2 T f_in, T f_in] matrix (80 % structural zeros for the five towers of the ZINC configuration).  The dense
// streaming kernels (ts_linear / ts_wgrad) multiply the zeros; these kernels walk the towers in a static loop instead:
//
//   bd_forward          pq[m][j Fm + t fi + a] = bias + sum_b h[m][t fi + b] W[j Fm + t fi + a][t fi + b]         j = 0 (P), 1 (Q)
//   bd_backward_input   gh[m][t fi + b] = (add1 + sum_{j,a} g[m][j Fm + t fi + a] W[j Fm + t fi + a][t fi + b]) + add2
//   bd_wgrad            dW[j Fm + t fi + a][t fi + b] = sum_m g[m][j Fm + t fi + a] h[m][t fi + b];   dbias[r] = sum_m g[m][r]
//
// Same strip pipeline as ts_linear (16-row strips per wave: global -> registers -> LDS, results leave through LDS as one contiguous
// run one iteration later), same exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), but 2 T FB^2 x 4 MFMAs per strip instead of
// ceil(2 Fm / 16) ceil(Fm / 16) x 4 (ZINC towers: 40 instead of 180), and a tower's operand tile is its own 16 columns (the columns past
// f_in are masked to zero: they belong to the next tower).  The strip and its results share one LDS region per wave (the results are
// written when the operand is dead), so 16 waves per CU fit.  W keeps the dense [2 Fm, ldw] layout of the fused operand buffer: only
// the diagonal blocks are read; bd_wgrad writes the diagonal blocks and ZEROS the rest.
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

struct BdParams {
    int64_t M;
    int T, fi;
    const float* A;                                  // forward: h [M][Fm]; input gradient / weight gradient: g [M][2 Fm]
    const float* X;                                  // weight gradient: h [M][Fm]
    const float* W; int64_t ldw;
    const float* bias;                               // [2 Fm] or NULL (forward)
    float* C;                                        // forward: [M][2 Fm]; input gradient: [M][Fm]
    const float* add1; const float* add2;            // input gradient epilogue (NULL = absent)
    float* part;                                     // weight gradient: [groups][tiles][16][16]
    int groups;
};

// threads per workgroup = the register budget per lane (1024: 128, 768: 168, 512: 256, 256: 512), by the number of accumulator tiles
// and prefetched strip registers of the shape (checked by the build: no kernel may spill)
constexpr int bd_fwd_threads(int T, int FB) { return T * FB <= 3 ? 1024 : (T * FB <= 6 ? 768 : (T * FB <= 8 ? 512 : 256)); }
constexpr int bd_bwd_threads(int T, int FB) { return T * FB <= 2 ? 1024 : (T * FB <= 5 ? 512 : 256); }
constexpr int bd_wg_waves_per_simd(int T, int FB) { return 2 * T * FB * FB <= 10 ? 2 : 1; }

__device__ __forceinline__ f4 bd_operand(const float* x, int valid) {      // four consecutive floats from an 8-byte aligned address,
    const float2 lo = *reinterpret_cast<const float2*>(x), hi = *reinterpret_cast<const float2*>(x + 2);      // entries >= valid zeroed
    return f4{valid > 0 ? lo.x : 0.f, valid > 1 ? lo.y : 0.f, valid > 2 ? hi.x : 0.f, valid > 3 ? hi.y : 0.f};
}

// a strip of `width`-float rows staged in U (16 * width floats, contiguous) -> global, 16-byte lanes for full strips
template <int NLC, bool ADD>
__device__ __forceinline__ void bd_store_out(const float* U, float* dst, int width, int rows, int lane, const float2 (&pe1)[ADD ? NLC : 1],
                                             const float2 (&pe2)[ADD ? NLC : 1], bool has2) {
    const int cnt2 = rows * (width >> 1);
    if (rows == kStrip) {
#pragma unroll
        for (int jq = 0; jq < NLC / 2; ++jq) {
            if (jq * 64 + lane < (kStrip / 4) * width) {
                float4 c = reinterpret_cast<const float4*>(U)[jq * 64 + lane];
                if constexpr (ADD) {
                    c = make_float4(pe1[2 * jq].x + c.x, pe1[2 * jq].y + c.y, pe1[2 * jq + 1].x + c.z, pe1[2 * jq + 1].y + c.w);
                    if (has2) c = make_float4(c.x + pe2[2 * jq].x, c.y + pe2[2 * jq].y, c.z + pe2[2 * jq + 1].x, c.w + pe2[2 * jq + 1].y);
                }
                reinterpret_cast<float4*>(dst)[jq * 64 + lane] = c;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NLC; ++j) {
            const int i2 = ADD ? tail_idx2(j, lane) : j * 64 + lane;
            if (i2 < cnt2) {
                float2 c = reinterpret_cast<const float2*>(U)[i2];
                if constexpr (ADD) {
                    c = make_float2(pe1[j].x + c.x, pe1[j].y + c.y);
                    if (has2) c = make_float2(c.x + pe2[j].x, c.y + pe2[j].y);
                }
                reinterpret_cast<float2*>(dst)[i2] = c;
            }
        }
    }
}

template <int T, int FI>
__global__ __launch_bounds__(bd_fwd_threads(T, (FI + 15) / 16)) void bd_forward(BdParams p) {
    extern __shared__ float lds[];
    constexpr int FB = (FI + 15) / 16, fi = FI;
    constexpr int KPB = 16 * FB + 4, NTL = T * 2 * FB, NL = 2 * T * FB, NLC = 4 * T * FB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    constexpr int k = T * fi, n = 2 * k;
    float* Wl = lds;                                 // [T][2][FB][16][KPB]: tile (t, j, q) row = output column 16 q + row of the tower
    float* Bl = Wl + NTL * 16 * KPB;                 // [T][2][FB][16]
    float* U = Bl + NTL * 16 + wave * (kStrip * n);  // this wave's strip (16 k floats), later its results (16 n floats)
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    const int64_t first = (int64_t)blockIdx.x * n_waves + wave, step = (int64_t)p.groups * n_waves;
    float2 pre[NL];
    if (first < n_strips) load_strip<NL>(pre, p.A, p.M, k, first, lane);
    for (int i = tid; i < NTL * 16 * KPB + NTL * 16; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 2 * T * fi * fi; i += blockDim.x) {
        const int b = i % fi, a = (i / fi) % fi, tj = i / (fi * fi), t = tj >> 1, j = tj & 1;
        Wl[((tj * FB + (a >> 4)) * 16 + (a & 15)) * KPB + b] = p.W[(int64_t)(j * k + t * fi + a) * p.ldw + t * fi + b];
    }
    if (p.bias)
        for (int i = tid; i < 2 * T * fi; i += blockDim.x) {
            const int a = i % fi, tj = i / fi, t = tj >> 1, j = tj & 1;
            Bl[(tj * FB + (a >> 4)) * 16 + (a & 15)] = p.bias[j * k + t * fi + a];
        }
    __syncthreads();

    const int m = lane & 15, g = lane >> 4;
    int64_t out_strip = -1;
    const float2 none[1] = {};
    for (int64_t strip = first; strip < n_strips; strip += step) {
        if (out_strip >= 0)
            bd_store_out<NLC, false>(U, p.C + out_strip * kStrip * n, n, (int)min((int64_t)kStrip, p.M - out_strip * kStrip), lane, none, none, false);
        store_strip<NL>(U, pre, k, lane);
        if (strip + step < n_strips) load_strip<NL>(pre, p.A, p.M, k, strip + step, lane);
        f4 acc[NTL];
#pragma unroll
        for (int q = 0; q < NTL; ++q) acc[q] = *reinterpret_cast<const f4*>(Bl + 16 * q + 4 * g);
        const float* xrow = U + m * k + 4 * g;
        // (tower, block) pairs one after the other, the next pair's LDS operands requested before the current pair's MFMAs are issued;
        // the scheduling barriers keep the compiler from hoisting every pair's operands to the top (spills at 128 registers)
        f4 xn = bd_operand(xrow, fi - 4 * g), wn[2 * FB];
#pragma unroll
        for (int q = 0; q < 2 * FB; ++q) wn[q] = *reinterpret_cast<const f4*>(Wl + ((q * 16) + m) * KPB + 4 * g);
#pragma unroll
        for (int tb = 0; tb < T * FB; ++tb) {
            const int t = tb / FB;
            const f4 xv = xn;
            f4 wv[2 * FB];
#pragma unroll
            for (int q = 0; q < 2 * FB; ++q) wv[q] = wn[q];
            if (tb + 1 < T * FB) {
                const int t1 = (tb + 1) / FB, b1 = (tb + 1) % FB;
                xn = bd_operand(xrow + t1 * fi + 16 * b1, fi - 16 * b1 - 4 * g);
#pragma unroll
                for (int q = 0; q < 2 * FB; ++q) wn[q] = *reinterpret_cast<const f4*>(Wl + (((t1 * 2 * FB + q) * 16) + m) * KPB + 16 * b1 + 4 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 2 * FB; ++q)
                    acc[t * 2 * FB + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q][s], xv[s], acc[t * 2 * FB + q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // lane (m, g) holds pq[row0 + m][j k + t fi + 16 q + 4 g .. + 3]; the operand strip is dead: its place takes the results
        float* c = U + m * n + 4 * g;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < FB; ++q) {
                    const f4 v = acc[(t * 2 + j) * FB + q];
                    const int col = 16 * q + 4 * g;
                    float* d = c + j * k + t * fi + 16 * q;
                    if (col < fi) *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
                    if (col + 2 < fi) *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
                }
        out_strip = strip;
    }
    if (out_strip >= 0)
        bd_store_out<NLC, false>(U, p.C + out_strip * kStrip * n, n, (int)min((int64_t)kStrip, p.M - out_strip * kStrip), lane, none, none, false);
}

template <int T, int FI>
__global__ __launch_bounds__(bd_bwd_threads(T, (FI + 15) / 16)) void bd_backward_input(BdParams p) {
    extern __shared__ float lds[];
    constexpr int FB = (FI + 15) / 16, fi = FI;
    constexpr int KP2 = 32 * FB + 4, NTL = T * FB, NL = 4 * T * FB, NLC = 2 * T * FB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    constexpr int n = T * fi, k2 = 2 * n;
    float* Wl = lds;                                 // [T][FB][16][KP2]: tile (t, q) row = output column c, k index = j 16 FB + r
    float* U = Wl + NTL * 16 * KP2 + wave * strip_floats(k2);
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    const int64_t first = (int64_t)blockIdx.x * n_waves + wave, step = (int64_t)p.groups * n_waves;
    float2 pre[NL];
    if (first < n_strips) load_strip<NL>(pre, p.A, p.M, k2, first, lane);
    for (int i = tid; i < NTL * 16 * KP2; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 2 * T * fi * fi; i += blockDim.x) {
        const int c = i % fi, r = (i / fi) % fi, tj = i / (fi * fi), t = tj >> 1, j = tj & 1;
        Wl[((t * FB + (c >> 4)) * 16 + (c & 15)) * KP2 + j * 16 * FB + r] = p.W[(int64_t)(j * n + t * fi + r) * p.ldw + t * fi + c];
    }
    __syncthreads();

    const int m = lane & 15, g = lane >> 4;
    const bool has1 = p.add1 != nullptr, has2 = p.add2 != nullptr;
    float2 pe1[NLC], pe2[NLC];
    auto load_adds = [&](int64_t s_) {               // (an absent operand reads zeros and is not used)
        load_strip<NLC>(pe1, p.add1, p.M, n, s_, lane, has1);
        load_strip<NLC>(pe2, p.add2, p.M, n, s_, lane, has2);
    };
    int64_t out_strip = -1;
    auto store_out = [&]() {
        const int rows = (int)min((int64_t)kStrip, p.M - out_strip * kStrip);
        if (has1) bd_store_out<NLC, true>(U, p.C + out_strip * kStrip * n, n, rows, lane, pe1, pe2, has2);
        else {
            const float2 none[1] = {};
            bd_store_out<NLC, false>(U, p.C + out_strip * kStrip * n, n, rows, lane, none, none, false);
        }
    };
    for (int64_t strip = first; strip < n_strips; strip += step) {
        if (out_strip >= 0) store_out();
        store_strip<NL>(U, pre, k2, lane);
        load_adds(strip);                            // (unconditional: a load inside a branch makes the join wait for it)
        if (strip + step < n_strips) load_strip<NL>(pre, p.A, p.M, k2, strip + step, lane);
        f4 acc[NTL];
#pragma unroll
        for (int q = 0; q < NTL; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
        const float* xrow = U + m * k2 + 4 * g;
        // (tower, half, block) triples one after the other, software-pipelined like bd_forward's
        f4 xn = bd_operand(xrow, fi - 4 * g), wn[FB];
#pragma unroll
        for (int q = 0; q < FB; ++q) wn[q] = *reinterpret_cast<const f4*>(Wl + (q * 16 + m) * KP2 + 4 * g);
#pragma unroll
        for (int u = 0; u < T * 2 * FB; ++u) {
            const int t = u / (2 * FB);
            const f4 xv = xn;
            f4 wv[FB];
#pragma unroll
            for (int q = 0; q < FB; ++q) wv[q] = wn[q];
            if (u + 1 < T * 2 * FB) {
                const int t1 = (u + 1) / (2 * FB), j1 = ((u + 1) / FB) % 2, b1 = (u + 1) % FB;
                xn = bd_operand(xrow + j1 * n + t1 * fi + 16 * b1, fi - 16 * b1 - 4 * g);
#pragma unroll
                for (int q = 0; q < FB; ++q) wn[q] = *reinterpret_cast<const f4*>(Wl + ((t1 * FB + q) * 16 + m) * KP2 + j1 * 16 * FB + 16 * b1 + 4 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < FB; ++q)
                    acc[t * FB + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q][s], xv[s], acc[t * FB + q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float* c = U + m * n + 4 * g;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < FB; ++q) {
                const f4 v = acc[t * FB + q];
                const int col = 16 * q + 4 * g;
                float* d = c + t * fi + 16 * q;
                if (col < fi) *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
                if (col + 2 < fi) *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
            }
        out_strip = strip;
    }
    if (out_strip >= 0) store_out();
}

// One wave = one partial sum of all diagonal blocks over its strips (2 T FB^2 accumulator tiles); the bias gradient rides as a column
// of ones at local column f_in of every tower's operand tile (f_in % 16 != 0).  Four waves per workgroup add up in LDS in wave order.
template <int T, int FI>
__global__ __launch_bounds__(256, bd_wg_waves_per_simd(T, (FI + 15) / 16)) void bd_wgrad(BdParams p) {
    extern __shared__ float lds[];
    constexpr int FB = (FI + 15) / 16, fi = FI;
    constexpr int NLG = 4 * T * FB, NLX = 2 * T * FB, NTL = 2 * T * FB * FB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    constexpr int k = T * fi, n2 = 2 * k;
    float* Gl = lds + wave * (strip_floats(n2) + strip_floats(k));
    float* Xl = Gl + strip_floats(n2);
    for (int i = lane; i < strip_floats(n2) + strip_floats(k); i += 64) Gl[i] = 0.f;
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    const int64_t first = (int64_t)blockIdx.x * n_waves + wave, step = (int64_t)p.groups * n_waves;
    float2 pg[NLG], px[NLX];
    f4 acc[NTL];
#pragma unroll
    for (int q = 0; q < NTL; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, mq = lane >> 4;
    if (first < n_strips) {
        load_strip<NLG>(pg, p.A, p.M, n2, first, lane);
        load_strip<NLX>(px, p.X, p.M, k, first, lane);
    }
    for (int64_t strip = first; strip < n_strips; strip += step) {
        const int rows = (int)min((int64_t)kStrip, p.M - strip * kStrip);
        if (rows < kStrip) {                         // rows past the end contribute zero
#pragma unroll
            for (int j = 0; j < NLG; ++j)
                if (strip_idx2(j, lane) >= rows * (n2 >> 1)) pg[j] = make_float2(0.f, 0.f);
        }
        store_strip<NLG>(Gl, pg, n2, lane);
        store_strip<NLX>(Xl, px, k, lane);
        if (strip + step < n_strips) {
            load_strip<NLG>(pg, p.A, p.M, n2, strip + step, lane);
            load_strip<NLX>(px, p.X, p.M, k, strip + step, lane);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* grow = Gl + (4 * mq + s) * n2 + i16;
            const float* xr = Xl + (4 * mq + s) * k + i16;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float xv[FB], gv[2 * FB];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    xv[b] = xr[t * fi + 16 * b];
                    if (16 * b + i16 == fi) xv[b] = 1.f;                           // column f_in of the tile := 1: dW[:, f_in] = sum_m g[m, :]
                }
#pragma unroll
                for (int ja = 0; ja < 2 * FB; ++ja) gv[ja] = grow[(ja / FB) * k + t * fi + 16 * (ja % FB)];
#pragma unroll
                for (int ja = 0; ja < 2 * FB; ++ja)
#pragma unroll
                    for (int b = 0; b < FB; ++b)
                        acc[(t * 2 * FB + ja) * FB + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[ja], xv[b], acc[(t * 2 * FB + ja) * FB + b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);        // (one row quartet's operands at a time: hoisting all four costs the second wave per SIMD)
        }
    }
    // lane holds tile (t, j, a, b) entries [4 mq + r][i16]
    __syncthreads();
    float* red = lds;                                // [NTL][16][16]
    for (int w = 0; w < n_waves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < NTL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = red + q * 256 + (4 * mq + r) * 16 + i16;
                    *d = w == 0 ? acc[q][r] : *d + acc[q][r];
                }
        }
        __syncthreads();
    }
    float* out = p.part + (int64_t)blockIdx.x * NTL * 256;
    for (int i = tid; i < NTL * 64; i += blockDim.x) reinterpret_cast<f4*>(out)[i] = reinterpret_cast<const f4*>(red)[i];
}

// Round 6: the input gradient AND the weight gradient of the block-diagonal product in ONE pass over g = d(P|Q): bd_backward_input and
// bd_wgrad each streamed the [M][2 Fm] gradient (both run at 5-6 TB/s: nothing left inside either), together they read it once --
// 6 Fm instead of 8 Fm floats per row.  A wave stages the g strip and the h strip, issues bd_backward_input's MFMAs (rows of g against the
// towers' W blocks), then bd_wgrad's (g^T h over the strip's rows, the ones column for the bias gradient), and writes the input-gradient
// rows over the dead g strip; they leave one iteration later with the two epilogue operands added.  Same arithmetic, same order of
// accumulation per input-gradient output as bd_backward_input (bit-identical); the weight-gradient partials meet in another order than bd_wgrad's
// (other strip ownership: fp32 rounding, reproducible); 512 threads, one workgroup per CU.
template <int T, int FI>
__global__ __launch_bounds__(512) void bd_backward_both(BdParams p) {
    extern __shared__ float lds[];
    constexpr int FB = (FI + 15) / 16, fi = FI;
    constexpr int KP2 = 32 * FB + 4, NTL = T * FB, NLG = 4 * T * FB, NLX = 2 * T * FB, NLC = 2 * T * FB, NTW = 2 * T * FB * FB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    constexpr int n = T * fi, k2 = 2 * n;
    float* Wl = lds;                                 // [T][FB][16][KP2] as in bd_backward_input
    float* U = Wl + NTL * 16 * KP2 + wave * (strip_floats(k2) + strip_floats(n));      // this wave's g strip, later its input-gradient rows
    float* Xl = U + strip_floats(k2);                // this wave's h strip
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    const int64_t first = (int64_t)blockIdx.x * n_waves + wave, step = (int64_t)p.groups * n_waves;
    float2 pg[NLG], px[NLX];
    if (first < n_strips) {
        load_strip<NLG>(pg, p.A, p.M, k2, first, lane);
        load_strip<NLX>(px, p.X, p.M, n, first, lane);
    }
    for (int i = tid; i < NTL * 16 * KP2; i += blockDim.x) lds[i] = 0.f;
    for (int i = lane; i < strip_floats(k2) + strip_floats(n); i += 64) U[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 2 * T * fi * fi; i += blockDim.x) {
        const int c = i % fi, r = (i / fi) % fi, tj = i / (fi * fi), t = tj >> 1, j = tj & 1;
        Wl[((t * FB + (c >> 4)) * 16 + (c & 15)) * KP2 + j * 16 * FB + r] = p.W[(int64_t)(j * n + t * fi + r) * p.ldw + t * fi + c];
    }
    __syncthreads();

    const int m = lane & 15, g = lane >> 4;           // (the weight gradient's (i16, mq) are the same two numbers)
    const bool has1 = p.add1 != nullptr, has2 = p.add2 != nullptr;
    float2 pe1[NLC], pe2[NLC];
    f4 accw[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) accw[q] = f4{0.f, 0.f, 0.f, 0.f};
    int64_t out_strip = -1;
    auto store_out = [&]() {
        const int rows = (int)min((int64_t)kStrip, p.M - out_strip * kStrip);
        if (has1) bd_store_out<NLC, true>(U, p.C + out_strip * kStrip * n, n, rows, lane, pe1, pe2, has2);
        else {
            const float2 none[1] = {};
            bd_store_out<NLC, false>(U, p.C + out_strip * kStrip * n, n, rows, lane, none, none, false);
        }
    };
    for (int64_t strip = first; strip < n_strips; strip += step) {
        if (out_strip >= 0) store_out();
        store_strip<NLG>(U, pg, k2, lane);           // (rows past the batch's end were read as zeros)
        store_strip<NLX>(Xl, px, n, lane);
        load_strip<NLC>(pe1, p.add1, p.M, n, strip, lane, has1);
        load_strip<NLC>(pe2, p.add2, p.M, n, strip, lane, has2);
        if (strip + step < n_strips) {
            load_strip<NLG>(pg, p.A, p.M, k2, strip + step, lane);
            load_strip<NLX>(px, p.X, p.M, n, strip + step, lane);
        }
        // ---- input gradient: bd_backward_input's product
        f4 acc[NTL];
#pragma unroll
        for (int q = 0; q < NTL; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
        const float* xrow = U + m * k2 + 4 * g;
        f4 xn = bd_operand(xrow, fi - 4 * g), wn[FB];
#pragma unroll
        for (int q = 0; q < FB; ++q) wn[q] = *reinterpret_cast<const f4*>(Wl + (q * 16 + m) * KP2 + 4 * g);
#pragma unroll
        for (int u = 0; u < T * 2 * FB; ++u) {
            const int t = u / (2 * FB);
            const f4 xv = xn;
            f4 wv[FB];
#pragma unroll
            for (int q = 0; q < FB; ++q) wv[q] = wn[q];
            if (u + 1 < T * 2 * FB) {
                const int t1 = (u + 1) / (2 * FB), j1 = ((u + 1) / FB) % 2, b1 = (u + 1) % FB;
                xn = bd_operand(xrow + j1 * n + t1 * fi + 16 * b1, fi - 16 * b1 - 4 * g);
#pragma unroll
                for (int q = 0; q < FB; ++q) wn[q] = *reinterpret_cast<const f4*>(Wl + ((t1 * FB + q) * 16 + m) * KP2 + j1 * 16 * FB + 16 * b1 + 4 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < FB; ++q)
                    acc[t * FB + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q][s], xv[s], acc[t * FB + q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- weight gradient: bd_wgrad's product over the same two strips
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* grow = U + (4 * g + s) * k2 + m;
            const float* xr = Xl + (4 * g + s) * n + m;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float xv[FB], gv[2 * FB];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    xv[b] = xr[t * fi + 16 * b];
                    if (16 * b + m == fi) xv[b] = 1.f;                             // column f_in of the tile := 1: dW[:, f_in] = sum_m g[m, :]
                }
#pragma unroll
                for (int ja = 0; ja < 2 * FB; ++ja) gv[ja] = grow[(ja / FB) * n + t * fi + 16 * (ja % FB)];
#pragma unroll
                for (int ja = 0; ja < 2 * FB; ++ja)
#pragma unroll
                    for (int b = 0; b < FB; ++b)
                        accw[(t * 2 * FB + ja) * FB + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[ja], xv[b], accw[(t * 2 * FB + ja) * FB + b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the input-gradient rows over the (now dead) g strip
        float* c = U + m * n + 4 * g;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < FB; ++q) {
                const f4 v = acc[t * FB + q];
                const int col = 16 * q + 4 * g;
                float* d = c + t * fi + 16 * q;
                if (col < fi) *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
                if (col + 2 < fi) *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
            }
        out_strip = strip;
    }
    if (out_strip >= 0) store_out();
    // the workgroup's waves add their weight-gradient tiles up in LDS in wave order (bd_wgrad's epilogue)
    __syncthreads();
    float* red = lds;                                // [NTW][16][16]
    for (int w = 0; w < n_waves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < NTW; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = red + q * 256 + (4 * g + r) * 16 + m;
                    *d = w == 0 ? accw[q][r] : *d + accw[q][r];
                }
        }
        __syncthreads();
    }
    float* out = p.part + (int64_t)blockIdx.x * NTW * 256;
    for (int i = tid; i < NTW * 64; i += blockDim.x) reinterpret_cast<f4*>(out)[i] = reinterpret_cast<const f4*>(red)[i];
}

// dW (dense [2 Fm][lddw]: diagonal blocks = the slot sums in a fixed order, everything else zero) and dbias [2 Fm]
static __global__ __launch_bounds__(64 * kFinWaves) void bd_wgrad_finalize(int T, int fi, int FB, int slots, const float* __restrict__ part,
                                                                           float* __restrict__ dW, int64_t lddw, float* __restrict__ dbias) {
    __shared__ float red[kFinWaves][64];
    const int Fm = T * fi, kk = Fm + 1, ntl = 2 * T * FB * FB;
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const bool live = e < (int64_t)2 * Fm * kk;
    int r = 0, c = 0;
    bool diag = false;
    float s0 = 0.f, s1 = 0.f;
    if (live) {
        r = (int)(e / kk);
        c = (int)(e - (int64_t)r * kk);
        const int j = r / Fm, rr = r - j * Fm, t = rr / fi, a = rr - t * fi;
        const int b = c == Fm ? fi : c - t * fi;                                   // local column; f_in = the ones column
        diag = c == Fm || (c >= t * fi && c < (t + 1) * fi);
        if (diag) {
            const int tile = ((t * 2 + j) * FB + (a >> 4)) * FB + (b >> 4);
            const float* src = part + tile * 256 + (a & 15) * 16 + (b & 15);
            int q = sg;
            for (; q + kFinWaves < slots; q += 2 * kFinWaves) {
                s0 += src[(int64_t)q * ntl * 256];
                s1 += src[(int64_t)(q + kFinWaves) * ntl * 256];
            }
            if (q < slots) s0 += src[(int64_t)q * ntl * 256];
        }
    }
    red[sg][lane] = s0 + s1;
    __syncthreads();
    if (live && sg == 0) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kFinWaves; ++w) v += red[w][lane];
        if (c < Fm) dW[(int64_t)r * lddw + c] = diag ? v : 0.f;
        else if (dbias) dbias[r] = v;
    }
}

namespace {
int bd_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// The instantiated (towers, f_in) shapes -- f_in is a template argument so that every tower's LDS offsets are instruction immediates
// (as a runtime value they cost one address register per tower and operand: the kernels then spill).  The list: five towers (the
// reference's default, nets/dgn_layer.py:330) over the even per-tower widths up to 30; hidden 70 = 5 x 14 is BASELINE configs[1].
// Other shapes run the dense kernels (dgn_linear_forward / _wgrad).
#define DGN_BD_SHAPES(X) X(5, 14) X(5, 10) X(5, 12) X(5, 18) X(5, 20) X(5, 22) X(5, 24) X(5, 26) X(5, 28) X(5, 30) X(4, 14) X(2, 14)

int bd_blocks(int T, int fi) {                       // 16-column blocks per tower of the instantiation that takes (T, fi), 0 = none
#define DGN_CASE(TT, FF) if (T == TT && fi == FF) return (FF + 15) / 16;
    DGN_BD_SHAPES(DGN_CASE)
#undef DGN_CASE
    return 0;
}

template <typename K>
hipError_t bd_set_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
}

enum { kBdFwd = 0, kBdBwd = 1, kBdWg = 2, kBdBoth = 3 };

template <int T, int FI>
hipError_t bd_launch_one(int which, const BdParams& p, int threads, size_t lds, hipStream_t st) {
    static_assert(FI % 2 == 0 && FI % 16 != 0 && T * FI <= 16 * kMaxTiles, "even widths with room for the ones column");
    static bool attr = false;
    if (!attr) {
        hipError_t e = bd_set_lds(&bd_forward<T, FI>);
        if (e == hipSuccess) e = bd_set_lds(&bd_backward_input<T, FI>);
        if (e == hipSuccess) e = bd_set_lds(&bd_wgrad<T, FI>);
        if constexpr (FI <= 16) { if (e == hipSuccess) e = bd_set_lds(&bd_backward_both<T, FI>); }      // (two 16-column blocks per tower: 2 x the accumulators, spills)
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (which == kBdFwd) hipLaunchKernelGGL((bd_forward<T, FI>), dim3(p.groups), dim3(threads), lds, st, p);
    else if (which == kBdBwd) hipLaunchKernelGGL((bd_backward_input<T, FI>), dim3(p.groups), dim3(threads), lds, st, p);
    else if (which == kBdBoth) {
        if constexpr (FI <= 16) hipLaunchKernelGGL((bd_backward_both<T, FI>), dim3(p.groups), dim3(threads), lds, st, p);
        else return hipErrorInvalidValue;
    }
    else hipLaunchKernelGGL((bd_wgrad<T, FI>), dim3(p.groups), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t bd_launch(int which, const BdParams& p, int threads, size_t lds, hipStream_t st) {
#define DGN_CASE(TT, FF) if (p.T == TT && p.fi == FF) return bd_launch_one<TT, FF>(which, p, threads, lds, st);
    DGN_BD_SHAPES(DGN_CASE)
#undef DGN_CASE
    return hipErrorInvalidValue;
}

int bd_max_threads(int which, int T, int FB) { return which == kBdFwd ? bd_fwd_threads(T, FB) : bd_bwd_threads(T, FB); }

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

int bd_wgrad_groups(int64_t n_rows) {
    const int64_t n_strips = (n_rows + kStrip - 1) / kStrip;
    return (int)std::max<int64_t>(1, std::min<int64_t>(2 * bd_cus(), (n_strips + 3) / 4));
}

// forward / input gradient: persistent workgroups, as many waves per CU as LDS and registers allow
int bd_launch_stream(const char* fn, int which, BdParams& p, void* stream) {
    const int FB = bd_blocks(p.T, p.fi);
    const int k = p.T * p.fi;
    const size_t w_floats = which == kBdFwd ? (size_t)p.T * 2 * FB * 16 * (16 * FB + 4) + (size_t)p.T * 2 * FB * 16 : (size_t)p.T * FB * 16 * (32 * FB + 4);
    const size_t wave_floats = which == kBdFwd ? (size_t)kStrip * 2 * k : (size_t)strip_floats(2 * k);
    // waves per workgroup: the choice that puts the most waves on a CU (LDS: the tiles once per workgroup + a strip per wave)
    const int max_waves = bd_max_threads(which, p.T, FB) / 64;
    int waves = 0, per_cu = 0;
    for (int w = max_waves; w >= 1; --w) {
        const size_t bytes = (w_floats + w * wave_floats) * 4;
        if (bytes > (size_t)kLdsBudget) continue;
        const int pc = std::min((int)(kLdsBudget / bytes), 16 / w);
        if (pc * w > per_cu * waves) { waves = w; per_cu = pc; }
    }
    if (!waves) { set_error("%s: operands do not fit in LDS", fn); return -1; }
    const size_t lds = (w_floats + waves * wave_floats) * 4;
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    p.groups = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)bd_cus() * per_cu, (n_strips + waves - 1) / waves));
    DGN_HIP_CHECK(bd_launch(which, p, waves * 64, lds, static_cast<hipStream_t>(stream)));
    return 0;
}
}  // namespace
}  // namespace lin
}  // namespace dgn

using namespace dgn;
using namespace dgn::lin;

extern "C" int dgn_linear_bd_supported(int32_t n_towers, int32_t f_in) { return bd_blocks(n_towers, f_in) != 0; }

extern "C" int dgn_linear_bd_forward(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* a, const float* w, int64_t ldw,
                                     const float* bias, float* c, void* stream) {
    const char* fn = "dgn_linear_bd_forward";
    if (n_rows < 0 || !bd_blocks(n_towers, f_in)) { set_error("%s: unsupported towers x width (%d x %d)", fn, n_towers, f_in); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!a || !w || !c || !al16(a) || !al16(c) || ldw < (int64_t)n_towers * f_in) { set_error("%s: null / misaligned operand (16-byte aligned dense rows)", fn); return DGN_ERR_INVALID; }
    BdParams p{};
    p.M = n_rows; p.T = n_towers; p.fi = f_in; p.A = a; p.W = w; p.ldw = ldw; p.bias = bias; p.C = c;
    return bd_launch_stream(fn, kBdFwd, p, stream);
}

extern "C" int dgn_linear_bd_backward_input(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* g, const float* w, int64_t ldw,
                                            const float* add1, const float* add2, float* c, void* stream) {
    const char* fn = "dgn_linear_bd_backward_input";
    if (n_rows < 0 || !bd_blocks(n_towers, f_in)) { set_error("%s: unsupported towers x width (%d x %d)", fn, n_towers, f_in); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!g || !w || !c || !al16(g) || !al16(c) || (add1 && !al16(add1)) || (add2 && !al16(add2)) || (add2 && !add1) || ldw < (int64_t)n_towers * f_in) {
        set_error("%s: null / misaligned operand (16-byte aligned dense rows; add2 needs add1)", fn);
        return DGN_ERR_INVALID;
    }
    BdParams p{};
    p.M = n_rows; p.T = n_towers; p.fi = f_in; p.A = g; p.W = w; p.ldw = ldw; p.C = c; p.add1 = add1; p.add2 = add2;
    return bd_launch_stream(fn, kBdBwd, p, stream);
}

extern "C" size_t dgn_linear_bd_wgrad_workspace_bytes(int64_t n_rows, int32_t n_towers, int32_t f_in) {
    const int FB = bd_blocks(n_towers, f_in);
    if (n_rows <= 0 || !FB) return 0;
    return (size_t)bd_wgrad_groups(n_rows) * 2 * n_towers * FB * FB * 256 * sizeof(float);
}

extern "C" int dgn_linear_bd_wgrad(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* g, const float* x, float* dw, int64_t lddw,
                                   float* dbias, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "dgn_linear_bd_wgrad";
    const int FB = bd_blocks(n_towers, f_in);
    if (n_rows < 0 || !FB) { set_error("%s: unsupported towers x width (%d x %d)", fn, n_towers, f_in); return DGN_ERR_INVALID; }
    const int Fm = n_towers * f_in;
    if (!dw || lddw < Fm) { set_error("%s: null output", fn); return DGN_ERR_INVALID; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_rows == 0) {
        if (zero_rows_async(dw, 2 * Fm, Fm, lddw, st)) return DGN_ERR_HIP;
        if (dbias && zero_rows_async(dbias, 1, 2 * Fm, 2 * Fm, st)) return DGN_ERR_HIP;
        return DGN_OK;
    }
    if (!g || !x || !al16(g) || !al16(x)) { set_error("%s: null / misaligned operand (16-byte aligned dense rows)", fn); return DGN_ERR_INVALID; }
    const size_t need = dgn_linear_bd_wgrad_workspace_bytes(n_rows, n_towers, f_in);
    if (!ws || ws_bytes < need) { set_error("%s: workspace too small (%zu < %zu)", fn, ws_bytes, need); return DGN_ERR_WORKSPACE; }
    BdParams p{};
    p.M = n_rows; p.T = n_towers; p.fi = f_in; p.A = g; p.X = x; p.part = static_cast<float*>(ws);
    p.groups = bd_wgrad_groups(n_rows);
    const int ntl = 2 * n_towers * FB * FB;
    const size_t lds = std::max((size_t)4 * (strip_floats(2 * Fm) + strip_floats(Fm)), (size_t)ntl * 256) * 4;
    DGN_HIP_CHECK(bd_launch(kBdWg, p, 256, lds, st));
    const int64_t total = (int64_t)2 * Fm * (Fm + 1);
    hipLaunchKernelGGL(bd_wgrad_finalize, dim3((unsigned)((total + 63) / 64)), dim3(64 * kFinWaves), 0, st, n_towers, f_in, FB, p.groups, p.part, dw, lddw,
                       dbias);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}


// Input gradient + weight gradient (+ bias gradient) of the block-diagonal product in one pass (bd_backward_both): the arguments of
// dgn_linear_bd_backward_input and dgn_linear_bd_wgrad together, the workspace of dgn_linear_bd_wgrad_workspace_bytes().  Library-internal
// (dgn_towers_layer_backward, option bd_bwd_fused); returns DGN_ERR_INVALID where the pair of calls has to be used instead.
namespace dgn {
namespace lin {
int bd_backward_both_launch(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* g, const float* w, int64_t ldw, const float* x,
                            const float* add1, const float* add2, float* g_h, float* dw, int64_t lddw, float* dbias, void* ws, size_t ws_bytes,
                            void* stream) {
    const char* fn = "bd_backward_both";
    const int FB = bd_blocks(n_towers, f_in);
    const int Fm = n_towers * f_in;
    if (n_rows <= 0 || FB != 1 || !g || !w || !x || !g_h || !dw || !al16(g) || !al16(x) || !al16(g_h) || (add1 && !al16(add1)) || (add2 && !al16(add2)) ||
        (add2 && !add1) || ldw < Fm || lddw < Fm) {
        set_error("%s: unsupported shape or operand", fn);
        return DGN_ERR_INVALID;
    }
    const size_t need = dgn_linear_bd_wgrad_workspace_bytes(n_rows, n_towers, f_in);
    if (!ws || ws_bytes < need) { set_error("%s: workspace too small (%zu < %zu)", fn, ws_bytes, need); return DGN_ERR_WORKSPACE; }
    const int ntw = 2 * n_towers * FB * FB;
    const size_t w_floats = (size_t)n_towers * FB * 16 * (32 * FB + 4), wave_floats = (size_t)strip_floats(2 * Fm) + strip_floats(Fm);
    int waves = 8;
    while (waves > 1 && (w_floats + waves * wave_floats) * 4 > (size_t)kLdsBudget) waves /= 2;
    const size_t lds = std::max((w_floats + waves * wave_floats) * 4, (size_t)ntw * 256 * 4);
    if (lds > (size_t)kLdsBudget) { set_error("%s: operands do not fit in LDS", fn); return DGN_ERR_INVALID; }
    const int64_t n_strips = (n_rows + kStrip - 1) / kStrip;
    BdParams p{};
    p.M = n_rows; p.T = n_towers; p.fi = f_in; p.A = g; p.X = x; p.W = w; p.ldw = ldw; p.C = g_h; p.add1 = add1; p.add2 = add2;
    p.part = static_cast<float*>(ws);
    p.groups = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(bd_cus(), bd_wgrad_groups(n_rows)), (n_strips + waves - 1) / waves));
    hipStream_t st = static_cast<hipStream_t>(stream);
    DGN_HIP_CHECK(bd_launch(kBdBoth, p, waves * 64, lds, st));
    const int64_t total = (int64_t)2 * Fm * (Fm + 1);
    hipLaunchKernelGGL(bd_wgrad_finalize, dim3((unsigned)((total + 63) / 64)), dim3(64 * kFinWaves), 0, st, n_towers, f_in, FB, p.groups, p.part, dw, lddw,
                       dbias);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
}  // namespace lin
}  // namespace dgn
