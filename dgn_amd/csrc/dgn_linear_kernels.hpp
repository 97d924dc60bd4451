// Tall-skinny fp32 Linear for the layer's pre/post-aggregation transforms (see include/dgn_hip.h, dgn_linear_*):
//   forward / input gradient   C[t] = A[t] . W[t]^T (+ bias)     A: [M, k] with M = number of nodes, k, n <= 160
//   weight gradient            dW[t] = G[t]^T . X[t]             reduction over the M rows
// The reference does these with nn.Linear inside its pretrans/posttrans MLPs (layers.py:101-112 via
// nets/dgn_layer.py:67-75,116-119).  On ZINC-sized batches M is ~3e5 while k and n are 42..140: a library GEMM
// spends its time in per-tile prologues of a 5-iteration K loop, and picks a different (often 3x slower) kernel for
// every new M.  Here the whole weight matrix sits in LDS for the life of a persistent workgroup and the rows stream
// through: every wave owns 16-row strips, copies a strip global -> registers -> its private LDS slice with
// contiguous 16-byte lanes (the next strip's loads are in flight while the current one is multiplied), and feeds
// v_mfma_f32_16x16x4_f32 (exact fp32) from LDS.  The product is formed transposed (D[n][m]) so that a lane ends up
// with four consecutive output columns of one row: 16-byte stores.  The reduction index is visited in the order
// k = 16b + 4*(lane/16) + s, which lets a lane fetch its four operands of a 16-k block with one ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "dgn_common.hpp"

namespace dgn {
namespace lin {

using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int kStrip = 16;          // rows per strip = one MFMA tile edge
constexpr int kMaxTiles = 10;       // k, n <= 160
constexpr int kLdsBudget = 160 * 1024;
constexpr int kFacFloats = 2 * kStrip * 4;
constexpr int kMaxWgradTiles = 45;  // accumulator tiles (4 registers each) one wave can hold for the weight gradient

__host__ __device__ inline int lds_stride(int k) { return ((k + 15) / 16) * 16 + 4; }   // == 4 mod 8: spreads rows over banks

// The per-row factors of a strip (scale table / graph norm) travel with the strip's prefetch as RAW loaded values: replacing an absent
// factor by 1 (`p.sc ? f : 1.f`) is done where the factors are consumed, one iteration later.  Anything that consumes a just-issued
// load makes the wave wait, at that point, for every load issued before it -- the next strip's prefetch included (loads retire in
// order) -- i.e. the prefetch became a synchronous round trip in front of the strip's MFMAs (seen in the ISA of the round-2 kernels).
struct ExpandSrc {
    const float* gy; int64_t sT;                     // [T][M][fo] dense rows, sT between towers
    const float* sc;                                 // [M][S] or null
    int S, fo;
};

struct LinParams {
    int64_t M;
    int k, n, T, kp;
    const float* A; int64_t sA;                      // dense rows: [M][k]
    const float* W; int64_t ldw, sW; int w_kn;      // 0: W[n][k] (C = A W^T)   1: W[k][n] (C = A W)
    const float* bias; int64_t sBias;
    float* C; int64_t sC;                            // dense rows: [M][n]
    int groups;                                      // workgroups per batch entry
    ExpandSrc ex;                                    // kExpand: A is formed from ex (A, sA unused)
    // scale-combine epilogue (S > 0): y[m][t*fo + o] = rs[m] * (cb[t*fo + o] + sum_s sc[m][s] * c[t][m][s*fo + o]); C is not written
    int S, fo;
    const float* sc; const float* rs; const float* cb;
    float* Y; int64_t ldy;
    // kBnPlain: the operand is BatchNorm(A) formed while the strip is staged: ((a - mean[c]) * invstd[c]) * gamma[c] + beta[c], the
    // arithmetic of bn_apply (dgn_bn_tail.hip) in the same order; gamma / beta may be NULL (1 / 0)
    const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;
    // kActPlain: the operand is a * act'(z + bias[c]) -- the gradient through bias + activation (dgn_bias_act_backward's arithmetic) --
    // formed while the strip is staged from a and z (both [M, k] dense); gz_out (may be NULL) receives the formed operand
    const float* act_z; const float* act_bias; int act_kind; float act_slope; float* gz_out;
    // kActMask: the activation's derivative comes from a byte mask instead of z: byte i2 describes the float2 number i2 of the dense [M, k]
    // tensor, bit 0 / bit 1 = (z + bias > 0) of its first / second element (written by kMixFwd with zmask_out set, which then does
    // not write the pre-activation C at all: 1/8 of its bytes)
    const unsigned char* act_mask; unsigned char* zmask_out;
    // kAddPlain: C = (add1 + A op(W)) + add2 -- two more [M, n] dense operands added in the epilogue (add2 may be NULL), the sum of
    // gradient contributions that otherwise costs its own pass
    const float* add1; const float* add2;
    // kMixFwd: kBnPlain's prologue AND an epilogue that also delivers out2 = act(C + ep_bias[c]) (+ add1: the residual), C itself being
    // written as well (the pre-activation the backward needs): BatchNorm -> Linear -> bias + LeakyReLU + residual in one pass
    const float* ep_bias; float* out2;
    // kActMaskBnb (round 6; the towers layer's mixing-network input gradient): kActMask's operand, and an epilogue that turns the product's
    // rows g_y1 [M, n] straight into the gradient at posttrans' output -- BatchNorm's backward and the graph norm,
    //     g_yr[m][c] = rs[m] * (gamma[c] * invstd[c] * (g_y1[m][c] - sums[c] / M - xhat[m][c] * sums[n + c] / M)),  xhat = (y[m][c] - mean[c]) * invstd[c]
    // (dgn_combine.hip combine_bwd's arithmetic in its order) -- written TOWER-MAJOR: gz[t][m][o], c = t * fo + o, sT floats between
    // towers.  bnb_y [M, n] dense is BatchNorm's input, bn_mean / bn_invstd / bn_gamma its tables (width n here), bnb_sums [2 n] the column
    // sums (sum g_y1, sum g_y1 xhat); `rs` (may be NULL) and `fo` as in the combine epilogue.  g_y1 itself is never written.
    const float* bnb_y; const float* bnb_sums; float* bnb_gz; int64_t bnb_sT;
    int wreg;                                        // launch the WREG instance (set by launch_linear: shape, mode and option lin_wreg)
    // combine epilogue, round 6: BatchNorm's training statistics of the output ride in the pass -- every lane adds the y values it stores (and
    // their squares) to fp64 cells of its own in LDS (ds_add_f64: no registers, nothing waits), the workgroup folds them per column at its
    // end and leaves bn_part[(q * bn_F + t * fo + o) * groups + grp], q = 0 (sum y) / 1 (sum y^2): what bn_stats would have computed
    // in a pass of its own over y, in the layout bn_finalize reads.  st_off: where the cells start in LDS (floats; set by launch_linear)
    double* bn_part; int bn_F; int st_off;
};

// registers a lane needs: accumulators + one block of W operands + the prefetched strip
constexpr int linear_extra_regs(int KB, int mode) { return mode == 3 ? 12 : (mode == 4 ? 4 * KB + 12 : (mode == 7 ? 2 * KB + 28 : (mode == 8 ? 2 * KB + 28 + 64 : (mode == 5 || mode == 6 ? 64 : 0)))); }    // kBnPlain: column state; kActPlain: a second prefetched strip
constexpr int linear_threads(int NT, int KB, int mode = 0) {
    const int est = 8 * NT + 4 * KB + 52 + linear_extra_regs(KB, mode);
    return est <= 116 ? 1024 : (mode == 6 && est <= 180 && NT <= 5 ? 768 : 512);      // (kMixFwd at hidden 65 .. 80: 157 registers, three waves per SIMD)
}
__host__ __device__ inline int strip_floats(int k) { return kStrip * k + 16; }     // + slack read by the last row's last block

// A strip is 16 * k consecutive floats of A (rows are dense: lda == k) starting at a multiple of 64 bytes: it is copied with
// 16-byte lanes, float4 number q = jq * 64 + lane living in pre[2 jq], pre[2 jq + 1] (NL = 2 KB is even); the batch's last,
// partial strip has the same layout, the lanes past its rows hold zeros.
__device__ __forceinline__ int strip_idx2(int j, int lane) { return 2 * ((j >> 1) * 64 + lane) + (j & 1); }   // float2 number held by pre[j]
__device__ __forceinline__ int tail_idx2(int j, int lane) { return strip_idx2(j, lane); }                      // (the epilogue registers pe1[j] / pe2[j] likewise)
// Round 6: the strip is read with BUFFER loads (one V# per strip: base = the strip's first row, num_records = its valid bytes) -- lanes past
// the strip's end, and past the tensor's end on the batch's last, partial strip, read zeros by the range check (checked per dword), so the
// loop has ONE load shape.  With global loads the full strip's 16-byte lanes and the partial strip's clamped 8-byte lanes were two branches
// that the compiler merged into dword + dword + dwordx2 per lane: three memory instructions for one, on every strip of every kernel here.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<const float*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t strip_rsrc(const float* base, int bytes) {      // (wave-uniform arguments)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
// The S <= 3 scale factors of row row0 + (lane & 15) in ONE 12-byte buffer load (they were three dword loads): the V# covers the strip's rows of
// the [M][S] table, so for S < 3 the extra components are the next row's factors (never used) and past the strip / the table they read
// zero.  sc == NULL: zeros (not used either).
typedef float f3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ f3v load_row_scales(const float* sc, int S, int64_t M, int64_t row0, int lane) {
    const int rows = __builtin_amdgcn_readfirstlane((int)min((int64_t)kStrip, M - row0));
    const __amdgpu_buffer_rsrc_t rs = strip_rsrc(uniform_ptr(sc + row0 * S), sc ? rows * S * 4 : 0);
    return __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(rs, (lane & 15) * S * 4, 0, 0));
}
template <int NL>
__device__ __forceinline__ void load_strip(float2 (&pre)[NL], const float* A, int64_t M, int k, int64_t strip, int lane, bool present = true) {      // (!present: an absent operand, all lanes read zeros)
    const int64_t row0 = strip * kStrip;
    const float* base = uniform_ptr(A + row0 * k);
    const int rows = __builtin_amdgcn_readfirstlane((int)min((int64_t)kStrip, M - row0));
    const __amdgpu_buffer_rsrc_t rs = strip_rsrc(base, present ? rows * k * 4 : 0);
#pragma unroll
    for (int jq = 0; jq < NL / 2; ++jq) {
        const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (jq * 64 + lane) * 16, 0, 0));
        pre[2 * jq] = make_float2(v[0], v[1]);
        pre[2 * jq + 1] = make_float2(v[2], v[3]);
    }
}
template <int NL>
__device__ __forceinline__ void store_strip(float* Xl, const float2 (&pre)[NL], int k, int lane) {
#pragma unroll
    for (int jq = 0; jq < NL / 2; ++jq)
        if (jq * 64 + lane < (kStrip / 4) * k)
            reinterpret_cast<float4*>(Xl)[jq * 64 + lane] = make_float4(pre[2 * jq].x, pre[2 * jq].y, pre[2 * jq + 1].x, pre[2 * jq + 1].y);
}


// "Expanded" strips: the operand is not in memory but G[t][m][s*fo + o] = scale[m][s] * gy[t][m][o] -- the gradient of the
// posttrans product behind the scale-combine -- formed while the strip is staged.  gy is tower-major, so a strip's 16 * fo
// floats are one contiguous run; they and the rows' scale factors travel as the prefetch.  NL covers the expanded width:
// loads past the run repeat its last element (one cache line).
template <int NL>
__device__ __forceinline__ void load_expand(float2 (&pre)[NL], f4& fac, const ExpandSrc& e, int t, int64_t M, int64_t strip, int lane) {
    // the tower's 16 * fo contiguous floats as 16-byte buffer lanes: float4 number jq * 64 + lane in pre[2 jq], pre[2 jq + 1] (S >= 2: the run is at
    // most half the expanded width, NLE / 2 = ceil(NL / 4) lanes-of-64 cover it); rows past the batch's end read zeros
    constexpr int NLE = 2 * ((NL + 3) / 4);
    float2 run[NLE];
    load_strip<NLE>(run, e.gy + t * e.sT, M, e.fo, strip, lane);
#pragma unroll
    for (int j = 0; j < NLE; ++j) pre[j] = run[j];
    const f3v s3 = load_row_scales(e.sc, e.S, M, strip * kStrip, lane);       // (without a table the values are dummies: store_expand puts 1)
    fac = f4{s3[0], s3[1], s3[2], 0.f};
}
// Xl: [16][S*fo]; Fl: 16 x f4 scratch of this wave.  Rows >= rows_valid become zero.
template <int NL>
__device__ __forceinline__ void store_expand(float* Xl, float* Fl, const float2 (&pre)[NL], const f4& fac, const ExpandSrc& e,
                                             int rows_valid, int lane) {
    if (lane < 16) *reinterpret_cast<f4*>(Fl + 4 * lane) = e.sc ? fac : f4{1.f, 1.f, 1.f, 0.f};
    const int fo2 = e.fo >> 1, width = e.S * e.fo;
    constexpr int NLE = 2 * ((NL + 3) / 4);
#pragma unroll
    for (int j = 0; j < NLE; ++j) {
        const int idx = strip_idx2(j, lane), r = idx / fo2, o2 = idx - r * fo2;
        if (idx < kStrip * fo2) {
            const f4 f = *reinterpret_cast<const f4*>(Fl + 4 * r);
            const float2 v = r < rows_valid ? pre[j] : make_float2(0.f, 0.f);
            float* d = Xl + r * width + 2 * o2;
            *reinterpret_cast<float2*>(d) = make_float2(v.x * f[0], v.y * f[0]);
            if (e.S > 1) *reinterpret_cast<float2*>(d + e.fo) = make_float2(v.x * f[1], v.y * f[1]);
            if (e.S > 2) *reinterpret_cast<float2*>(d + 2 * e.fo) = make_float2(v.x * f[2], v.y * f[2]);
        }
    }
}

// store_strip with the BatchNorm transform of the operand (tables in LDS: Bn[0..3][kp16] = mean, invstd, gamma, beta).  pre[j] holds
// float2 number strip_idx2(j, lane) of the strip, i.e. columns (2 * idx2) % k and the next one (k is even: a pair never straddles a row)
template <int NL>
__device__ __forceinline__ void store_strip_bn(float* Xl, const float2 (&pre)[NL], int k, int lane, const float* Bn, int kp16) {
    auto tf = [&](float v, int c) { return (v - Bn[c]) * Bn[kp16 + c] * Bn[2 * kp16 + c] + Bn[3 * kp16 + c]; };
    int c = (4 * lane) % k;                                  // column of the lane's first element in piece jq = 0
    const int dc = 256 % k;
#pragma unroll
    for (int jq = 0; jq < NL / 2; ++jq) {
        if (jq * 64 + lane < (kStrip / 4) * k) {
            const int c2 = c + 2 >= k ? c + 2 - k : c + 2;
            reinterpret_cast<float4*>(Xl)[jq * 64 + lane] = make_float4(tf(pre[2 * jq].x, c), tf(pre[2 * jq].y, c + 1), tf(pre[2 * jq + 1].x, c2), tf(pre[2 * jq + 1].y, c2 + 1));
        }
        c += dc;
        if (c >= k) c -= k;
    }
}

// store_strip for kActPlain: v = a * act'(z + bias[c]) (act: 1 ReLU, 2 LeakyReLU(slope), else identity), also written to gz (the strip's
// place in the side output, NULL = not wanted; n2 = float2's of the strip that exist)
__device__ __forceinline__ float act_grad_lin(float v, int act, float slope) {
    if (act == 1) return v > 0.f ? 1.f : 0.f;
    if (act == 2) return v > 0.f ? 1.f : slope;
    return 1.f;
}
template <int NL>
__device__ __forceinline__ void store_strip_act(float* Xl, const float2 (&pre)[NL], const float2 (&prez)[NL], int k, int lane, const float* Ba,
                                                int act, float slope, float* gz, int n2) {
    auto tf = [&](float a, float z, int c) { return a * act_grad_lin(z + Ba[c], act, slope); };
    int c = (4 * lane) % k;
    const int dc = 256 % k;
#pragma unroll
    for (int jq = 0; jq < NL / 2; ++jq) {
        const int q = jq * 64 + lane;
        if (q < (kStrip / 4) * k) {
            const int c2 = c + 2 >= k ? c + 2 - k : c + 2;
            const float4 v = make_float4(tf(pre[2 * jq].x, prez[2 * jq].x, c), tf(pre[2 * jq].y, prez[2 * jq].y, c + 1),
                                         tf(pre[2 * jq + 1].x, prez[2 * jq + 1].x, c2), tf(pre[2 * jq + 1].y, prez[2 * jq + 1].y, c2 + 1));
            reinterpret_cast<float4*>(Xl)[q] = v;
            if (gz) {
                if (2 * q + 1 < n2) reinterpret_cast<float4*>(gz)[q] = v;
                else if (2 * q < n2) reinterpret_cast<float2*>(gz)[2 * q] = make_float2(v.x, v.y);
            }
        }
        c += dc;
        if (c >= k) c -= k;
    }
}

// kActMask: the mask bytes of a strip travel as the prefetch's second part: mz[jq] = byte of pre[2 jq] | byte of pre[2 jq + 1] << 8
template <int NL>
__device__ __forceinline__ void load_mask(unsigned (&mz)[NL / 2], const unsigned char* mask, int64_t M, int k, int64_t strip, int lane) {
    const int64_t row0 = strip * kStrip;
    const int n2 = (int)min((int64_t)kStrip, M - row0) * (k >> 1);               // float2's (= mask bytes) of the strip that exist
    const unsigned char* base = mask + strip * (kStrip / 2) * k;                 // wave-uniform; 8 k bytes per strip
    if (n2 == kStrip * (k >> 1)) {
        const int last4 = (kStrip / 4) * k - 1;
#pragma unroll
        for (int jq = 0; jq < NL / 2; ++jq) mz[jq] = reinterpret_cast<const unsigned short*>(base)[min(jq * 64 + lane, last4)];
    } else {
#pragma unroll
        for (int jq = 0; jq < NL / 2; ++jq)
            mz[jq] = (unsigned)base[min(strip_idx2(2 * jq, lane), n2 - 1)] | ((unsigned)base[min(strip_idx2(2 * jq + 1, lane), n2 - 1)] << 8);
    }
}
// store_strip_act with the derivative read from the mask (act_grad_lin's values: 1 where z + bias > 0, else slope / 0)
template <int NL>
__device__ __forceinline__ void store_strip_mask(float* Xl, const float2 (&pre)[NL], const unsigned (&mz)[NL / 2], int k, int lane, int act, float slope,
                                                 float* gz, int n2) {
    const float off = act == 2 ? slope : (act == 1 ? 0.f : 1.f);
    auto tf = [&](float a, unsigned bit) { return a * (bit ? 1.f : off); };
#pragma unroll
    for (int jq = 0; jq < NL / 2; ++jq) {
        const int q = jq * 64 + lane;
        if (q < (kStrip / 4) * k) {
            const unsigned m = mz[jq];
            const float4 v = make_float4(tf(pre[2 * jq].x, m & 1u), tf(pre[2 * jq].y, m & 2u), tf(pre[2 * jq + 1].x, m & 0x100u), tf(pre[2 * jq + 1].y, m & 0x200u));
            reinterpret_cast<float4*>(Xl)[q] = v;
            if (gz) {
                if (2 * q + 1 < n2) reinterpret_cast<float4*>(gz)[q] = v;
                else if (2 * q < n2) reinterpret_cast<float2*>(gz)[2 * q] = make_float2(v.x, v.y);
            }
        }
    }
}

enum { kPlain = 0, kCombine = 1, kExpand = 2, kBnPlain = 3, kActPlain = 4, kAddPlain = 5, kMixFwd = 6, kActMask = 7, kActMaskBnb = 8 };        // ts_linear variants

// WREG (round 6): the lane's W operands of ALL 16-k blocks stay in registers (KB x NT f4) instead of being re-read from LDS for every strip --
// the matrix pipe fed from LDS runs at 95 / 108 / 148 TFLOP/s with one ds_read_b128 per 4 / 8 / 16 MFMAs (profiles/r03_mfma_peak.txt), and
// the strip products read NT + 1 operands per 4 NT MFMAs (posttrans of the towers: one per 3).  With WREG: one per 4 NT.  Costs the
// registers of 4 NT KB floats: 512 threads per workgroup (256 registers per lane), shapes up to kWregTiles tiles.
constexpr int kWregTiles = 18;
// (12 .. 18 tiles; the shapes below need more than the 168 registers of three waves per SIMD -- the build refuses scratch)
constexpr bool linear_wreg_ok(int NT, int KB, int MODE) {
    if (!(MODE == 1 || MODE == 2) || NT * KB > kWregTiles || NT * KB < 12) return false;
    if (MODE == 1) return !(NT == 2 && KB == 9);
    return !((NT == 2 && KB >= 8) || (NT == 3 && KB == 6) || (NT == 9 && KB == 2));
}
template <int NT, int KB, int MODE, bool WREG = false>
__global__ __launch_bounds__(WREG ? 256 : linear_threads(NT, KB, MODE), WREG ? 3 : 1) void ts_linear(LinParams p) {
    constexpr bool COMBINE = MODE == kCombine, EXPAND = MODE == kExpand, MIX = MODE == kMixFwd, BNP = MODE == kBnPlain || MIX, ACT = MODE == kActPlain;
    constexpr bool BNB = MODE == kActMaskBnb, ADD = MODE == kAddPlain || MIX || BNB;      // (BNB: the epilogue's strip-shaped operand is BatchNorm's input)
    constexpr bool ACTM = MODE == kActMask || BNB;
    extern __shared__ float lds[];
    constexpr int NL = 2 * KB;                       // float2 loads per lane and strip: 16 * (k/2) / 64 <= 2 * KB
    constexpr int NLC = 2 * NT;                      // the same for a strip of C
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int t = blockIdx.x % p.T, grp = blockIdx.x / p.T;
    const int kp = p.kp, k = p.k, n = p.n;
    float* Bl = lds;                                 // [NT*16] bias
    float* Cb = Bl + NT * 16;                        // [NT*16] the combine epilogue's bias for this tower
    float* Bn = Cb + NT * 16;                        // kBnPlain: [4][KB*16] mean, invstd, gamma, beta of the operand's columns
    constexpr int kTabFloats = 2 * NT * 16 + (BNP ? 4 * KB * 16 : (ACT ? KB * 16 : (BNB ? 5 * NT * 16 : 0)));
    float* Wl = lds + kTabFloats;                    // [NT*16][kp], zero beyond (n, k)
    // (WREG: the weights are dead in LDS once every lane holds its operands -- the waves' strips lie over them)
    float* Xl = (WREG ? Wl : Wl + NT * 16 * kp) + wave * (strip_floats(k) + kStrip * n + kFacFloats);   // this wave's strip, as it lies in memory
    float* Cl = Xl + strip_floats(k);                // results of the previous strip, [16][n]
    float* Fl = Cl + kStrip * n;                     // [2][16][4] per-row factors of the combine epilogue (scale_0..2, row_scale)
    // (LinParams.bn_part: this wave's statistics cells, [4 per epilogue trip][64 lanes] doubles -- each lane's own, the wave's LDS operations run in order)
    const int st_cells = COMBINE ? 4 * ((kStrip * (p.fo >> 1) + 63) >> 6) * 64 : 0;
    double* St = reinterpret_cast<double*>(lds + p.st_off) + wave * st_cells;
    if constexpr (COMBINE) {
        if (p.bn_part) for (int i = lane; i < st_cells; i += 64) St[i] = 0.0;
    }

    const float* A = p.A + (int64_t)t * p.sA;
    float* C = p.C + (int64_t)t * p.sC;
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    const int64_t first = (int64_t)grp * n_waves + wave, step = (int64_t)p.groups * n_waves;
    float2 pre[NL];
    f4 fac = f4{1.f, 1.f, 1.f, 1.f};                 // the combine epilogue's row factors travel with the strip's prefetch
    // (branch-free: a conditional load would make the compiler wait for everything in flight where the branches join)
    const int S1 = COMBINE ? p.S : 1;
    auto load_fac = [&](int64_t strip) {
        if constexpr (BNB) {                          // the row's graph-norm factor alone
            const int64_t row = min(strip * kStrip + (lane & 15), p.M - 1);
            fac[3] = *(p.rs ? p.rs + row : A);
            return;
        }
        if constexpr (!COMBINE) return;
        const int64_t row = min(strip * kStrip + (lane & 15), p.M - 1);
        const f3v s3 = load_row_scales(p.sc, S1, p.M, strip * kStrip, lane);          // (absent factors: dummy loads, replaced by 1 where `fac` is consumed)
        const float* rsp = p.rs ? p.rs + row : A;
        fac = f4{s3[0], s3[1], s3[2], *rsp};
    };
    float2 prez[ACT ? NL : 1];
    unsigned mz[ACTM ? NL / 2 : 1];
    auto fetch = [&](int64_t strip) {
        if constexpr (EXPAND) load_expand<NL>(pre, fac, p.ex, t, p.M, strip, lane);
        else { load_strip<NL>(pre, A, p.M, k, strip, lane); load_fac(strip); }
        if constexpr (ACT) load_strip<NL>(prez, p.act_z, p.M, k, strip, lane);
        if constexpr (ACTM) load_mask<NL>(mz, p.act_mask, p.M, k, strip, lane);
    };
    if (first < n_strips) fetch(first);              // in flight while the weights are set up

    for (int i = tid; i < NT * 16 * kp + kTabFloats; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const float* Wg = p.W + (int64_t)t * p.sW;
    for (int i = tid; i < n * k; i += blockDim.x) {
        int r, c;                                    // r: output column, c: reduction index
        if (p.w_kn) { c = i / n; r = i - c * n; } else { r = i / k; c = i - r * k; }
        Wl[r * kp + c] = p.w_kn ? Wg[(int64_t)c * p.ldw + r] : Wg[(int64_t)r * p.ldw + c];
    }
    if (p.bias) for (int i = tid; i < n; i += blockDim.x) Bl[i] = p.bias[(int64_t)t * p.sBias + i];
    if (COMBINE && p.cb) for (int i = tid; i < p.fo; i += blockDim.x) Cb[i] = p.cb[t * p.fo + i];
    if constexpr (ACT) {
        for (int i = tid; i < KB * 16; i += blockDim.x) Bn[i] = (i < k && p.act_bias) ? p.act_bias[i] : 0.f;
    }
    if constexpr (MIX) {
        for (int i = tid; i < NT * 16; i += blockDim.x) Cb[i] = (i < n && p.ep_bias) ? p.ep_bias[i] : 0.f;
    }
    if constexpr (BNB) {      // [mean | invstd | gamma * invstd | sum g / M | sum g xhat / M] of the OUTPUT's columns
        const float inv_n = 1.f / (float)p.M;
        for (int i = tid; i < NT * 16; i += blockDim.x) {
            const bool in = i < n;
            const float is = in ? p.bn_invstd[i] : 0.f;
            Bn[i] = in ? p.bn_mean[i] : 0.f;
            Bn[NT * 16 + i] = is;
            Bn[2 * NT * 16 + i] = (in && p.bn_gamma) ? p.bn_gamma[i] : 1.f;
            Bn[3 * NT * 16 + i] = in ? p.bnb_sums[i] * inv_n : 0.f;
            Bn[4 * NT * 16 + i] = in ? p.bnb_sums[n + i] * inv_n : 0.f;
        }
    }
    if constexpr (BNP) {
        for (int i = tid; i < KB * 16; i += blockDim.x) {
            const bool in = i < k;
            Bn[i] = in ? p.bn_mean[i] : 0.f;
            Bn[KB * 16 + i] = in ? p.bn_invstd[i] : 0.f;
            Bn[2 * KB * 16 + i] = (in && p.bn_gamma) ? p.bn_gamma[i] : 1.f;
            Bn[3 * KB * 16 + i] = (in && p.bn_beta) ? p.bn_beta[i] : 0.f;
        }
    }
    __syncthreads();

    const int m = lane & 15, g = lane >> 4;
    const float* xrow = Xl + m * k + 4 * g;
    const float* wrow = Wl + m * kp + 4 * g;
    const bool k4 = (k & 3) == 0;
    f4 wreg[WREG ? KB : 1][WREG ? NT : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int q = 0; q < NT; ++q) wreg[b][q] = *reinterpret_cast<const f4*>(wrow + 16 * q * kp + 16 * b);
        __syncthreads();                             // (the strips overwrite the weights)
    }
    // lane (m, g) holds C[row0 + m][16q + 4g .. + 3].  A strip's results are stored one iteration late, right before the
    // loads of the strip after next are issued: loads and stores retire through one in-order counter, so a wave that
    // stored at the end of an iteration would sit out the store acknowledgement (~2.5 us) before touching its prefetch.
    // They wait in LDS, from where the strip's 16 * n floats leave as one contiguous run (C's rows are dense)
    // (8-byte pieces with gaps, straight from the accumulators, ran at half the store rate).
    int64_t out_strip = -1;
    int it = 0, out_it = 0;
    float2 pe1[ADD ? NLC : 1], pe2[ADD ? NLC : 1];
    const int mix_c4 = (4 * lane) % n, mix_d4 = 256 % n;       // kMixFwd: column of the lane's float4 number lane (+ 64 jq) of a result strip
    // kActMaskBnb: the epilogue walks a result strip TOWER-MAJOR, as the output lies in memory: float2 number i2 = j * 64 + lane is pair
    // (i2 % fo2) of row (i2 / fo2) % 16 of tower i2 / (8 fo) -- a store instruction covers 512 contiguous bytes of a tower's [N][fo] plane
    // instead of nine 56-byte pieces of it (the strip's own row-major order).  bnb_at[j] = row | column << 4 | tower << 12 | column inside
    // the tower << 16, the same for every strip; BatchNorm's input rows are loaded in the same order (8-byte lanes of the strip's V#).
    unsigned bnb_at[BNB ? NLC : 1];
    if constexpr (BNB) {
        const int fo2 = p.fo >> 1, per_t = kStrip * fo2;
#pragma unroll
        for (int j = 0; j < NLC; ++j) {
            const int i2 = min(j * 64 + lane, kStrip * (n >> 1) - 1);
            const int t = i2 / per_t, rem = i2 - t * per_t, r = rem / fo2, o = 2 * (rem - r * fo2), cc = t * p.fo + o;
            bnb_at[j] = (unsigned)(r & 15) | ((unsigned)cc << 4) | ((unsigned)t << 12) | ((unsigned)o << 16);
        }
    }
    auto store_out = [&]() {
        if constexpr (BNB) {
            const int64_t row0 = out_strip * kStrip;
            const int rows_valid = (int)min((int64_t)kStrip, p.M - row0);
            const float* F = Fl + (out_it & 1) * (kStrip * 4);
            constexpr int T16 = NT * 16;
#pragma unroll
            for (int j = 0; j < NLC; ++j) {
                const unsigned at = bnb_at[j];
                const int r = at & 15, cc = (at >> 4) & 255, t = (at >> 12) & 15, o = at >> 16;
                const float2 c = *reinterpret_cast<const float2*>(Cl + r * n + cc);
                const float2 yv = pe1[j];
                const float rs = F[4 * r + 3];
                float out[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ce = cc + e;
                    const float is = Bn[T16 + ce], ga = Bn[2 * T16 + ce];
                    const float xh = ((e ? yv.y : yv.x) - Bn[ce]) * is;
                    float g = e ? c.y : c.x;
                    g = ga * is * (g - Bn[3 * T16 + ce] - xh * Bn[4 * T16 + ce]);
                    if (p.rs) g *= rs;
                    out[e] = g;
                }
                if (j * 64 + lane < kStrip * (n >> 1) && r < rows_valid)
                    *reinterpret_cast<float2*>(p.bnb_gz + (int64_t)t * p.bnb_sT + (row0 + r) * p.fo + o) = make_float2(out[0], out[1]);
            }
            return;
        }
        if constexpr (COMBINE) {
            const int64_t row0 = out_strip * kStrip;
            const int fo2 = p.fo >> 1;                // (f_out is even: pairs of output columns, 8-byte stores)
            const int cnt = (int)min((int64_t)kStrip, p.M - row0) * fo2;
            const float* F = Fl + (out_it & 1) * (kStrip * 4);
            double* st = St + lane;
            for (int idx = lane; idx < cnt; idx += 64, st += 256) {
                const int r = idx / fo2, o = 2 * (idx - r * fo2);
                const f4 f = *reinterpret_cast<const f4*>(F + 4 * r);
                float2 v = *reinterpret_cast<const float2*>(Cb + o);
                for (int s = 0; s < p.S; ++s) {
                    const float2 z = *reinterpret_cast<const float2*>(Cl + r * n + s * p.fo + o);
                    v.x += f[s] * z.x;
                    v.y += f[s] * z.y;
                }
                const float y0 = v.x * f[3], y1 = v.y * f[3];
                *reinterpret_cast<float2*>(p.Y + (row0 + r) * p.ldy + t * p.fo + o) = make_float2(y0, y1);
                if (p.bn_part) {                      // (uniform; bn_stats' per-element values: y and the fp32 square, added in fp64)
                    lds_add_f64(st, (double)y0);
                    lds_add_f64(st + 64, (double)y1);
                    lds_add_f64(st + 128, (double)(y0 * y0));
                    lds_add_f64(st + 192, (double)(y1 * y1));
                }
            }
            return;
        }
        const int cnt2 = (int)min((int64_t)kStrip, p.M - out_strip * kStrip) * (n >> 1);
        float* dst = C + out_strip * kStrip * n;
        if (cnt2 == kStrip * (n >> 1)) {             // a full strip: 16 n floats from a 64-byte aligned address, 16-byte lanes
#pragma unroll
            for (int jq = 0; jq < NLC / 2; ++jq) {
                if (jq * 64 + lane < (kStrip / 4) * n) {
                    float4 c = reinterpret_cast<const float4*>(Cl)[jq * 64 + lane];
                    if constexpr (MIX) {
                        const int cc = (mix_c4 + jq * mix_d4) % n, c2 = cc + 2 >= n ? cc + 2 - n : cc + 2;
                        auto af = [&](float v) { return p.act_kind == 1 ? fmaxf(v, 0.f) : (p.act_kind == 2 ? (v > 0.f ? v : v * p.act_slope) : v); };
                        const float v0 = c.x + Cb[cc], v1 = c.y + Cb[cc + 1], v2 = c.z + Cb[c2], v3 = c.w + Cb[c2 + 1];
                        float4 o = make_float4(af(v0), af(v1), af(v2), af(v3));
                        if (p.add1) o = make_float4(o.x + pe1[2 * jq].x, o.y + pe1[2 * jq].y, o.z + pe1[2 * jq + 1].x, o.w + pe1[2 * jq + 1].y);
                        reinterpret_cast<float4*>(p.out2 + out_strip * kStrip * n)[jq * 64 + lane] = o;
                        if (p.zmask_out) {          // the backward's act'(z + b) as two mask bytes; the pre-activation itself is not written
                            const unsigned m = (v0 > 0.f ? 1u : 0u) | (v1 > 0.f ? 2u : 0u) | (v2 > 0.f ? 0x100u : 0u) | (v3 > 0.f ? 0x200u : 0u);
                            reinterpret_cast<unsigned short*>(p.zmask_out + out_strip * (kStrip / 2) * n)[jq * 64 + lane] = (unsigned short)m;
                            continue;
                        }
                    } else if constexpr (ADD) {
                        c = make_float4(pe1[2 * jq].x + c.x, pe1[2 * jq].y + c.y, pe1[2 * jq + 1].x + c.z, pe1[2 * jq + 1].y + c.w);
                        if (p.add2) c = make_float4(c.x + pe2[2 * jq].x, c.y + pe2[2 * jq].y, c.z + pe2[2 * jq + 1].x, c.w + pe2[2 * jq + 1].y);
                    }
                    if constexpr (EXPAND) st_stream(reinterpret_cast<float4*>(dst) + (jq * 64 + lane), c);        // (the [N, A F] gradient: read by the NEXT kernel, too large to stay cached)
                    else reinterpret_cast<float4*>(dst)[jq * 64 + lane] = c;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NLC; ++j) {
                const int i2 = tail_idx2(j, lane);
                if (i2 < cnt2) {
                    float2 c = reinterpret_cast<const float2*>(Cl)[i2];
                    if constexpr (MIX) {
                        const int cc = (2 * (i2)) % n;
                        auto af = [&](float v) { return p.act_kind == 1 ? fmaxf(v, 0.f) : (p.act_kind == 2 ? (v > 0.f ? v : v * p.act_slope) : v); };
                        const float v0 = c.x + Cb[cc], v1 = c.y + Cb[cc + 1];
                        float2 o = make_float2(af(v0), af(v1));
                        if (p.add1) o = make_float2(o.x + pe1[j].x, o.y + pe1[j].y);
                        reinterpret_cast<float2*>(p.out2 + out_strip * kStrip * n)[i2] = o;
                        if (p.zmask_out) {
                            p.zmask_out[out_strip * (kStrip / 2) * n + i2] = (unsigned char)((v0 > 0.f ? 1u : 0u) | (v1 > 0.f ? 2u : 0u));
                            continue;
                        }
                    } else if constexpr (ADD) {
                        c = make_float2(pe1[j].x + c.x, pe1[j].y + c.y);
                        if (p.add2) c = make_float2(c.x + pe2[j].x, c.y + pe2[j].y);
                    }
                    if constexpr (EXPAND) st_stream(reinterpret_cast<float2*>(dst) + (i2), c);
                    else reinterpret_cast<float2*>(dst)[i2] = c;
                }
            }
        }
    };
    // kAddPlain: the epilogue's two extra operands of strip `s_` in store_out's indexing (loaded one iteration ahead of their use)
    auto load_adds = [&](int64_t s_) {
        if constexpr (BNB) {                          // BatchNorm's input rows of strip s_, in the epilogue's tower-major order
            const int64_t row0 = s_ * kStrip;
            const int rows = __builtin_amdgcn_readfirstlane((int)min((int64_t)kStrip, p.M - row0));
            const __amdgpu_buffer_rsrc_t ry = strip_rsrc(uniform_ptr(p.bnb_y + row0 * n), rows * n * 4);
#pragma unroll
            for (int j = 0; j < NLC; ++j) {
                const unsigned at = bnb_at[j];
                const unsigned long long v = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(ry, (int)(((at & 15) * n + ((at >> 4) & 255)) * 4), 0, 0));
                pe1[j] = make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
            }
            return;
        }
        if constexpr (ADD) {
            load_strip<NLC>(pe1, p.add1, p.M, n, s_, lane, p.add1 != nullptr);
            load_strip<NLC>(pe2, p.add2, p.M, n, s_, lane, p.add2 != nullptr);
        }
    };
    for (int64_t strip = first; strip < n_strips; strip += step) {
        if constexpr (EXPAND) store_expand<NL>(Xl, Fl, pre, fac, p.ex, kStrip, lane);
        else if constexpr (BNP) store_strip_bn<NL>(Xl, pre, k, lane, Bn, KB * 16);
        else if constexpr (ACT) store_strip_act<NL>(Xl, pre, prez, k, lane, Bn, p.act_kind, p.act_slope,
                                                    p.gz_out ? p.gz_out + strip * kStrip * k : nullptr, (int)min((int64_t)kStrip, p.M - strip * kStrip) * (k >> 1));
        else if constexpr (ACTM) store_strip_mask<NL>(Xl, pre, mz, k, lane, p.act_kind, p.act_slope,
                                                      p.gz_out ? p.gz_out + strip * kStrip * k : nullptr, (int)min((int64_t)kStrip, p.M - strip * kStrip) * (k >> 1));
        else store_strip<NL>(Xl, pre, k, lane);
        if (COMBINE && lane < 16)
            *reinterpret_cast<f4*>(Fl + (it & 1) * (kStrip * 4) + 4 * lane) = f4{p.sc ? fac[0] : 1.f, p.sc ? fac[1] : 1.f, p.sc ? fac[2] : 1.f, p.rs ? fac[3] : 1.f};
        if (BNB && lane < 16) Fl[(it & 1) * (kStrip * 4) + 4 * lane + 3] = p.rs ? fac[3] : 1.f;
#ifndef DGN_EXP_NO_STORE
        if (out_strip >= 0) store_out();
#endif
        load_adds(strip);
#ifndef DGN_EXP_NO_FETCH
        if (strip + step < n_strips) fetch(strip + step);
#endif

        f4 acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = *reinterpret_cast<const f4*>(Bl + 16 * q + 4 * g);
        auto load_x = [&](int b) {                   // 16-k blocks: lane group g takes k = 16b + 4g + s in the s-th MFMA
            f4 xv;
            if (k4) {
                xv = *reinterpret_cast<const f4*>(xrow + 16 * b);
            } else {
                const float2 lo = *reinterpret_cast<const float2*>(xrow + 16 * b), hi = *reinterpret_cast<const float2*>(xrow + 16 * b + 2);
                xv = f4{lo.x, lo.y, hi.x, hi.y};
            }
            if (b == KB - 1) {                       // columns past k belong to the next row: W is zero there, but 0 * inf is not
#pragma unroll
                for (int s = 0; s < 4; ++s) xv[s] = 16 * b + 4 * g + s < k ? xv[s] : 0.f;
            }
            return xv;
        };
#ifdef DGN_EXP_MFMA_QUARTER
        constexpr int SN = 1;                        // (what-if ablation: a quarter of the MFMA work, wrong results)
#else
        constexpr int SN = 4;
#endif
        if constexpr (WREG) {
            f4 xq[KB];
#pragma unroll
            for (int b = 0; b < KB; ++b) xq[b] = load_x(b);
#pragma unroll
            for (int b = 0; b < KB; ++b)
#pragma unroll
                for (int s = 0; s < SN; ++s)
#pragma unroll
                    for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[b][q][s], xq[b][s], acc[q], 0, 0, 0);
        } else {
#pragma unroll 1
            for (int b = 0; b < KB; ++b) {
                const f4 xv = load_x(b);
                f4 wv[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) wv[q] = *reinterpret_cast<const f4*>(wrow + 16 * q * kp + 16 * b);
#pragma unroll
                for (int s = 0; s < SN; ++s)         // s outer: consecutive MFMAs on one accumulator would wait for each other
#pragma unroll
                    for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q][s], xv[s], acc[q], 0, 0, 0);
            }
        }
        float* c = Cl + m * n + 4 * g;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = 16 * q + 4 * g;
            if (col + 1 < n) *reinterpret_cast<float2*>(c + 16 * q) = make_float2(acc[q][0], acc[q][1]);
            if (col + 3 < n) *reinterpret_cast<float2*>(c + 16 * q + 2) = make_float2(acc[q][2], acc[q][3]);
        }
        out_strip = strip;
        out_it = it++;
    }
    if (out_strip >= 0) store_out();
    if constexpr (COMBINE) {
        if (p.bn_part) {                              // (uniform) this workgroup's column partials: waves, then the 16 row slots of a column, in a fixed order
            __syncthreads();
            const int fo2 = p.fo >> 1;
            if (tid < 2 * p.fo) {
                const int q = tid / p.fo, c = tid - q * p.fo, c2 = c >> 1;
                double a = 0.0;
                for (int w = 0; w < n_waves; ++w) {
                    const double* sw = reinterpret_cast<const double*>(lds + p.st_off) + w * st_cells;
                    for (int r = 0; r < kStrip; ++r) {
                        const int idx = r * fo2 + c2;
                        a += sw[(4 * (idx >> 6) + 2 * q + (c & 1)) * 64 + (idx & 63)];
                    }
                }
                p.bn_part[((int64_t)q * p.bn_F + t * p.fo + c) * p.groups + grp] = a;
            }
        }
    }
}

// ---- weight gradient -------------------------------------------------------------------------------------------
struct WgParams {
    int64_t M;
    int n, k, T;
    const float* G; int64_t sG;                      // dense rows: [M][n]
    const float* X; int64_t sX;                      // [M][k]
    float* part;                                     // [T][slots][NT*16][KT*16]
    int groups;
    int ones;                                        // append a column of ones to X (k % 16 != 0)
    // X := BatchNorm(X) formed while the strip is staged (bn_apply's arithmetic, see LinParams); tables live bn_off floats into the LDS
    const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta; int bn_off;
    ExpandSrc ex;                                    // EXPAND: G is formed from ex (G, sG unused)
    // GMASK (round 6): G := G * act'(.) with the derivative read from kMixFwd's byte mask (LinParams.act_mask: one byte per float2 of the
    // dense [M, n] tensor) while the strip is staged -- the mixing network's weight gradient straight from the layer's output gradient,
    // the masked tensor g_z is never written or re-read
    const unsigned char* g_mask; int act_kind; float act_slope;
};

// One wave = one partial sum of the whole [n, k] gradient over its strips (NT x KT accumulator tiles); four waves
// per workgroup, one per SIMD.  MFMA: D[n][k] += G[m][n] * X[m][k] with the strip's rows as the reduction index
// (m = 4*(lane/16) + s for the s-th instruction).  Columns past n / k of a strip row alias the next row: they only
// reach accumulator entries that are never read.
// GMASK: the strip's mask bytes (8 n per strip, contiguous) travel as ONE direct-to-LDS load per 64 pieces of 16 bytes
// (global_load_lds_dwordx4: no registers -- five prefetched mask words per lane put the 70 x 70 shape at 262 registers, one wave per SIMD
// instead of two, 104.7 us against the plain kernel's 65.8 on ZINC-12k), double-buffered per wave, read back when the strip is staged.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
constexpr int wgrad_wave_floats(int n, int k, bool gmask) { return strip_floats(n) + strip_floats(k) + 64 + (gmask ? 4 * n : 0); }
// (GMASK at up to 25 accumulator tiles: bounded to the two waves per SIMD the plain kernel reaches on its own -- 254 registers there)
constexpr int wgrad_min_blocks(int NT, int KT, bool GMASK) { return GMASK && NT == 5 && KT == 5 ? 2 : 1; }      // (hidden 65 .. 80: the shipped widths; 3 x 8 / 8 x 3 would spill)
template <int NT, int KT, bool EXPAND, bool GMASK = false>
__global__ __launch_bounds__(256, wgrad_min_blocks(NT, KT, GMASK)) void ts_wgrad(WgParams p) {
    static_assert(!(EXPAND && GMASK), "one G prologue at a time");
    extern __shared__ float lds[];
    constexpr int NLG = 2 * NT, NLX = 2 * KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int t = blockIdx.x % p.T, grp = blockIdx.x / p.T;
    const int n = p.n, k = p.k;
    float* Gl = lds + wave * wgrad_wave_floats(n, k, GMASK);
    float* Xl = Gl + strip_floats(n);
    float* Fl = Xl + strip_floats(k);                // 16 x f4 scale factors (EXPAND)
    float* Ml = Fl + 64;                             // GMASK: two mask buffers of 8 n bytes (2 n floats) each
    for (int i = lane; i < strip_floats(n) + strip_floats(k); i += 64) Gl[i] = 0.f;
    const float* BnW = nullptr;
    if (!EXPAND && p.bn_mean) {                      // (uniform)
        float* tab = lds + p.bn_off;
        for (int i = tid; i < KT * 16; i += blockDim.x) {
            const bool in = i < k;
            tab[i] = in ? p.bn_mean[i] : 0.f;
            tab[KT * 16 + i] = in ? p.bn_invstd[i] : 0.f;
            tab[2 * KT * 16 + i] = (in && p.bn_gamma) ? p.bn_gamma[i] : 1.f;
            tab[3 * KT * 16 + i] = (in && p.bn_beta) ? p.bn_beta[i] : 0.f;
        }
        __syncthreads();
        BnW = tab;
    }

    const float* G = p.G + (int64_t)t * p.sG;
    const float* X = p.X + (int64_t)t * p.sX;
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    const int64_t first = (int64_t)grp * n_waves + wave, step = (int64_t)p.groups * n_waves;

    float2 pg[NLG], px[NLX];
    f4 fac = f4{1.f, 1.f, 1.f, 0.f};
    int buf = 0;                                     // GMASK: the mask buffer the NEXT fetch lands in
    auto fetch_g = [&](int64_t strip) {
        if constexpr (EXPAND) load_expand<NLG>(pg, fac, p.ex, t, p.M, strip, lane);
        else load_strip<NLG>(pg, G, p.M, n, strip, lane);
        if constexpr (GMASK) {
            // 16-byte pieces of the strip's mask; the last strip's pieces past its rows repeat its last one (the allocation ends there)
            const int rows_s = (int)min((int64_t)kStrip, p.M - strip * kStrip);
            const int pieces = (rows_s * (n >> 1) + 15) >> 4;
            const unsigned char* mb = p.g_mask + strip * (kStrip / 2) * n;
#pragma unroll
            for (int q = 0; q < (NT * 16 / 2 + 63) / 64; ++q) {
                const int pi = q * 64 + lane;
                if (pi < (n >> 1))
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(mb + 16 * min(pi, pieces - 1)), (lds_void_t*)(Ml + buf * 2 * n + q * 256), 16, 0, 0);
            }
            buf ^= 1;
        }
    };
    f4 acc[NT][KT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int i16 = lane & 15, mq = lane >> 4;
    if (first < n_strips) {
        fetch_g(first);
        load_strip<NLX>(px, X, p.M, k, first, lane);
    }
    for (int64_t strip = first; strip < n_strips; strip += step) {
        const int rows = (int)min((int64_t)kStrip, p.M - strip * kStrip);
        if constexpr (EXPAND) {
            store_expand<NLG>(Gl, Fl, pg, fac, p.ex, rows, lane);
        } else {
            if (rows < kStrip) {                     // rows past the end contribute zero
#pragma unroll
                for (int j = 0; j < NLG; ++j)
                    if (strip_idx2(j, lane) >= rows * (n >> 1)) pg[j] = make_float2(0.f, 0.f);
            }
            if constexpr (GMASK) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the direct-to-LDS mask of this strip has landed: nothing else is in flight here)
                const unsigned short* mw = reinterpret_cast<const unsigned short*>(Ml + (buf ^ 1) * 2 * n);      // (buf names the NEXT fetch's buffer: this strip's is the other one)
                unsigned gmz[NLG / 2];
#pragma unroll
                for (int jq = 0; jq < NLG / 2; ++jq) gmz[jq] = mw[min(jq * 64 + lane, 4 * n - 1)];
                store_strip_mask<NLG>(Gl, pg, gmz, n, lane, p.act_kind, p.act_slope, nullptr, 0);
            } else {
                store_strip<NLG>(Gl, pg, n, lane);
            }
        }
        if (BnW) store_strip_bn<NLX>(Xl, px, k, lane, BnW, KT * 16);
        else store_strip<NLX>(Xl, px, k, lane);
        if (strip + step < n_strips) {
            fetch_g(strip + step);
            load_strip<NLX>(px, X, p.M, k, strip + step, lane);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float gv[NT], xv[KT];
#pragma unroll
            for (int a = 0; a < NT; ++a) gv[a] = Gl[(4 * mq + s) * n + 16 * a + i16];
#pragma unroll
            for (int b = 0; b < KT; ++b) xv[b] = Xl[(4 * mq + s) * k + 16 * b + i16];
            if (p.ones && 16 * (KT - 1) + i16 == k) xv[KT - 1] = 1.f;      // column k of X := 1, so dW[:, k] = sum_m G[m, :] (the bias gradient)
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < KT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[a], xv[b], acc[a][b], 0, 0, 0);
        }
    }
    // lane holds D[n = 16a + 4*mq + r][k = 16b + i16]: the workgroup's waves add up in LDS in wave order, one slot
    // per workgroup goes to memory
    __syncthreads();
    float* red = lds;                                // [NT*16][KT*16], reuses the strips
    for (int w = 0; w < n_waves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < KT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = red + (16 * a + 4 * mq + r) * (KT * 16) + 16 * b + i16;
                        *d = w == 0 ? acc[a][b][r] : *d + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    float* out = p.part + ((int64_t)t * p.groups + grp) * (NT * 16) * (KT * 16);
    for (int i = tid; i < NT * 16 * KT * 16 / 4; i += blockDim.x) reinterpret_cast<f4*>(out)[i] = reinterpret_cast<const f4*>(red)[i];
}

// dW[t][n][k] = sum over the workgroup slots in a fixed order (bitwise reproducible): a block covers 64 consecutive
// elements, its sixteen waves take every sixteenth slot (all loads of a lane in flight together), LDS joins them
constexpr int kFinWaves = 16;
static __global__ __launch_bounds__(64 * kFinWaves) void ts_wgrad_finalize(int T, int n, int k, int slots, int npad, int kpad,
                                                         const float* __restrict__ part, float* __restrict__ dW,
                                                         int64_t lddw, int64_t sdW, float* __restrict__ dbias, int64_t sdb,
                                                         float* __restrict__ pick = nullptr, int pick_off = 0, int pick_w = 0) {
    // (pick: rows [pick_off, pick_off + pick_w) of every batch entry's bias gradient also go, densely, to pick[t * pick_w + ...]: the identity
    //  scaler's block of the expanded gradient's column sums IS the posttrans bias gradient -- no launch of its own)
    __shared__ float red[kFinWaves][64];
    const int kk = dbias ? k + 1 : k;                // with the ones column: k + 1 columns per row, the last one is the bias gradient
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const bool live = e < (int64_t)T * n * kk;
    int t = 0, r = 0, c = 0;
    float s0 = 0.f, s1 = 0.f;
    if (live) {
        t = (int)(e / ((int64_t)n * kk));
        const int rem = (int)(e - (int64_t)t * n * kk);
        r = rem / kk;
        c = rem - r * kk;
        const float* src = part + (int64_t)t * slots * npad * kpad + (int64_t)r * kpad + c;
        int q = sg;
        for (; q + kFinWaves < slots; q += 2 * kFinWaves) {
            s0 += src[(int64_t)q * npad * kpad];
            s1 += src[(int64_t)(q + kFinWaves) * npad * kpad];
        }
        if (q < slots) s0 += src[(int64_t)q * npad * kpad];
    }
    red[sg][lane] = s0 + s1;
    __syncthreads();
    if (live && sg == 0) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kFinWaves; ++w) v += red[w][lane];
        if (c < k) dW[(int64_t)t * sdW + (int64_t)r * lddw + c] = v;
        else {
            dbias[(int64_t)t * sdb + r] = v;
            if (pick && r >= pick_off && r < pick_off + pick_w) pick[t * pick_w + r - pick_off] = v;
        }
    }
}

// ---- dispatch -------------------------------------------------------------------------------------------------
// ---- dispatch: one translation unit per kernel family (dgn_linear*.hip), each instantiating its (NT, KB) grid --------
// kActPlain holds two prefetched strips: the widest tile shapes would not fit the registers and are not instantiated
constexpr bool linear_act_shape_ok(int NT, int KB) { return 4 * NT + 8 * KB <= 104; }
constexpr bool linear_add_shape_ok(int NT, int KB) { return 16 * NT + 4 * KB <= 128; }      // (kAddPlain: two result-shaped strips per wave in registers)
constexpr bool linear_bnb_shape_ok(int NT, int KB) { return NT <= 5 && KB <= 7 && NT + KB <= 11; }              // (kActMaskBnb: the unrolled epilogue; wider shapes spill)

template <int NT, int KB, int MODE>
hipError_t launch_linear_nkm(const LinParams& p, int threads, size_t lds, hipStream_t st) {
    if constexpr (((MODE == kActPlain || MODE == kActMask || MODE == kActMaskBnb) && !linear_act_shape_ok(NT, KB)) ||
                  ((MODE == kAddPlain || MODE == kMixFwd || MODE == kActMaskBnb) && !linear_add_shape_ok(NT, KB)) ||
                  (MODE == kActMaskBnb && !linear_bnb_shape_ok(NT, KB))) {
        return hipErrorInvalidValue;
    } else {
    if constexpr (linear_wreg_ok(NT, KB, MODE)) {
        if (p.wreg) {
            static bool attr_w = false;
            if (!attr_w) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ts_linear<NT, KB, MODE, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
                if (e != hipSuccess) return e;
                attr_w = true;
            }
            hipLaunchKernelGGL((ts_linear<NT, KB, MODE, true>), dim3(p.T * p.groups), dim3(threads), lds, st, p);
            return hipGetLastError();
        }
    }
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ts_linear<NT, KB, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL((ts_linear<NT, KB, MODE>), dim3(p.T * p.groups), dim3(threads), lds, st, p);
    return hipGetLastError();
    }
}
template <int NT, int KT, bool EXPAND, bool GMASK = false>
hipError_t launch_wgrad_nke(const WgParams& p, size_t lds, hipStream_t st) {
    if constexpr (NT * KT > kMaxWgradTiles) {
        return hipErrorInvalidValue;
    } else {
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ts_wgrad<NT, KT, EXPAND, GMASK>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL((ts_wgrad<NT, KT, EXPAND, GMASK>), dim3(p.T * p.groups), dim3(256), lds, st, p);
        return hipGetLastError();
    }
}

#define DGN_LIN_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)

template <int NT, int MODE>
hipError_t launch_linear_n(int kb, const LinParams& p, int threads, size_t lds, hipStream_t st) {
    switch (kb) {
#define DGN_CASE(K) case K: return launch_linear_nkm<NT, K, MODE>(p, threads, lds, st);
        DGN_LIN_CASES(DGN_CASE)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}
template <int MODE>
hipError_t launch_linear_grid(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st) {
    switch (nt) {
#define DGN_CASE(N) case N: return launch_linear_n<N, MODE>(kb, p, threads, lds, st);
        DGN_LIN_CASES(DGN_CASE)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}
template <int NT, bool EXPAND, bool GMASK = false>
hipError_t launch_wgrad_n(int kt, const WgParams& p, size_t lds, hipStream_t st) {
    switch (kt) {
#define DGN_CASE(K) case K: return launch_wgrad_nke<NT, K, EXPAND, GMASK>(p, lds, st);
        DGN_LIN_CASES(DGN_CASE)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}
template <bool EXPAND, bool GMASK = false>
hipError_t launch_wgrad_grid(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st) {
    switch (nt) {
#define DGN_CASE(N) case N: return launch_wgrad_n<N, EXPAND, GMASK>(kt, p, lds, st);
        DGN_LIN_CASES(DGN_CASE)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}

// defined in dgn_linear.hip (plain), dgn_linear_combine.hip, dgn_linear_expand.hip, dgn_linear_wgrad.hip
hipError_t launch_linear_plain(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_combine(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_expand(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_actm(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_bn(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_act(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_add(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_mix(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_linear_bnb(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st);
hipError_t launch_wgrad_plain(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st);
hipError_t launch_wgrad_expand(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st);
hipError_t launch_wgrad_gmask(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st);

}  // namespace lin
}  // namespace dgn
