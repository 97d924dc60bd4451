// Scale-combine epilogue of the post-aggregation Linear (see include/dgn_hip.h, dgn_scale_combine_*): streaming
// kernels between the tower-major GEMM output z / g_z [T][N][S*fo] and the node-major y / g_y [N][T*fo].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dgn_common.hpp"

namespace dgn {
namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;          // rows in flight per thread (forward)
constexpr int kMaxGroups = 2048;    // workgroups (= bias-gradient partial slots) of the backward
constexpr int kTileFloats = 8192;   // 32 KB g_y tile budget per workgroup (backward)
constexpr int kMaxWidth = 4096;     // widest y row accepted (the backward keeps a row tile + a bias row in LDS)

// forward slab height: 64 rows for large inputs, down to 8 when that would leave most of the 256 CUs without a
// workgroup (a 3 000-row molecule batch)
int fwd_slab_rows(int64_t n) {
    int r = 8;
    while (r < 64 && (n + r - 1) / r > 2048) r *= 2;
    return r;
}

// Forward: a workgroup owns a slab of rows, a thread one output column (and a row phase when the row is narrower
// than the workgroup).  kUnroll rows are loaded (row indices clamped) before any use and only the stores are
// predicated: with one 4-byte load in flight per thread the kernel ran at 1.9 TB/s, with S * kUnroll at 4.3 TB/s.
template <int S_>
__global__ __launch_bounds__(kThreads) void combine_fwd(int64_t n_nodes, int rows_per_block, int T, int S_rt, int fo,
                                                        const float* __restrict__ z, const float* __restrict__ scale,
                                                        const float* __restrict__ bias, const float* __restrict__ row_scale,
                                                        float* __restrict__ y, int64_t ld_y, double* __restrict__ bn_part) {
    // bn_part (round 6; width <= kThreads): BatchNorm's training statistics of y ride in the pass -- a thread's column is fixed, it adds the values it
    // stores (and their fp32 squares) in fp64, the workgroup folds its row phases and leaves bn_part[(q * width + c) * gridDim.x + blockIdx.x]
    // (q = 0 sum, 1 sum of squares): bn_stats' partials in bn_finalize's layout, without bn_stats' pass over y
    __shared__ double red[2][kThreads];
    double st0 = 0.0, st1 = 0.0;
    const int S = S_ ? S_ : S_rt;
    const int width = T * fo;
    const int P = max(1, kThreads / width);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, n_nodes);
    for (int c0 = 0; c0 < width; c0 += kThreads) {
        const int p = (int)threadIdx.x / width, c = c0 + (int)threadIdx.x % width;
        if (p >= P || c >= width) continue;
        const int t = c / fo, o = c - t * fo;
        const float b = bias ? bias[c] : 0.f;
        const float* zt = z + (int64_t)t * n_nodes * ((int64_t)S * fo) + o;
        for (int64_t n = r0 + p; n < r1; n += (int64_t)kUnroll * P) {
            float acc[kUnroll], rs[kUnroll];
            if constexpr (S_ != 0) {
                float zv[kUnroll][S_ ? S_ : 1], sv[kUnroll][S_ ? S_ : 1];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int64_t m = min(n + (int64_t)u * P, r1 - 1);
#pragma unroll
                    for (int q = 0; q < S_; ++q) {
                        zv[u][q] = zt[m * ((int64_t)S_ * fo) + q * fo];
                        sv[u][q] = scale ? scale[m * S_ + q] : 1.f;
                    }
                    rs[u] = row_scale ? row_scale[m] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    acc[u] = b;
#pragma unroll
                    for (int q = 0; q < S_; ++q) acc[u] += sv[u][q] * zv[u][q];
                }
            } else {
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int64_t m = min(n + (int64_t)u * P, r1 - 1);
                    acc[u] = b;
                    for (int q = 0; q < S; ++q) acc[u] += scale[m * S + q] * zt[m * ((int64_t)S * fo) + q * fo];
                    rs[u] = row_scale ? row_scale[m] : 1.f;
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t m = n + (int64_t)u * P;
                if (m < r1) {
                    const float v = acc[u] * rs[u];
                    y[m * ld_y + c] = v;
                    st0 += (double)v;
                    st1 += (double)(v * v);
                }
            }
        }
    }
    if (bn_part) {                                    // (uniform; width <= kThreads: one trip of the column loop, thread = (row phase, column))
        red[0][threadIdx.x] = st0;
        red[1][threadIdx.x] = st1;
        __syncthreads();
        const int c = (int)threadIdx.x;
        if (c < width) {
            for (int q = 1; q < P; ++q) { st0 += red[0][c + q * width]; st1 += red[1][c + q * width]; }
            bn_part[(int64_t)c * gridDim.x + blockIdx.x] = st0;
            bn_part[((int64_t)width + c) * gridDim.x + blockIdx.x] = st1;
        }
    }
}

// (row, column) cursor over a [rows, w] tile walked linearly with a stride of kThreads elements
struct Cursor {
    int r, c, dr, dc, w;
    __device__ Cursor(int start, int w_) : r(start / w_), c(start - (start / w_) * w_), dr(kThreads / w_), dc(kThreads - (kThreads / w_) * w_), w(w_) {}
    __device__ void next() {
        r += dr;
        c += dc;
        if (c >= w) { c -= w; ++r; }
    }
};

// backward slab height: what the tile budget allows, at most 32 rows
int bwd_slab_rows(int wy) { return std::max(1, std::min(32, kTileFloats / wy)); }

// Backward: g_z[t][n][s*fo+o] = row_scale[n] * scale[n,s] * g_y[n, t*fo+o].  A workgroup walks slabs b, b + G, ...;
// each slab of row_scale * g_y is staged in LDS so that the g_z stores of one tower are one contiguous run (a
// thread-per-column layout wrote 168-byte pieces).  The bias gradient sum_n row_scale[n] * g_y[n, :] is kept per
// workgroup in LDS and written to the workgroup's own slot of `bias_part` [T*fo][G]; bias_finalize adds the slots.
// No atomics: 4 300 workgroups adding into 70 addresses serialised at ~22 ns per add (measured 170 us of a 220 us
// kernel), and the fixed slot order makes the gradient reproducible.
__global__ __launch_bounds__(kThreads) void combine_bwd(int64_t n_nodes, int rows_per_block, int T, int S, int fo,
                                                        const float* __restrict__ gy, int64_t ld_gy,
                                                        const float* __restrict__ scale, const float* __restrict__ row_scale,
                                                        float* __restrict__ gz, float* __restrict__ bias_part, const DgnBnGrad bn,
                                                        int has_bn) {
    extern __shared__ float lds[];
    const int zw = S * fo, wy = T * fo, G = (int)gridDim.x;
    float* g_t = lds;                                // [rows][wy]   row_scale * g_y
    float* s_t = g_t + (size_t)rows_per_block * wy;  // [rows][S]
    float* b_t = s_t + (size_t)rows_per_block * S;   // [wy]         bias-gradient partial of this workgroup
    float* c_t = b_t + wy;                           // [6][wy]      BatchNorm-backward column constants (fused form)
    // rows >= n_valid (DgnBnGrad.n_valid: padding rows of a batch held at a fixed capacity) took no part in the batch statistics and
    // get a zero gradient
    const int64_t n_valid = (has_bn && bn.n_valid) ? *bn.n_valid : n_nodes;
    for (int c = threadIdx.x; c < wy; c += kThreads) {
        b_t[c] = 0.f;
        if (has_bn) {
            const float inv_n = 1.f / (float)n_valid;
            c_t[c] = bn.mean[c];
            c_t[wy + c] = bn.invstd[c];
            c_t[2 * wy + c] = bn.gamma ? bn.gamma[c] : 1.f;
            c_t[3 * wy + c] = bn.beta ? bn.beta[c] : 0.f;
            c_t[4 * wy + c] = bn.sums[c] * inv_n;
            c_t[5 * wy + c] = bn.sums[wy + c] * inv_n;
        }
    }
    const int64_t n_slabs = (n_nodes + rows_per_block - 1) / rows_per_block;
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += G) {
        const int64_t r0 = slab * rows_per_block;
        const int rows = (int)min((int64_t)rows_per_block, n_nodes - r0);
        __syncthreads();                             // previous slab fully consumed
        const bool pairs = has_bn && (wy & 1) == 0 && (bn.ld & 1) == 0 &&
                           ((reinterpret_cast<uintptr_t>(bn.y) | reinterpret_cast<uintptr_t>(bn.g_out)) & 7) == 0;
        if (pairs) {
            // two columns per thread (8-byte loads of the BatchNorm input and of the tail's gradient; the same arithmetic per element)
            const int wy2 = wy >> 1;
            Cursor k((int)threadIdx.x, wy2);
#pragma unroll 4
            for (int i = threadIdx.x; i < rows * wy2; i += kThreads, k.next()) {
                const int c = 2 * k.c;
                const int64_t off = (r0 + k.r) * bn.ld + c;
                const float2 yv = *reinterpret_cast<const float2*>(bn.y + off), gv = *reinterpret_cast<const float2*>(bn.g_out + off);
                const float rs = row_scale ? row_scale[r0 + k.r] : 1.f;
                float out[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ce = c + e;
                    const float is = c_t[wy + ce], ga = c_t[2 * wy + ce];
                    const float xh = ((e ? yv.y : yv.x) - c_t[ce]) * is;
                    float g = e ? gv.y : gv.x;
                    if (bn.relu && !(xh * ga + c_t[3 * wy + ce] > 0.f)) g = 0.f;
                    g = ga * is * (g - c_t[4 * wy + ce] - xh * c_t[5 * wy + ce]);
                    if (r0 + k.r >= n_valid) g = 0.f;
                    if (row_scale) g *= rs;
                    out[e] = g;
                }
                *reinterpret_cast<float2*>(g_t + k.r * wy + c) = make_float2(out[0], out[1]);
            }
        } else {
            Cursor k((int)threadIdx.x, wy);
#pragma unroll 4
            for (int i = threadIdx.x; i < rows * wy; i += kThreads, k.next()) {
                float g;
                if (has_bn) {       // g_y of the combine = BatchNorm backward of the tail's gradient, formed here
                    const int c = k.c;
                    const float is = c_t[wy + c], ga = c_t[2 * wy + c];
                    const float xh = (bn.y[(r0 + k.r) * bn.ld + c] - c_t[c]) * is;
                    g = bn.g_out[(r0 + k.r) * bn.ld + c];
                    if (bn.relu && !(xh * ga + c_t[3 * wy + c] > 0.f)) g = 0.f;
                    g = ga * is * (g - c_t[4 * wy + c] - xh * c_t[5 * wy + c]);
                    if (r0 + k.r >= n_valid) g = 0.f;
                } else {
                    g = gy[(r0 + k.r) * ld_gy + k.c];
                }
                if (row_scale) g *= row_scale[r0 + k.r];
                g_t[i] = g;
            }
        }
        for (int i = threadIdx.x; i < rows * S; i += kThreads) s_t[i] = scale ? scale[r0 * S + i] : 1.f;
        __syncthreads();
        if (bias_part) {
            for (int c = threadIdx.x; c < wy; c += kThreads) {
                float b = b_t[c];
                for (int r = 0; r < rows; ++r) b += g_t[r * wy + c];
                b_t[c] = b;
            }
        }
        if ((fo & 1) == 0) {
            // even f_out: pairs (8-byte lanes; a pair never straddles a scaler block, and every offset below is even)
            const int zw2 = zw >> 1;
            for (int t = 0; t < T; ++t) {
                float2* dst = reinterpret_cast<float2*>(gz + ((int64_t)t * n_nodes + r0) * zw);
                Cursor k((int)threadIdx.x, zw2);
                for (int i = threadIdx.x; i < rows * zw2; i += kThreads, k.next()) {
                    const int c = 2 * k.c, q = c / fo, o = c - q * fo;
                    const float2 g = *reinterpret_cast<const float2*>(g_t + k.r * wy + t * fo + o);
                    const float sc = s_t[k.r * S + q];
                    dst[i] = make_float2(g.x * sc, g.y * sc);
                }
            }
        } else {
            for (int t = 0; t < T; ++t) {
                float* dst = gz + ((int64_t)t * n_nodes + r0) * zw;
                Cursor k((int)threadIdx.x, zw);
                for (int i = threadIdx.x; i < rows * zw; i += kThreads, k.next()) {
                    const int q = k.c / fo, o = k.c - q * fo;
                    dst[i] = g_t[k.r * wy + t * fo + o] * s_t[k.r * S + q];
                }
            }
        }
    }
    if (bias_part) {
        for (int c = threadIdx.x; c < wy; c += kThreads) bias_part[(int64_t)c * G + blockIdx.x] = b_t[c];   // own column: no sync needed
    }
}

// Round 6: the backward of a ONE-tower, scaler-free combine behind BatchNorm (the simple / complex layers on the degree-class route and the
// identity-scaler configs: g_z [N, wy] = row_scale * BatchNorm-backward(g_out, y)) on dense rows of a width that is no multiple of four,
// as ONE flat array of 16-byte chunks (dgn_bn_tail.hip: column_partials_flat4 -- a thread takes chunk j of every period of lcm(wy, 4) floats, so
// its four columns, their BatchNorm constants and its row offsets inside the period never change).  combine_bwd staged 4-byte lanes
// through an LDS slab: 2.5 TB/s on [275 k, 75]; this form streams 16-byte lanes.  The same arithmetic per element in the same order; the bias
// gradient's per-workgroup partial is summed in another order (fp32, fixed: reproducible).
constexpr int kFlatUnroll = 4;
__global__ __launch_bounds__(kThreads) void combine_bwd_flat4(int64_t n_nodes, int wy, const float* __restrict__ row_scale, float* __restrict__ gz,
                                                              float* __restrict__ bias_part, const DgnBnGrad bn) {
    __shared__ float red[4][kThreads];
    const int G = (int)gridDim.x, b = (int)blockIdx.x, F = wy;
    const int g4 = (F & 1) ? 1 : 2, Pc = F / g4, R = 4 / g4;
    const int P = kThreads / Pc;
    const int p = (int)threadIdx.x / Pc, j = (int)threadIdx.x - p * Pc;
    const int64_t n_valid = bn.n_valid ? *bn.n_valid : n_nodes;
    const float inv_n = 1.f / (float)n_valid;
    struct Col { float mu, is, ga, be, m1, m2; };
    auto col = [&](int c) { return Col{bn.mean[c], bn.invstd[c], bn.gamma ? bn.gamma[c] : 1.f, bn.beta ? bn.beta[c] : 0.f, bn.sums[c] * inv_n, bn.sums[F + c] * inv_n}; };
    auto one = [&](float yv, float g, const Col& k, int64_t row, float rs) {      // (combine_bwd's arithmetic)
        const float xh = (yv - k.mu) * k.is;
        if (bn.relu && !(xh * k.ga + k.be > 0.f)) g = 0.f;
        g = k.ga * k.is * (g - k.m1 - xh * k.m2);
        if (row >= n_valid) g = 0.f;
        if (row_scale) g *= rs;
        return g;
    };
    const int64_t n_periods = n_nodes / R, stride = (int64_t)G * P;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < P) {
        const Col k0 = col((4 * j) % F), k1 = col((4 * j + 1) % F), k2 = col((4 * j + 2) % F), k3 = col((4 * j + 3) % F);
        const int ro0 = (4 * j) / F, ro1 = (4 * j + 1) / F, ro2 = (4 * j + 2) / F, ro3 = (4 * j + 3) / F;      // (a chunk touches at most two rows)
        for (int64_t s = (int64_t)b * P + p; s < n_periods; s += kFlatUnroll * stride) {
            float4 yv[kFlatUnroll], gv[kFlatUnroll];
            float ra[kFlatUnroll], rb[kFlatUnroll];
#pragma unroll
            for (int u = 0; u < kFlatUnroll; ++u) {
                const int64_t su = min(s + u * stride, n_periods - 1), off = su * ((int64_t)R * F) + 4 * j;
                yv[u] = *reinterpret_cast<const float4*>(bn.y + off);
                gv[u] = *reinterpret_cast<const float4*>(bn.g_out + off);
                ra[u] = row_scale ? row_scale[su * R + ro0] : 1.f;
                rb[u] = row_scale ? row_scale[su * R + ro3] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < kFlatUnroll; ++u) {
                if (s + u * stride < n_periods) {
                    const int64_t r0 = (s + u * stride) * R;
                    const float o0 = one(yv[u].x, gv[u].x, k0, r0 + ro0, ra[u]), o1 = one(yv[u].y, gv[u].y, k1, r0 + ro1, ro1 == ro0 ? ra[u] : rb[u]);
                    const float o2 = one(yv[u].z, gv[u].z, k2, r0 + ro2, ro2 == ro0 ? ra[u] : rb[u]), o3 = one(yv[u].w, gv[u].w, k3, r0 + ro3, rb[u]);
                    *reinterpret_cast<float4*>(gz + r0 * F + 4 * j) = make_float4(o0, o1, o2, o3);
                    acc[0] += o0; acc[1] += o1; acc[2] += o2; acc[3] += o3;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[e][threadIdx.x] = acc[e];
    __syncthreads();
    const int c = (int)threadIdx.x;
    if (c < F) {
        float sum = 0.f;
        for (int e = 0; e < 4; ++e) {
            for (int k = 0; k <= 4; ++k) {                      // the chunk of a period whose element e is column c: 4 j + e = c + k F
                const int v = c - e + k * F;
                if (v < 0 || (v & 3) || (v >> 2) >= Pc) continue;
                for (int q = 0; q < P; ++q) sum += red[e][q * Pc + (v >> 2)];
            }
        }
        if (b == 0) {                                            // the rows behind the last whole period
            const Col kc = col(c);
            for (int64_t n = n_periods * R; n < n_nodes; ++n) {
                const float o = one(bn.y[n * F + c], bn.g_out[n * F + c], kc, n, row_scale ? row_scale[n] : 1.f);
                gz[n * F + c] = o;
                sum += o;
            }
        }
        if (bias_part) bias_part[(int64_t)c * G + b] = sum;
    }
}

// g_bias[c] += (set: =) sum of the G slots (one workgroup per column, fixed order)
__global__ __launch_bounds__(kThreads) void bias_finalize(int wy, int G, const float* __restrict__ bias_part, float* __restrict__ g_bias, int set) {
    __shared__ float red[kThreads / 64];
    const int c = (int)blockIdx.x;
    float s = 0.f;
    for (int g = threadIdx.x; g < G; g += kThreads) s += bias_part[(int64_t)c * G + g];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (red[0] + red[1]) + (red[2] + red[3]);
        g_bias[c] = set ? 0.f + v : g_bias[c] + v;          // (0.f + v: the bits of "zero-fill, then add")
    }
}

int bwd_groups(int64_t n_nodes, int rows) { return (int)std::min<int64_t>((n_nodes + rows - 1) / rows, kMaxGroups); }

int check_shape(const char* fn, int64_t n_nodes, int32_t T, int32_t S, int32_t fo, bool has_scale) {
    if (n_nodes < 0 || T < 1 || S < 1 || fo < 1 || (!has_scale && S != 1)) { set_error("%s: bad shape", fn); return DGN_ERR_INVALID; }
    if ((int64_t)T * fo > kMaxWidth) { set_error("%s: n_towers * f_out > 4096 is not supported", fn); return DGN_ERR_INVALID; }
    return DGN_OK;
}

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" int dgn_scale_combine_forward(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* z, const float* scale,
                                         const float* bias, const float* row_scale, float* y, int64_t ld_y, void* stream) {
    return dgn::scale_combine_forward_stats(n_nodes, T, S, fo, z, scale, bias, row_scale, y, ld_y, nullptr, 0, nullptr, stream);
}

// ... with BatchNorm's column partials of y riding in the pass (combine_fwd: bn_part).  part == NULL: the plain pass.  *slots = the G bn_finalize
// reads, or 0 where the statistics did not ride (rows wider than a workgroup, more slabs than `part` holds): the caller runs bn_stats.
int dgn::scale_combine_forward_stats(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* z, const float* scale, const float* bias,
                                     const float* row_scale, float* y, int64_t ld_y, double* part, size_t part_bytes, int* slots, void* stream) {
    if (int rc = check_shape("dgn_scale_combine_forward", n_nodes, T, S, fo, scale != nullptr)) return rc;
    if (slots) *slots = 0;
    if (n_nodes == 0) return DGN_OK;
    if (!z || !y || ld_y < (int64_t)T * fo) { set_error("dgn_scale_combine_forward: null buffer or ld_y too small"); return DGN_ERR_INVALID; }
    const int rows = fwd_slab_rows(n_nodes);
    const dim3 grid((unsigned)((n_nodes + rows - 1) / rows)), block(kThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* bp = (part && slots && T * fo <= kThreads && (size_t)2 * T * fo * grid.x * sizeof(double) <= part_bytes) ? part : nullptr;
    if (bp) *slots = (int)grid.x;
    switch (S) {
        case 1: hipLaunchKernelGGL(combine_fwd<1>, grid, block, 0, st, n_nodes, rows, T, S, fo, z, scale, bias, row_scale, y, ld_y, bp); break;
        case 2: hipLaunchKernelGGL(combine_fwd<2>, grid, block, 0, st, n_nodes, rows, T, S, fo, z, scale, bias, row_scale, y, ld_y, bp); break;
        case 3: hipLaunchKernelGGL(combine_fwd<3>, grid, block, 0, st, n_nodes, rows, T, S, fo, z, scale, bias, row_scale, y, ld_y, bp); break;
        default: hipLaunchKernelGGL(combine_fwd<0>, grid, block, 0, st, n_nodes, rows, T, S, fo, z, scale, bias, row_scale, y, ld_y, bp); break;
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" size_t dgn_scale_combine_backward_workspace_bytes(int64_t n_nodes, int32_t T, int32_t fo) {
    if (n_nodes <= 0 || T < 1 || fo < 1 || (int64_t)T * fo > kMaxWidth) return 0;
    return (size_t)T * fo * bwd_groups(n_nodes, bwd_slab_rows(T * fo)) * sizeof(float);
}

extern "C" int dgn_scale_combine_backward(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* g_y, int64_t ld_gy,
                                          const float* scale, const float* row_scale, float* g_z, float* g_bias, void* ws,
                                          size_t ws_bytes, const DgnBnGrad* bn, void* stream) {
    return dgn::scale_combine_backward_impl(n_nodes, T, S, fo, g_y, ld_gy, scale, row_scale, g_z, g_bias, ws, ws_bytes, bn, stream, 0);
}

// set_bias: g_bias is WRITTEN (the whole-layer calls: no zero-fill launch in front), else accumulated into (the C ABI's contract)
int dgn::scale_combine_backward_impl(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* g_y, int64_t ld_gy,
                                     const float* scale, const float* row_scale, float* g_z, float* g_bias, void* ws,
                                     size_t ws_bytes, const DgnBnGrad* bn, void* stream, int set_bias) {
    if (int rc = check_shape("dgn_scale_combine_backward", n_nodes, T, S, fo, scale != nullptr)) return rc;
    if (n_nodes == 0) return DGN_OK;
    const int wy = T * fo;
    if (bn) {
        if (!bn->g_out || !bn->y || !bn->mean || !bn->invstd || !bn->sums || bn->ld < wy || wy > 1024) {
            set_error("dgn_scale_combine_backward: incomplete DgnBnGrad (or n_towers * f_out > 1024)");
            return DGN_ERR_INVALID;
        }
    } else if (!g_y || ld_gy < wy) {
        set_error("dgn_scale_combine_backward: null g_y or ld_gy too small");
        return DGN_ERR_INVALID;
    }
    if (!g_z) { set_error("dgn_scale_combine_backward: null g_z"); return DGN_ERR_INVALID; }
    if (g_bias && (!ws || ws_bytes < dgn_scale_combine_backward_workspace_bytes(n_nodes, T, fo))) {
        set_error("dgn_scale_combine_backward: workspace too small (the bias gradient needs dgn_scale_combine_backward_workspace_bytes())");
        return DGN_ERR_WORKSPACE;
    }
    const int rows = bwd_slab_rows(wy), G = bwd_groups(n_nodes, rows);
    const size_t lds = ((size_t)rows * wy + (size_t)rows * S + wy + (bn ? 6 * (size_t)wy : 0)) * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = g_bias ? static_cast<float*>(ws) : nullptr;
    {   // one tower, no scaler table, dense rows of a width that is no multiple of four, 16-byte aligned: flat 16-byte chunks
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        if (bn && T == 1 && S == 1 && !scale && (wy & 3) != 0 && bn->ld == wy && wy / ((wy & 1) ? 1 : 2) <= kThreads && al16(bn->y) && al16(bn->g_out) &&
            al16(g_z)) {
            const int g4 = (wy & 1) ? 1 : 2, P = kThreads / (wy / g4);
            const int64_t n_periods = n_nodes / (4 / g4);
            const int Gf = (int)std::max<int64_t>(1, std::min<int64_t>(G, (n_periods + (int64_t)P * kFlatUnroll * 2 - 1) / ((int64_t)P * kFlatUnroll * 2)));
            hipLaunchKernelGGL(combine_bwd_flat4, dim3(Gf), dim3(kThreads), 0, st, n_nodes, wy, row_scale, g_z, part, *bn);
            if (g_bias) hipLaunchKernelGGL(bias_finalize, dim3(wy), dim3(kThreads), 0, st, wy, Gf, (const float*)part, g_bias, set_bias);
            DGN_HIP_CHECK(hipGetLastError());
            return DGN_OK;
        }
    }
    const DgnBnGrad none{};
    hipLaunchKernelGGL(combine_bwd, dim3(G), dim3(kThreads), lds, st, n_nodes, rows, T, S, fo, g_y, ld_gy, scale, row_scale, g_z, part,
                       bn ? *bn : none, bn ? 1 : 0);
    if (g_bias) hipLaunchKernelGGL(bias_finalize, dim3(wy), dim3(kThreads), 0, st, wy, G, (const float*)part, g_bias, set_bias);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
