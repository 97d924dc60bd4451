// Layer tail for gfx950: BatchNorm1d over the node dimension (+ ReLU + residual), forward and backward
// (reference: realworld_benchmark/nets/dgn_layer.py:123-128, :194-199, :272-273 -- nn.BatchNorm1d, F.relu, h_in + h).
// Streaming kernels over [N, F] with F <= 1024.  Column statistics: every workgroup keeps fp64 partial sums of its
// rows (one pass, no fp32 cancellation in E[x^2] - mean^2) and writes them to its own workspace slot; a small
// finalize kernel adds the slots in a fixed order.  No atomics: same-address atomics cost ~22 ns each on this part
// (measured: 1 000 workgroups adding into 140 addresses took longer than streaming the 77 MB input), and the
// fixed order makes the statistics bitwise reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "dgn_common.hpp"

namespace dgn {
namespace {

constexpr int kMaxF = 1024;       // widest row the tail kernels accept
constexpr int kMaxGroups = 2048;  // workgroups (= workspace slots) of the statistics kernels
constexpr int kUnroll = 8;        // rows in flight per thread

// number of statistics workgroups: every thread should see about kUnroll rows, at most kMaxGroups
int stat_groups(int64_t n_rows, int F) {
    const int P = std::max(1, 256 / F);
    const int64_t g = (n_rows + (int64_t)P * kUnroll - 1) / ((int64_t)P * kUnroll);
    return (int)std::min<int64_t>(std::max<int64_t>(g, 1), kMaxGroups);
}

// Column sums of two quantities over the rows of this workgroup.  Threads are laid out (row phase p, column c):
// with F = 70 a 256-thread block reads 3 consecutive rows at a time, all lanes busy, every read coalesced.
// Workgroup b takes the row triples b, b + G, b + 2G, ...; kUnroll rows are loaded before the dependent fp64
// adds.  The phases are folded through LDS and the block's partial goes to part[(q * F + c) * G + b].
template <class Fn>
__device__ __forceinline__ void column_partials(int64_t n_rows, int F, double* __restrict__ part, Fn&& fn) {
    __shared__ double red[2][256];
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int P = max(1, 256 / F);                 // row phases per block
    const int64_t stride = (int64_t)G * P;
    for (int c0 = 0; c0 < F; c0 += 256) {          // (one iteration unless F > 256)
        const int p = (int)threadIdx.x / F, c = c0 + (int)threadIdx.x % F;
        double a0 = 0.0, a1 = 0.0;
        if (p < P && c < F) {
            for (int64_t n = (int64_t)b * P + p; n < n_rows; n += kUnroll * stride) {
                float v0[kUnroll], v1[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) fn(min(n + u * stride, n_rows - 1), c, v0[u], v1[u]);
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    if (n + u * stride < n_rows) {
                        a0 += (double)v0[u];
                        a1 += (double)v1[u];
                    }
                }
            }
        }
        red[0][threadIdx.x] = a0;
        red[1][threadIdx.x] = a1;
        __syncthreads();
        if (p == 0 && c < F) {
            for (int q = 1; q < P; ++q) { a0 += red[0][threadIdx.x + q * F]; a1 += red[1][threadIdx.x + q * F]; }
            part[(int64_t)c * G + b] = a0;
            part[((int64_t)F + c) * G + b] = a1;
        }
        __syncthreads();
    }
}

// The same with a thread per column PAIR (even F, rows 8-byte aligned): 8-byte lanes, 256 / (F/2) row phases.  fn fills
// v0[2], v1[2] for columns c, c + 1 of row n.
struct NoPost {
    __device__ __forceinline__ void operator()(int64_t, int, const float (&)[2]) const {}
};
// (post(n, c, v0) runs after ALL loads of the unrolled group: a store between the loads would be waited for with them)
template <class Fn, class Post = NoPost>
__device__ __forceinline__ void column_partials_pairs(int64_t n_rows, int F, double* __restrict__ part, Fn&& fn, Post&& post = Post()) {
    __shared__ double red[4][256];
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int F2 = F >> 1;
    const int P = max(1, 256 / F2);
    const int64_t stride = (int64_t)G * P;
    for (int c0 = 0; c0 < F2; c0 += 256) {         // (one iteration unless F > 512)
        const int p = (int)threadIdx.x / F2, c2 = c0 + (int)threadIdx.x % F2;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (p < P && c2 < F2) {
            for (int64_t n = (int64_t)b * P + p; n < n_rows; n += kUnroll * stride) {
                float v0[kUnroll][2], v1[kUnroll][2];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) fn(min(n + u * stride, n_rows - 1), 2 * c2, v0[u], v1[u]);
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    if (n + u * stride < n_rows) {
                        post(n + u * stride, 2 * c2, v0[u]);
                        a[0] += (double)v0[u][0];
                        a[1] += (double)v0[u][1];
                        a[2] += (double)v1[u][0];
                        a[3] += (double)v1[u][1];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = a[q];
        __syncthreads();
        if (p == 0 && c2 < F2) {
            for (int q = 1; q < P; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] += red[i][threadIdx.x + q * F2];
            const int c = 2 * c2;
            part[(int64_t)c * G + b] = a[0];
            part[(int64_t)(c + 1) * G + b] = a[1];
            part[((int64_t)F + c) * G + b] = a[2];
            part[((int64_t)F + c + 1) * G + b] = a[3];
        }
        __syncthreads();
    }
}

// one workgroup per column: sum of the G slots of both quantities in a fixed order (thread-strided partials,
// wave shuffles, then the four wave sums through LDS); the result is valid in thread 0
__device__ __forceinline__ void slot_sums(const double* __restrict__ part, int F, int G, int c, double& s0, double& s1) {
    __shared__ double red[2][4];
    s0 = 0.0;
    s1 = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) {
        s0 += part[(int64_t)c * G + g];
        s1 += part[((int64_t)F + c) * G + g];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    s0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}

// Flat element loop of the apply kernels (grid-stride, one element per trip: unrolling four trips with clamped
// indices measured 15-30 % SLOWER here -- the launch already has ~9 elements per thread in flight across the grid).
template <class Load, class Store>
__device__ __forceinline__ void flat_loop(int64_t total, int F, Load&& load, Store&& store) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / F;
        const int c = (int)(idx - n * F);
        store(n, c, load(n, c));
    }
}

// forward partials: sum x, sum x^2
// (n_valid != NULL: only the first *n_valid rows are the batch -- the rest is padding of a fixed-capacity batch, DgnBnGrad.n_valid)
template <bool PAIRS>
__global__ __launch_bounds__(256) void bn_stats(int64_t n_rows, int F, const float* __restrict__ x, int64_t ld,
                                                double* __restrict__ part, const int64_t* __restrict__ n_valid) {
    if (n_valid) n_rows = min(n_rows, *n_valid);
    if constexpr (PAIRS) {
        column_partials_pairs(n_rows, F, part, [&](int64_t n, int c, float (&v0)[2], float (&v1)[2]) {
            const float2 v = *reinterpret_cast<const float2*>(x + n * ld + c);
            v0[0] = v.x; v0[1] = v.y;
            v1[0] = v.x * v.x; v1[1] = v.y * v.y;
        });
    } else {
        column_partials(n_rows, F, part, [&](int64_t n, int c, float& v0, float& v1) {
            const float v = x[n * ld + c];
            v0 = v;
            v1 = v * v;
        });
    }
}

// ---- dense rows of a width that is not a multiple of four (hidden 75, 65: the reference's simple-layer configs), 16-byte lanes ---------------
// The [n_rows, F] tensor is ONE flat array when its rows are dense (ld == F): 16-byte chunk q covers floats 4q .. 4q + 3, whatever rows
// they belong to.  A PERIOD of F / gcd(F, 4) chunks covers 4 / gcd(F, 4) whole rows, so a thread that always takes chunk j of a period
// sees the same four columns (4j + e) mod F every time: per-thread column accumulators and column constants, as in the thread-per-column
// kernels, with four times the bytes per load instruction (used by bn_bwd_stats_flat4 on large batches; the forward statistics gained nothing from it:
// profiles/NOTES.md).  Thread (p, j): period b * P + p, + G * P, ...; the workgroup folds (p, the four (j, e) of a column) through LDS in
// a fixed order; the rows behind the last whole period are added by workgroup 0's column threads.  fn(offset, e0 columns, v0[4], v1[4])
// fills the two quantities of the chunk at float offset `offset`; fn1(row, c, v0, v1) the same for one element (the remainder rows).
constexpr int kFlatUnroll = 4;
__host__ __device__ inline int flat_gcd4(int F) { return (F & 3) == 0 ? 4 : ((F & 1) == 0 ? 2 : 1); }
template <class Fn4, class Fn1>
__device__ __forceinline__ void column_partials_flat4(int64_t n_rows, int F, double* __restrict__ part, Fn4&& fn4, Fn1&& fn1) {
    __shared__ double red[8][256];
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int g4 = flat_gcd4(F), Pc = F / g4, R = 4 / g4;       // chunks and rows per period
    const int P = 256 / Pc;                                      // periods per workgroup trip (F <= 256)
    const int p = (int)threadIdx.x / Pc, j = (int)threadIdx.x - p * Pc;
    const int64_t n_periods = n_rows / R, stride = (int64_t)G * P;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (p < P) {
        for (int64_t s = (int64_t)b * P + p; s < n_periods; s += kFlatUnroll * stride) {
            float v0[kFlatUnroll][4], v1[kFlatUnroll][4];
#pragma unroll
            for (int u = 0; u < kFlatUnroll; ++u) fn4(min(s + u * stride, n_periods - 1) * ((int64_t)R * F) + 4 * j, v0[u], v1[u]);
#pragma unroll
            for (int u = 0; u < kFlatUnroll; ++u) {
                if (s + u * stride < n_periods) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] += (double)v0[u][e]; a[4 + e] += (double)v1[u][e]; }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[q][threadIdx.x] = a[q];
    __syncthreads();
    const int c = (int)threadIdx.x;
    if (c < F) {
        double s0 = 0.0, s1 = 0.0;
        for (int e = 0; e < 4; ++e) {
            for (int k = 0; k <= 4; ++k) {                      // the chunk j of a period whose element e is column c: 4 j + e = c + k F
                const int v = c - e + k * F;
                if (v < 0 || (v & 3) || (v >> 2) >= Pc) continue;
                for (int q = 0; q < P; ++q) { s0 += red[e][q * Pc + (v >> 2)]; s1 += red[4 + e][q * Pc + (v >> 2)]; }
            }
        }
        if (b == 0) {
            for (int64_t n = n_periods * R; n < n_rows; ++n) {
                float v0, v1;
                fn1(n, c, v0, v1);
                s0 += (double)v0;
                s1 += (double)v1;
            }
        }
        part[(int64_t)c * G + b] = s0;
        part[((int64_t)F + c) * G + b] = s1;
    }
}
// (dense rows, a 16-byte aligned base, a width the thread-per-pair kernels do not take or take at 8 bytes only, a period within one workgroup)
bool flat4_ok(int F, int64_t ld, const void* a, const void* b = nullptr) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return ld == F && (F & 3) != 0 && F / flat_gcd4(F) <= 256 && al16(a) && al16(b);
}

// mean / invstd per column, running statistics (unbiased variance, like torch)
__global__ __launch_bounds__(256) void bn_finalize(int64_t n_rows, int F, int G, const double* __restrict__ part,
                                                   float* running_mean, float* running_var, float momentum, float eps,
                                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                   const int64_t* __restrict__ n_valid, int64_t* nbt = nullptr, int n_nbt = 0) {
    if (n_valid) n_rows = min(n_rows, *n_valid);
    const int c = (int)blockIdx.x;
    if (c == 0 && (int)threadIdx.x < n_nbt) nbt[threadIdx.x] += 1;       // torch's num_batches_tracked += 1 (n_nbt <= 256 modules)
    double s0, s1;
    slot_sums(part, F, G, c, s0, s1);
    if (threadIdx.x != 0) return;
    const double mu = s0 / (double)n_rows;
    double m2 = s1 - mu * s0;                      // sum (x - mean)^2
    if (m2 < 0.0) m2 = 0.0;
    const float mean = (float)mu;
    save_mean[c] = mean;
    save_invstd[c] = (float)(1.0 / sqrt(m2 / (double)n_rows + (double)eps));
    if (running_mean) {
        const float unbiased = (float)(n_rows > 1 ? m2 / (double)(n_rows - 1) : m2 / (double)n_rows);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// normalise (+ReLU, +residual).  Training: mean / invstd from bn_finalize; eval (mean == NULL): running statistics.
// VEC columns per thread, the column constants once per workgroup in LDS (eval: 1 / sqrt(var + eps) was a square root and a division
// per ELEMENT in the scalar round-1 kernel) and the (row, column) of a thread's next element by increments instead of a 64-bit division
// per element: C5's [10 M, 128] rows 5.85 -> 2.3 ms.  Same arithmetic, element by element.
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_vec(int64_t n_rows, int F, const float* __restrict__ x, int64_t ld,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                    float eps, int relu, const float* __restrict__ residual, float* __restrict__ y) {
    __shared__ float s_mu[kMaxF], s_is[kMaxF], s_ga[kMaxF], s_be[kMaxF];
    for (int c = threadIdx.x; c < F; c += 256) {
        s_mu[c] = mean ? mean[c] : running_mean[c];
        s_is[c] = mean ? invstd[c] : 1.f / sqrtf(running_var[c] + eps);
        s_ga[c] = gamma ? gamma[c] : 1.f;
        s_be[c] = beta ? beta[c] : 0.f;
    }
    __syncthreads();
    using V = float __attribute__((ext_vector_type(VEC)));
    const int FV = F / VEC;
    const int64_t total = n_rows * FV, stride = (int64_t)gridDim.x * 256;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t n = idx / FV;
    int c = (int)(idx - n * FV);
    const int64_t dq = stride / FV;
    const int dr = (int)(stride - dq * FV);
    for (; idx < total; idx += stride) {
        const int64_t at = n * ld + (int64_t)c * VEC;
        V v = *reinterpret_cast<const V*>(x + at), r = v;
        if (residual) r = *reinterpret_cast<const V*>(residual + at);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int cc = c * VEC + i;
            float e = (v[i] - s_mu[cc]) * s_is[cc] * s_ga[cc] + s_be[cc];
            if (relu) e = fmaxf(e, 0.f);
            if (residual) e += r[i];
            v[i] = e;
        }
        *reinterpret_cast<V*>(y + at) = v;
        n += dq; c += dr;
        if (c >= FV) { c -= FV; ++n; }
    }
}

// bn_apply_vec on dense rows of a width that is not a multiple of four: flat 16-byte chunks (see column_partials_flat4), the column of a
// chunk's first element kept by increments; the (n_rows F) mod 4 floats behind the last chunk by the last workgroup's first threads
__global__ __launch_bounds__(256) void bn_apply_flat4(int64_t n_rows, int F, const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                                      int relu, const float* __restrict__ residual, float* __restrict__ y) {
    __shared__ float s_a[kMaxF], s_b[kMaxF], s_mu[kMaxF], s_ga[kMaxF];
    for (int c = threadIdx.x; c < F; c += 256) {
        s_mu[c] = mean ? mean[c] : running_mean[c];
        s_a[c] = mean ? invstd[c] : 1.f / sqrtf(running_var[c] + eps);
        s_ga[c] = gamma ? gamma[c] : 1.f;
        s_b[c] = beta ? beta[c] : 0.f;
    }
    __syncthreads();
    auto one = [&](float v, float r, int c) {        // (bn_apply_vec's arithmetic, in its order)
        float e = (v - s_mu[c]) * s_a[c] * s_ga[c] + s_b[c];
        if (relu) e = fmaxf(e, 0.f);
        if (residual) e += r;
        return e;
    };
    const int64_t total = n_rows * F, chunks = total >> 2, stride = (int64_t)gridDim.x * 256;
    int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int c0 = (int)((4 * q) % F);
    const int dc = (int)((4 * stride) % F);
    for (; q < chunks; q += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[q];
        float4 r = v;
        if (residual) r = reinterpret_cast<const float4*>(residual)[q];
        int c1 = c0 + 1; if (c1 >= F) c1 -= F;
        int c2 = c1 + 1; if (c2 >= F) c2 -= F;
        int c3 = c2 + 1; if (c3 >= F) c3 -= F;
        reinterpret_cast<float4*>(y)[q] = make_float4(one(v.x, r.x, c0), one(v.y, r.y, c1), one(v.z, r.z, c2), one(v.w, r.w, c3));
        c0 += dc;
        if (c0 >= F) c0 -= F;
    }
    if (blockIdx.x == gridDim.x - 1) {
        const int64_t i = 4 * chunks + threadIdx.x;
        if (i < total) y[i] = one(x[i], residual ? residual[i] : 0.f, (int)(i % F));
    }
}

// backward partials: sum g', sum g' * xhat      (g' = g masked by the ReLU)
template <bool PAIRS>
__global__ __launch_bounds__(256) void bn_bwd_stats(int64_t n_rows, int F, const float* __restrict__ gy, const float* __restrict__ x,
                                                    int64_t ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                    double* __restrict__ part, const int64_t* __restrict__ n_valid) {
    if (n_valid) n_rows = min(n_rows, *n_valid);
    // a thread's columns are fixed for the whole row loop (one pass of the helpers' column loop unless F > 256 / 512): their BatchNorm
    // constants are read ONCE -- inside the loop they were four more (cached) loads per element in the same in-order queue as the rows,
    // and the ReLU variant ran at half the rate of the plain one
    struct Col { float mu, is, ga, be; };
    auto col = [&](int c) {
        const int cc = min(c, F - 1);
        return Col{mean[cc], invstd[cc], gamma ? gamma[cc] : 1.f, beta ? beta[cc] : 0.f};
    };
    auto one = [&](float xv, float g, const Col& k, float& v0, float& v1) {
        const float xh = (xv - k.mu) * k.is;
        if (relu && !(xh * k.ga + k.be > 0.f)) g = 0.f;
        v0 = g;
        v1 = g * xh;
    };
    if constexpr (PAIRS) {
        const int F2 = F >> 1;
        const bool fixed = F2 <= 256;
        const int mine = 2 * ((int)threadIdx.x % F2);
        const Col k0 = col(mine), k1 = col(mine + 1);
        column_partials_pairs(n_rows, F, part, [&](int64_t n, int c, float (&v0)[2], float (&v1)[2]) {
            const float2 xv = *reinterpret_cast<const float2*>(x + n * ld + c), g = *reinterpret_cast<const float2*>(gy + n * ld + c);
            one(xv.x, g.x, fixed ? k0 : col(c), v0[0], v1[0]);
            one(xv.y, g.y, fixed ? k1 : col(c + 1), v0[1], v1[1]);
        });
    } else {
        const bool fixed = F <= 256;
        const Col k0 = col((int)threadIdx.x % F);
        column_partials(n_rows, F, part, [&](int64_t n, int c, float& v0, float& v1) { one(x[n * ld + c], gy[n * ld + c], fixed ? k0 : col(c), v0, v1); });
    }
}

__global__ __launch_bounds__(256) void bn_bwd_stats_flat4(int64_t n_rows, int F, const float* __restrict__ gy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                          double* __restrict__ part, const int64_t* __restrict__ n_valid) {
    if (n_valid) n_rows = min(n_rows, *n_valid);
    struct Col { float mu, is, ga, be; };
    auto col = [&](int c) { return Col{mean[c], invstd[c], gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f}; };
    auto one = [&](float xv, float g, const Col& k, float& v0, float& v1) {      // (bn_bwd_stats' arithmetic)
        const float xh = (xv - k.mu) * k.is;
        if (relu && !(xh * k.ga + k.be > 0.f)) g = 0.f;
        v0 = g;
        v1 = g * xh;
    };
    const int Pc = F / flat_gcd4(F), j = (int)threadIdx.x % Pc;
    const Col k0 = col((4 * j) % F), k1 = col((4 * j + 1) % F), k2 = col((4 * j + 2) % F), k3 = col((4 * j + 3) % F);
    column_partials_flat4(n_rows, F, part, [&](int64_t off, float (&v0)[4], float (&v1)[4]) {
        const float4 xv = *reinterpret_cast<const float4*>(x + off), g = *reinterpret_cast<const float4*>(gy + off);
        one(xv.x, g.x, k0, v0[0], v1[0]);
        one(xv.y, g.y, k1, v0[1], v1[1]);
        one(xv.z, g.z, k2, v0[2], v1[2]);
        one(xv.w, g.w, k3, v0[3], v1[3]);
    }, [&](int64_t n, int c, float& v0, float& v1) { one(x[n * F + c], gy[n * F + c], col(c), v0, v1); });
}

// sums[c] = sum g' (= d beta), sums[F + c] = sum g' xhat (= d gamma)
__global__ __launch_bounds__(256) void bn_bwd_finalize(int F, int G, const double* __restrict__ part, float* __restrict__ sums,
                                                       float* g_gamma, float* g_beta) {
    const int c = (int)blockIdx.x;
    double s0, s1;
    slot_sums(part, F, G, c, s0, s1);
    if (threadIdx.x != 0) return;
    sums[c] = (float)s0;
    sums[F + c] = (float)s1;
    if (g_beta) g_beta[c] = (float)s0;
    if (g_gamma) g_gamma[c] = (float)s1;
}

__global__ __launch_bounds__(256) void bn_bwd_apply(int64_t n_rows, int F, const float* __restrict__ gy, const float* __restrict__ x,
                                                    int64_t ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                    const float* __restrict__ sums, float* __restrict__ gx, const int64_t* __restrict__ n_valid) {
    const int64_t nv = n_valid ? min(n_rows, *n_valid) : n_rows;
    const float inv_n = 1.f / (float)nv;
    flat_loop(n_rows * F, F, [&](int64_t n, int c) {
        const float is = invstd[c], ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        const float xh = (x[n * ld + c] - mean[c]) * is;
        float g = gy[n * ld + c];
        if (relu && !(xh * ga + be > 0.f)) g = 0.f;
        return n < nv ? ga * is * (g - sums[c] * inv_n - xh * sums[F + c] * inv_n) : 0.f;
    }, [&](int64_t n, int c, float v) { gx[n * ld + c] = v; });
}

// ---- bias + activation (+ residual): the tail of an FCLayer (Linear -> activation, layers.py:101-112) -------------------
// act: 0 none, 1 ReLU, 2 LeakyReLU(slope)
__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}
__device__ __forceinline__ float act_grad(float v, int act, float slope) {      // d act / d v  (torch: 0 / slope at v <= 0)
    if (act == 1) return v > 0.f ? 1.f : 0.f;
    if (act == 2) return v > 0.f ? 1.f : slope;
    return 1.f;
}

__global__ __launch_bounds__(256) void bias_act_fwd(int64_t n_rows, int F, const float* __restrict__ x, int64_t ld,
                                                    const float* __restrict__ bias, int act, float slope,
                                                    const float* __restrict__ residual, float* __restrict__ y) {
    flat_loop(n_rows * F, F, [&](int64_t n, int c) {
        float v = act_fwd(x[n * ld + c] + (bias ? bias[c] : 0.f), act, slope);
        if (residual) v += residual[n * ld + c];
        return v;
    }, [&](int64_t n, int c, float v) { y[n * ld + c] = v; });
}

// The same on dense rows of even width (ld == F): flat 8-byte lanes, a pair never straddles a row.
__global__ __launch_bounds__(256) void bias_act_fwd_pairs(int64_t n_pairs, int F, const float2* __restrict__ x,
                                                          const float* __restrict__ bias, int act, float slope,
                                                          const float2* __restrict__ residual, float2* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((2 * i) % F);
        const float2 xv = x[i];
        float2 v = make_float2(act_fwd(xv.x + (bias ? bias[c] : 0.f), act, slope), act_fwd(xv.y + (bias ? bias[c + 1] : 0.f), act, slope));
        if (residual) {
            const float2 r = residual[i];
            v.x += r.x;
            v.y += r.y;
        }
        y[i] = v;
    }
}

// g_x = g_y * act'(x + bias), and the partials of the bias gradient sum_n g_x[n, c] (second quantity unused)
template <bool PAIRS>
__global__ __launch_bounds__(256) void bias_act_bwd(int64_t n_rows, int F, const float* __restrict__ gy, const float* __restrict__ x,
                                                    int64_t ld, const float* __restrict__ bias, int act, float slope,
                                                    float* __restrict__ gx, double* __restrict__ part) {
    if constexpr (PAIRS) {
        column_partials_pairs(n_rows, F, part, [&](int64_t n, int c, float (&v0)[2], float (&v1)[2]) {
            const float2 g = *reinterpret_cast<const float2*>(gy + n * ld + c), xv = *reinterpret_cast<const float2*>(x + n * ld + c);
            v0[0] = g.x * act_grad(xv.x + (bias ? bias[c] : 0.f), act, slope);
            v0[1] = g.y * act_grad(xv.y + (bias ? bias[c + 1] : 0.f), act, slope);
            v1[0] = v1[1] = 0.f;
        }, [&](int64_t n, int c, const float (&v0)[2]) { *reinterpret_cast<float2*>(gx + n * ld + c) = make_float2(v0[0], v0[1]); });
    } else {
        column_partials(n_rows, F, part, [&](int64_t n, int c, float& v0, float& v1) {
            const float g = gy[n * ld + c] * act_grad(x[n * ld + c] + (bias ? bias[c] : 0.f), act, slope);
            gx[n * ld + c] = g;        // (rows clamped at the tail are rewritten with the same value)
            v0 = g;
            v1 = 0.f;
        });
    }
}

__global__ __launch_bounds__(256) void bias_act_finalize(int F, int G, const double* __restrict__ part, float* __restrict__ g_bias) {
    const int c = (int)blockIdx.x;
    double s0, s1;
    slot_sums(part, F, G, c, s0, s1);
    if (threadIdx.x == 0) g_bias[c] = (float)s0;
}

// thread-per-column-pair kernels: even width and row stride, 8-byte aligned bases
bool pairs_ok(int F, int64_t ld, const void* a, const void* b = nullptr, const void* c = nullptr) {
    auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
    return (F & 1) == 0 && (ld & 1) == 0 && al8(a) && al8(b) && al8(c);
}

unsigned flat_grid(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32); }

// widest vector the rows allow (F, ld and the three addresses), VEC = 1 otherwise
void launch_bn_apply(hipStream_t stream, int64_t n_rows, int F, const float* x, int64_t ld, const float* gamma, const float* beta, const float* mean,
                     const float* invstd, const float* running_mean, const float* running_var, float eps, int relu, const float* residual, float* y) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual);
    if (flat4_ok(F, ld, x, y) && (reinterpret_cast<uintptr_t>(residual) & 15) == 0) {
        hipLaunchKernelGGL(bn_apply_flat4, dim3(std::max(1u, flat_grid((n_rows * F) >> 2))), dim3(256), 0, stream, n_rows, F, x, gamma, beta, mean, invstd, running_mean,
                           running_var, eps, relu, residual, y);
        return;
    }
    const int vec = (F % 4 == 0 && ld % 4 == 0 && bits % 16 == 0) ? 4 : ((F % 2 == 0 && ld % 2 == 0 && bits % 8 == 0) ? 2 : 1);
    const dim3 grid(flat_grid(n_rows * (F / vec)));
    if (vec == 4) hipLaunchKernelGGL(bn_apply_vec<4>, grid, dim3(256), 0, stream, n_rows, F, x, ld, gamma, beta, mean, invstd, running_mean, running_var, eps, relu, residual, y);
    else if (vec == 2) hipLaunchKernelGGL(bn_apply_vec<2>, grid, dim3(256), 0, stream, n_rows, F, x, ld, gamma, beta, mean, invstd, running_mean, running_var, eps, relu, residual, y);
    else hipLaunchKernelGGL(bn_apply_vec<1>, grid, dim3(256), 0, stream, n_rows, F, x, ld, gamma, beta, mean, invstd, running_mean, running_var, eps, relu, residual, y);
}

size_t part_bytes(int64_t n_rows, int F) { return (size_t)2 * F * stat_groups(n_rows, F) * sizeof(double); }

}  // namespace
}  // namespace dgn

// ---- dropout (F.dropout at the end of DGNLayerSimple / DGNLayerComplex.forward, nets/dgn_layer.py:130, :201; configs HIV / PCBA /
// CIFAR10: dropout 0.3) -----------------------------------------------------------------------------------------------------------
// Philox4x32-10, counter = (index of the 8-element group, half, offset lo, offset hi), key = the two halves of *seed (a DEVICE scalar:
// drawn by the caller's generator, so the launch is capturable and replays draw new masks when the scalar is refreshed inside the
// captured region).  A thread owns 8 consecutive elements = one byte of the keep mask the backward re-applies.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__global__ __launch_bounds__(256) void dropout_fwd(int64_t n, const float* x /* may be y: in place */, uint32_t threshold, float scale,
                                                   const int64_t* __restrict__ seed, uint64_t offset, float* y,
                                                   unsigned char* __restrict__ mask) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // group of 8 elements
    const int64_t e0 = g * 8;
    if (e0 >= n) return;
    const uint64_t sd = (uint64_t)*seed;
    uint32_t r[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t c[4] = {(uint32_t)g, (uint32_t)((uint64_t)g >> 32) * 2u + (uint32_t)half, (uint32_t)offset, (uint32_t)(offset >> 32)};
        philox4x32_10(c, (uint32_t)sd, (uint32_t)(sd >> 32));
#pragma unroll
        for (int i = 0; i < 4; ++i) r[4 * half + i] = c[i];
    }
    float v[8];
    const bool full = e0 + 8 <= n && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (full) {
        const float4 a = *reinterpret_cast<const float4*>(x + e0), b = *reinterpret_cast<const float4*>(x + e0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = e0 + i < n ? x[e0 + i] : 0.f;
    }
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool keep = r[i] >= threshold;                                   // P(keep) = 1 - p
        m |= (keep ? 1u : 0u) << i;
        v[i] = keep ? v[i] * scale : 0.f;
    }
    mask[g] = (unsigned char)m;
    if (full) {
        *reinterpret_cast<float4*>(y + e0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(y + e0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (e0 + i < n) y[e0 + i] = v[i];
    }
}
__global__ __launch_bounds__(256) void dropout_bwd(int64_t n, const float* gy /* may be gx: in place */, const unsigned char* __restrict__ mask, float scale,
                                                   float* gx) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e0 = g * 8;
    if (e0 >= n) return;
    const unsigned m = mask[g];
    const bool full = e0 + 8 <= n && (reinterpret_cast<uintptr_t>(gy) & 15) == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0;
    if (full) {
        const float4 a = *reinterpret_cast<const float4*>(gy + e0), b = *reinterpret_cast<const float4*>(gy + e0 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (m >> i) & 1u ? v[i] * scale : 0.f;
        *reinterpret_cast<float4*>(gx + e0) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(gx + e0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (e0 + i < n) gx[e0 + i] = (m >> i) & 1u ? gy[e0 + i] * scale : 0.f;
    }
}

using namespace dgn;

extern "C" size_t dgn_dropout_mask_bytes(int64_t n_elems) { return n_elems > 0 ? (size_t)((n_elems + 7) / 8) : 0; }

extern "C" int dgn_dropout_forward(int64_t n_elems, const float* x, float p, const int64_t* seed, uint64_t offset, float* y,
                                   unsigned char* mask, void* stream_) {
    if (n_elems < 0 || !(p >= 0.f && p < 1.f)) { set_error("dgn_dropout_forward: need n_elems >= 0 and 0 <= p < 1"); return DGN_ERR_INVALID; }
    if (n_elems == 0) return DGN_OK;
    if (!x || !y || !mask || !seed) { set_error("dgn_dropout_forward: null buffer"); return DGN_ERR_INVALID; }
    const uint32_t threshold = (uint32_t)std::min<double>(4294967295.0, std::floor((double)p * 4294967296.0));
    const int64_t groups = (n_elems + 7) / 8;
    hipLaunchKernelGGL(dropout_fwd, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), n_elems, x, threshold,
                       1.f / (1.f - p), seed, offset, y, mask);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_dropout_backward(int64_t n_elems, const float* g_y, const unsigned char* mask, float p, float* g_x, void* stream_) {
    if (n_elems < 0 || !(p >= 0.f && p < 1.f)) { set_error("dgn_dropout_backward: need n_elems >= 0 and 0 <= p < 1"); return DGN_ERR_INVALID; }
    if (n_elems == 0) return DGN_OK;
    if (!g_y || !g_x || !mask) { set_error("dgn_dropout_backward: null buffer"); return DGN_ERR_INVALID; }
    const int64_t groups = (n_elems + 7) / 8;
    hipLaunchKernelGGL(dropout_bwd, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), n_elems, g_y, mask,
                       1.f / (1.f - p), g_x);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" size_t dgn_bn_tail_workspace_bytes(int64_t n_rows, int32_t F) {
    if (n_rows <= 0 || F < 1 || F > kMaxF) return 0;
    return part_bytes(n_rows, F) + (size_t)2 * F * sizeof(float);
}

extern "C" int dgn_bn_tail_forward(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                                   int32_t relu, const float* residual, float* y, float* save_mean, float* save_invstd, void* ws,
                                   size_t ws_bytes, const int64_t* n_valid, void* stream_) {
    return dgn::bn_tail_forward_nbt(n_rows, F, x, ld, gamma, beta, running_mean, running_var, momentum, eps, training, relu, residual, y, save_mean,
                                    save_invstd, ws, ws_bytes, n_valid, nullptr, 0, stream_);
}

int dgn::bn_finalize_launch(int64_t n_rows, int32_t F, int32_t G, const double* part, float* running_mean, float* running_var, float momentum, float eps,
                            float* save_mean, float* save_invstd, int64_t* nbt, int32_t n_nbt, void* stream_) {
    if (n_rows <= 0 || F < 1 || F > kMaxF || G < 1 || !part || !save_mean || !save_invstd || (nbt && (n_nbt < 0 || n_nbt > 256))) {
        set_error("bn_finalize_launch: bad argument");
        return DGN_ERR_INVALID;
    }
    hipLaunchKernelGGL(bn_finalize, dim3(F), dim3(256), 0, static_cast<hipStream_t>(stream_), n_rows, F, G, part, running_mean, running_var, momentum, eps,
                       save_mean, save_invstd, (const int64_t*)nullptr, nbt, nbt ? n_nbt : 0);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

int dgn::bn_tail_forward_from_partials(int64_t n_rows, int32_t F, int32_t G, const double* part, const float* x, int64_t ld, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var, float momentum, float eps, int32_t relu,
                                       const float* residual, float* y, float* save_mean, float* save_invstd, int64_t* nbt, int32_t n_nbt, void* stream_) {
    if (!x || !y || ld < F) { set_error("bn_tail_forward_from_partials: bad argument"); return DGN_ERR_INVALID; }
    if (int rc = bn_finalize_launch(n_rows, F, G, part, running_mean, running_var, momentum, eps, save_mean, save_invstd, nbt, n_nbt, stream_)) return rc;
    launch_bn_apply(static_cast<hipStream_t>(stream_), n_rows, F, x, ld, gamma, beta, save_mean, save_invstd, nullptr, nullptr, eps, relu, residual, y);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

// ... with the modules' num_batches_tracked counters incremented by the statistics' finalize kernel (training only; library-internal)
int dgn::bn_tail_forward_nbt(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, int32_t training, int32_t relu, const float* residual, float* y,
                             float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, const int64_t* n_valid, int64_t* nbt, int32_t n_nbt,
                             void* stream_) {
    if (nbt && (n_nbt < 0 || n_nbt > 256)) { set_error("dgn_bn_tail_forward: at most 256 num_batches_tracked counters"); return DGN_ERR_INVALID; }
    if (n_rows < 0 || F < 1 || F > kMaxF || ld < F) { set_error("dgn_bn_tail_forward: bad shape (need 1 <= F <= 1024, ld >= F)"); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!x || (!y && !training)) { set_error("dgn_bn_tail_forward: null buffer"); return DGN_ERR_INVALID; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (training) {
        if (!ws || !save_mean || !save_invstd) { set_error("dgn_bn_tail_forward: training needs ws, save_mean, save_invstd"); return DGN_ERR_INVALID; }
        if (ws_bytes < dgn_bn_tail_workspace_bytes(n_rows, F)) { set_error("dgn_bn_tail_forward: workspace too small"); return DGN_ERR_WORKSPACE; }
        double* part = static_cast<double*>(ws);
        const int G = stat_groups(n_rows, F);
        if (pairs_ok(F, ld, x)) hipLaunchKernelGGL(bn_stats<true>, dim3(G), dim3(256), 0, stream, n_rows, F, x, ld, part, n_valid);
        else hipLaunchKernelGGL(bn_stats<false>, dim3(G), dim3(256), 0, stream, n_rows, F, x, ld, part, n_valid);
        hipLaunchKernelGGL(bn_finalize, dim3(F), dim3(256), 0, stream, n_rows, F, G, (const double*)part, running_mean,
                           running_var, momentum, eps, save_mean, save_invstd, n_valid, nbt, nbt ? n_nbt : 0);
        // y == NULL: statistics only (the consumer normalises on the fly: dgn_linear_forward_bn / dgn_linear_wgrad_bn)
        if (y) launch_bn_apply(stream, n_rows, F, x, ld, gamma, beta, save_mean, save_invstd, nullptr, nullptr, eps, relu, residual, y);
    } else {
        if (!running_mean || !running_var) { set_error("dgn_bn_tail_forward: eval mode needs running statistics"); return DGN_ERR_INVALID; }
        launch_bn_apply(stream, n_rows, F, x, ld, gamma, beta, nullptr, nullptr, running_mean, running_var, eps, relu, residual, y);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_bn_tail_backward(int64_t n_rows, int32_t F, const float* g_y, const float* x, int64_t ld, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, int32_t relu, float* g_x,
                                    float* g_gamma, float* g_beta, float* sums_out, void* ws, size_t ws_bytes, const int64_t* n_valid,
                                    void* stream_) {
    if (n_rows < 0 || F < 1 || F > kMaxF || ld < F) { set_error("dgn_bn_tail_backward: bad shape (need 1 <= F <= 1024, ld >= F)"); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!g_y || !x || (!g_x && !sums_out) || !save_mean || !save_invstd || !ws) { set_error("dgn_bn_tail_backward: null buffer"); return DGN_ERR_INVALID; }
    if (ws_bytes < dgn_bn_tail_workspace_bytes(n_rows, F)) { set_error("dgn_bn_tail_backward: workspace too small"); return DGN_ERR_WORKSPACE; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* part = static_cast<double*>(ws);
    float* sums = sums_out ? sums_out : reinterpret_cast<float*>(static_cast<char*>(ws) + part_bytes(n_rows, F));
    const int G = stat_groups(n_rows, F);
    // (flat 16-byte chunks: odd widths of batches that fill the chip -- tools/bn_flat_vs_column.py: [275 k, 75] 47.0 -> 41.4 us, but [52 k, 70] 18.7 -> 23.0
    //  and [15 k, 65] 10.9 -> 14.1 against the thread-per-column(-pair) kernels, whose fold at the end of a workgroup is shorter)
    if ((F & 1) && n_rows >= 131072 && flat4_ok(F, ld, x, g_y)) hipLaunchKernelGGL(bn_bwd_stats_flat4, dim3(G), dim3(256), 0, stream, n_rows, F, g_y, x, gamma, beta, save_mean, save_invstd, relu, part, n_valid);
    else if (pairs_ok(F, ld, x, g_y)) hipLaunchKernelGGL(bn_bwd_stats<true>, dim3(G), dim3(256), 0, stream, n_rows, F, g_y, x, ld, gamma, beta, save_mean, save_invstd, relu, part, n_valid);
    else hipLaunchKernelGGL(bn_bwd_stats<false>, dim3(G), dim3(256), 0, stream, n_rows, F, g_y, x, ld, gamma, beta, save_mean, save_invstd, relu, part, n_valid);
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(F), dim3(256), 0, stream, F, G, (const double*)part, sums, g_gamma, g_beta);
    if (g_x)
        hipLaunchKernelGGL(bn_bwd_apply, dim3(flat_grid(n_rows * F)), dim3(256), 0, stream, n_rows, F, g_y, x, ld, gamma, beta, save_mean,
                           save_invstd, relu, (const float*)sums, g_x, n_valid);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_bias_act_forward(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* bias, int32_t act, float slope,
                                    const float* residual, float* y, void* stream) {
    if (n_rows < 0 || F < 1 || ld < F || act < 0 || act > 2) { set_error("dgn_bias_act_forward: bad shape or activation"); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!x || !y) { set_error("dgn_bias_act_forward: null buffer"); return DGN_ERR_INVALID; }
    auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
    if (ld == F && (F & 1) == 0 && al8(x) && al8(y) && al8(residual)) {
        const int64_t n_pairs = n_rows * F / 2;
        hipLaunchKernelGGL(bias_act_fwd_pairs, dim3(flat_grid(n_pairs)), dim3(256), 0, static_cast<hipStream_t>(stream), n_pairs, F,
                           reinterpret_cast<const float2*>(x), bias, act, slope, reinterpret_cast<const float2*>(residual),
                           reinterpret_cast<float2*>(y));
    } else {
        hipLaunchKernelGGL(bias_act_fwd, dim3(flat_grid(n_rows * F)), dim3(256), 0, static_cast<hipStream_t>(stream), n_rows, F, x, ld, bias,
                           act, slope, residual, y);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_bias_act_backward(int64_t n_rows, int32_t F, const float* g_y, const float* x, int64_t ld, const float* bias,
                                     int32_t act, float slope, float* g_x, float* g_bias, void* ws, size_t ws_bytes, void* stream_) {
    if (n_rows < 0 || F < 1 || F > kMaxF || ld < F || act < 0 || act > 2) { set_error("dgn_bias_act_backward: bad shape (need 1 <= F <= 1024) or activation"); return DGN_ERR_INVALID; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows == 0) {
        if (g_bias && zero_rows_async(g_bias, 1, F, F, stream)) return DGN_ERR_HIP;
        return DGN_OK;
    }
    if (!g_y || !x || !g_x || !ws) { set_error("dgn_bias_act_backward: null buffer"); return DGN_ERR_INVALID; }
    if (ws_bytes < dgn_bn_tail_workspace_bytes(n_rows, F)) { set_error("dgn_bias_act_backward: workspace too small (dgn_bn_tail_workspace_bytes)"); return DGN_ERR_WORKSPACE; }
    double* part = static_cast<double*>(ws);
    const int G = stat_groups(n_rows, F);
    if (pairs_ok(F, ld, x, g_y, g_x)) hipLaunchKernelGGL(bias_act_bwd<true>, dim3(G), dim3(256), 0, stream, n_rows, F, g_y, x, ld, bias, act, slope, g_x, part);
    else hipLaunchKernelGGL(bias_act_bwd<false>, dim3(G), dim3(256), 0, stream, n_rows, F, g_y, x, ld, bias, act, slope, g_x, part);
    if (g_bias) hipLaunchKernelGGL(bias_act_finalize, dim3(F), dim3(256), 0, stream, F, G, (const double*)part, g_bias);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
