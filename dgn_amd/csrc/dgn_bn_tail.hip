// Layer tail for gfx950: BatchNorm1d over the node dimension (+ ReLU + residual), forward and backward
// (reference: realworld_benchmark/nets/dgn_layer.py:123-128, :194-199, :272-273 -- nn.BatchNorm1d, F.relu, h_in + h).
// Streaming kernels over [N, F] with F <= 1024: a workgroup owns a slab of rows, threads own columns, so every
// row read is coalesced; column statistics are slab partials accumulated in fp64 (one pass, no fp32
// cancellation in E[x^2] - mean^2) and combined with fp64 atomics.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "dgn_common.hpp"

namespace dgn {
namespace {

constexpr int kRowsPerBlock = 256;

// Column sums of up to two quantities over a slab of rows.  Threads are laid out (row phase, column): with
// F = 70 a 256-thread block works on 3 rows at a time, all lanes busy, every row read coalesced; the phases are
// folded through LDS and one double-precision atomic per (block, column, quantity) goes to the workspace.
// Sums are kept in fp64 (E[x^2] - mean^2 in one pass without the fp32 cancellation).
template <class Fn>
__device__ __forceinline__ void slab_sums(int64_t n_rows, int F, double* __restrict__ ws, Fn&& fn) {
    __shared__ double red[2][256];
    const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock, r1 = min(r0 + kRowsPerBlock, n_rows);
    const int P = max(1, 256 / F);                 // row phases per block
    for (int c0 = 0; c0 < F; c0 += 256) {          // (one iteration unless F > 256)
        const int p = (int)threadIdx.x / F, c = c0 + (int)threadIdx.x % F;
        double a0 = 0.0, a1 = 0.0;
        if (p < P && c < F) {
            for (int64_t n = r0 + p; n < r1; n += P) {
                float v0, v1;
                fn(n, c, v0, v1);
                a0 += (double)v0;
                a1 += (double)v1;
            }
        }
        red[0][threadIdx.x] = a0;
        red[1][threadIdx.x] = a1;
        __syncthreads();
        if (p == 0 && c < F) {
            for (int q = 1; q < P; ++q) { a0 += red[0][threadIdx.x + q * F]; a1 += red[1][threadIdx.x + q * F]; }
            atomicAdd(ws + c, a0);
            atomicAdd(ws + F + c, a1);
        }
        __syncthreads();
    }
}

// forward statistics: ws[c] = sum x, ws[F + c] = sum x^2
__global__ __launch_bounds__(256) void bn_stats(int64_t n_rows, int F, const float* __restrict__ x, int64_t ld, double* __restrict__ ws) {
    slab_sums(n_rows, F, ws, [&](int64_t n, int c, float& v0, float& v1) {
        const float v = x[n * ld + c];
        v0 = v;
        v1 = v * v;
    });
}

__device__ __forceinline__ void col_moments(const double* ws, int F, int c, int64_t n_rows, float eps, float& mean, float& invstd, double& m2) {
    const double mu = ws[c] / (double)n_rows;
    m2 = ws[F + c] - mu * ws[c];                   // sum (x - mean)^2
    if (m2 < 0.0) m2 = 0.0;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(m2 / (double)n_rows + (double)eps));
}

// normalise (+ReLU, +residual).  Training: statistics from ws; block 0 also publishes save_mean / save_invstd and
// updates the running statistics.  Eval (ws == NULL): running statistics.
__global__ __launch_bounds__(256) void bn_apply(int64_t n_rows, int F, const float* __restrict__ x, int64_t ld,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                const double* __restrict__ ws, float* running_mean, float* running_var,
                                                float momentum, float eps, int relu, const float* __restrict__ residual,
                                                float* __restrict__ y, float* save_mean, float* save_invstd) {
    if (ws && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < F; c += blockDim.x) {
            float mean, invstd;
            double m2;
            col_moments(ws, F, c, n_rows, eps, mean, invstd, m2);
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const float unbiased = (float)(n_rows > 1 ? m2 / (double)(n_rows - 1) : m2 / (double)n_rows);
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    }
    const int64_t total = n_rows * F;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / F;
        const int c = (int)(idx - n * F);
        float mean, invstd;
        if (ws) {
            double m2;
            col_moments(ws, F, c, n_rows, eps, mean, invstd, m2);
        } else {
            mean = running_mean[c];
            invstd = 1.f / sqrtf(running_var[c] + eps);
        }
        float v = (x[n * ld + c] - mean) * invstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        if (residual) v += residual[n * ld + c];
        y[n * ld + c] = v;
    }
}

// backward statistics: ws[c] = sum g', ws[F + c] = sum g' * xhat      (g' = g masked by the ReLU)
__global__ __launch_bounds__(256) void bn_bwd_stats(int64_t n_rows, int F, const float* __restrict__ gy, const float* __restrict__ x,
                                                    int64_t ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                    double* __restrict__ ws) {
    slab_sums(n_rows, F, ws, [&](int64_t n, int c, float& v0, float& v1) {
        const float xh = (x[n * ld + c] - mean[c]) * invstd[c];
        float g = gy[n * ld + c];
        if (relu && !(xh * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f) > 0.f)) g = 0.f;
        v0 = g;
        v1 = g * xh;
    });
}

// d x; block 0 also publishes d gamma = sum g' xhat and d beta = sum g'
__global__ __launch_bounds__(256) void bn_bwd_apply(int64_t n_rows, int F, const float* __restrict__ gy, const float* __restrict__ x,
                                                    int64_t ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                    const double* __restrict__ ws, float* __restrict__ gx, float* g_gamma, float* g_beta) {
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < F; c += blockDim.x) {
            if (g_beta) g_beta[c] = (float)ws[c];
            if (g_gamma) g_gamma[c] = (float)ws[F + c];
        }
    }
    const int64_t total = n_rows * F;
    const float inv_n = 1.f / (float)n_rows;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / F;
        const int c = (int)(idx - n * F);
        const float is = invstd[c], ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        const float xh = (x[n * ld + c] - mean[c]) * is;
        float g = gy[n * ld + c];
        if (relu && !(xh * ga + be > 0.f)) g = 0.f;
        gx[n * ld + c] = ga * is * (g - (float)ws[c] * inv_n - xh * (float)ws[F + c] * inv_n);
    }
}

unsigned slabs(int64_t n_rows) { return (unsigned)((n_rows + kRowsPerBlock - 1) / kRowsPerBlock); }
unsigned flat_grid(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32); }

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" int dgn_bn_tail_forward(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                                   int32_t relu, const float* residual, float* y, float* save_mean, float* save_invstd, void* ws,
                                   void* stream_) {
    if (n_rows < 0 || F < 1 || ld < F) { set_error("dgn_bn_tail_forward: bad shape"); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!x || !y) { set_error("dgn_bn_tail_forward: null buffer"); return DGN_ERR_INVALID; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (training) {
        if (!ws || !save_mean || !save_invstd) { set_error("dgn_bn_tail_forward: training needs ws, save_mean, save_invstd"); return DGN_ERR_INVALID; }
        double* w = static_cast<double*>(ws);
        DGN_HIP_CHECK(hipMemsetAsync(w, 0, 2 * (size_t)F * sizeof(double), stream));
        hipLaunchKernelGGL(bn_stats, dim3(slabs(n_rows)), dim3(256), 0, stream, n_rows, F, x, ld, w);
        hipLaunchKernelGGL(bn_apply, dim3(flat_grid(n_rows * F)), dim3(256), 0, stream, n_rows, F, x, ld, gamma, beta, (const double*)w,
                           running_mean, running_var, momentum, eps, relu, residual, y, save_mean, save_invstd);
    } else {
        if (!running_mean || !running_var) { set_error("dgn_bn_tail_forward: eval mode needs running statistics"); return DGN_ERR_INVALID; }
        hipLaunchKernelGGL(bn_apply, dim3(flat_grid(n_rows * F)), dim3(256), 0, stream, n_rows, F, x, ld, gamma, beta, (const double*)nullptr,
                           running_mean, running_var, momentum, eps, relu, residual, y, (float*)nullptr, (float*)nullptr);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_bn_tail_backward(int64_t n_rows, int32_t F, const float* g_y, const float* x, int64_t ld, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, int32_t relu, float* g_x,
                                    float* g_gamma, float* g_beta, void* ws, void* stream_) {
    if (n_rows < 0 || F < 1 || ld < F) { set_error("dgn_bn_tail_backward: bad shape"); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!g_y || !x || !g_x || !save_mean || !save_invstd || !ws) { set_error("dgn_bn_tail_backward: null buffer"); return DGN_ERR_INVALID; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* w = static_cast<double*>(ws);
    DGN_HIP_CHECK(hipMemsetAsync(w, 0, 2 * (size_t)F * sizeof(double), stream));
    hipLaunchKernelGGL(bn_bwd_stats, dim3(slabs(n_rows)), dim3(256), 0, stream, n_rows, F, g_y, x, ld, gamma, beta, save_mean, save_invstd,
                       relu, w);
    hipLaunchKernelGGL(bn_bwd_apply, dim3(flat_grid(n_rows * F)), dim3(256), 0, stream, n_rows, F, g_y, x, ld, gamma, beta, save_mean,
                       save_invstd, relu, (const double*)w, g_x, g_gamma, g_beta);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
