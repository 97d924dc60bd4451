// Scale-combine epilogue of the post-aggregation Linear (see include/dgn_hip.h, dgn_scale_combine_*):
// pure streaming kernels, one thread per output element.
#include <hip/hip_runtime.h>

#include "dgn_common.hpp"

namespace dgn {
namespace {

// Slab layout (no per-element integer divisions): a workgroup owns kRows rows, a thread owns one output column
// (and a row phase when the row is narrower than the workgroup), so every row access is coalesced.
constexpr int kRows = 64;

__global__ __launch_bounds__(256) void combine_fwd(int64_t n_nodes, int T, int S, int fo, const float* __restrict__ z,
                                                   const float* __restrict__ scale, const float* __restrict__ bias,
                                                   const float* __restrict__ row_scale, float* __restrict__ y, int64_t ld_y) {
    const int width = T * fo;
    const int P = max(1, 256 / width);
    const int64_t r0 = (int64_t)blockIdx.x * kRows, r1 = min(r0 + kRows, n_nodes);
    for (int c0 = 0; c0 < width; c0 += 256) {
        const int p = (int)threadIdx.x / width, c = c0 + (int)threadIdx.x % width;
        if (p >= P || c >= width) continue;
        const int t = c / fo, o = c - t * fo;
        const float b = bias ? bias[c] : 0.f;
        for (int64_t n = r0 + p; n < r1; n += P) {
            const float* zr = z + ((int64_t)t * n_nodes + n) * ((int64_t)S * fo) + o;
            float acc = b;
            if (scale) {
                for (int s = 0; s < S; ++s) acc += scale[n * S + s] * zr[s * fo];
            } else {
                acc += zr[0];
            }
            if (row_scale) acc *= row_scale[n];
            y[n * ld_y + c] = acc;
        }
    }
}

// g_z[t][n][s*fo+o] = row_scale[n] * scale[n,s] * g_y[n, t*fo+o]; threads with s == 0 also accumulate the bias
// gradient sum_n row_scale[n] * g_y[n, t*fo+o] (one atomic per (block, column))
__global__ __launch_bounds__(256) void combine_bwd(int64_t n_nodes, int T, int S, int fo, const float* __restrict__ gy,
                                                   int64_t ld_gy, const float* __restrict__ scale,
                                                   const float* __restrict__ row_scale, float* __restrict__ gz,
                                                   float* __restrict__ g_bias) {
    const int zw = S * fo, width = T * zw;
    const int P = max(1, 256 / width);
    const int64_t r0 = (int64_t)blockIdx.x * kRows, r1 = min(r0 + kRows, n_nodes);
    for (int c0 = 0; c0 < width; c0 += 256) {
        const int p = (int)threadIdx.x / width, c = c0 + (int)threadIdx.x % width;
        if (p >= P || c >= width) continue;
        const int t = c / zw, so = c - t * zw, s = so / fo, o = so - s * fo;
        float bsum = 0.f;
        for (int64_t n = r0 + p; n < r1; n += P) {
            float g = gy[n * ld_gy + t * fo + o];
            if (row_scale) g *= row_scale[n];
            if (s == 0) bsum += g;
            if (scale) g *= scale[n * S + s];
            gz[((int64_t)t * n_nodes + n) * zw + so] = g;
        }
        if (g_bias && s == 0) unsafeAtomicAdd(g_bias + t * fo + o, bsum);
    }
}

unsigned row_blocks(int64_t n) { return (unsigned)((n + kRows - 1) / kRows); }

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" int dgn_scale_combine_forward(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* z, const float* scale,
                                         const float* bias, const float* row_scale, float* y, int64_t ld_y, void* stream) {
    if (n_nodes < 0 || T < 1 || S < 1 || fo < 1 || (!scale && S != 1)) { set_error("dgn_scale_combine_forward: bad shape"); return DGN_ERR_INVALID; }
    if (n_nodes == 0) return DGN_OK;
    if (!z || !y || ld_y < (int64_t)T * fo) { set_error("dgn_scale_combine_forward: null buffer or ld_y too small"); return DGN_ERR_INVALID; }
    hipLaunchKernelGGL(combine_fwd, dim3(row_blocks(n_nodes)), dim3(256), 0, static_cast<hipStream_t>(stream), n_nodes, T, S, fo,
                       z, scale, bias, row_scale, y, ld_y);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_scale_combine_backward(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* g_y, int64_t ld_gy,
                                          const float* scale, const float* row_scale, float* g_z, float* g_bias, void* stream) {
    if (n_nodes < 0 || T < 1 || S < 1 || fo < 1 || (!scale && S != 1)) { set_error("dgn_scale_combine_backward: bad shape"); return DGN_ERR_INVALID; }
    if (n_nodes == 0) return DGN_OK;
    if (!g_y || !g_z || ld_gy < (int64_t)T * fo) { set_error("dgn_scale_combine_backward: null buffer or ld_gy too small"); return DGN_ERR_INVALID; }
    hipLaunchKernelGGL(combine_bwd, dim3(row_blocks(n_nodes)), dim3(256), 0, static_cast<hipStream_t>(stream), n_nodes, T, S, fo, g_y, ld_gy,
                       scale, row_scale, g_z, g_bias);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
