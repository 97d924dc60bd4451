// Scale-combine epilogue of the post-aggregation Linear (see include/dgn_hip.h, dgn_scale_combine_*):
// pure streaming kernels, one thread per output element.
#include <hip/hip_runtime.h>

#include "dgn_common.hpp"

namespace dgn {
namespace {

__global__ __launch_bounds__(256) void combine_fwd(int64_t n_nodes, int T, int S, int fo, const float* __restrict__ z,
                                                   const float* __restrict__ scale, const float* __restrict__ bias,
                                                   const float* __restrict__ row_scale, float* __restrict__ y, int64_t ld_y) {
    const int width = T * fo;
    const int64_t total = n_nodes * width;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / width;
        const int c = (int)(idx - n * width);
        const int t = c / fo, o = c - t * fo;
        const float* zr = z + ((int64_t)t * n_nodes + n) * ((int64_t)S * fo) + o;
        float acc = bias ? bias[c] : 0.f;
        if (scale) {
            for (int s = 0; s < S; ++s) acc += scale[n * S + s] * zr[(int64_t)s * fo];
        } else {
            acc += zr[0];
        }
        if (row_scale) acc *= row_scale[n];
        y[n * ld_y + c] = acc;
    }
}

__global__ __launch_bounds__(256) void combine_bwd(int64_t n_nodes, int T, int S, int fo, const float* __restrict__ gy,
                                                   int64_t ld_gy, const float* __restrict__ scale,
                                                   const float* __restrict__ row_scale, float* __restrict__ gz) {
    const int zw = S * fo;
    const int64_t total = (int64_t)T * n_nodes * zw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tn = idx / zw;
        const int so = (int)(idx - tn * zw);
        const int s = so / fo, o = so - s * fo;
        const int64_t t = tn / n_nodes, n = tn - t * n_nodes;
        float g = gy[n * ld_gy + t * fo + o];
        if (row_scale) g *= row_scale[n];
        if (scale) g *= scale[n * S + s];
        gz[idx] = g;
    }
}

// bias gradient: column sums of row_scale[n] * gy[n, :]; a block owns a slab of rows, threads own columns
// (coalesced row reads), one atomic per (block, column)
__global__ __launch_bounds__(256) void combine_bias_grad(int64_t n_nodes, int width, const float* __restrict__ gy, int64_t ld_gy,
                                                         const float* __restrict__ row_scale, float* __restrict__ g_bias,
                                                         int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, n_nodes);
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float acc = 0.f;
        for (int64_t n = r0; n < r1; ++n) acc += (row_scale ? row_scale[n] : 1.f) * gy[n * ld_gy + c];
        unsafeAtomicAdd(g_bias + c, acc);
    }
}

unsigned grid_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32); }

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" int dgn_scale_combine_forward(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* z, const float* scale,
                                         const float* bias, const float* row_scale, float* y, int64_t ld_y, void* stream) {
    if (n_nodes < 0 || T < 1 || S < 1 || fo < 1 || (!scale && S != 1)) { set_error("dgn_scale_combine_forward: bad shape"); return DGN_ERR_INVALID; }
    if (n_nodes == 0) return DGN_OK;
    if (!z || !y || ld_y < (int64_t)T * fo) { set_error("dgn_scale_combine_forward: null buffer or ld_y too small"); return DGN_ERR_INVALID; }
    hipLaunchKernelGGL(combine_fwd, dim3(grid_for(n_nodes * T * fo)), dim3(256), 0, static_cast<hipStream_t>(stream), n_nodes, T, S, fo,
                       z, scale, bias, row_scale, y, ld_y);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_scale_combine_backward(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* g_y, int64_t ld_gy,
                                          const float* scale, const float* row_scale, float* g_z, float* g_bias, void* stream) {
    if (n_nodes < 0 || T < 1 || S < 1 || fo < 1 || (!scale && S != 1)) { set_error("dgn_scale_combine_backward: bad shape"); return DGN_ERR_INVALID; }
    if (n_nodes == 0) return DGN_OK;
    if (!g_y || !g_z || ld_gy < (int64_t)T * fo) { set_error("dgn_scale_combine_backward: null buffer or ld_gy too small"); return DGN_ERR_INVALID; }
    hipLaunchKernelGGL(combine_bwd, dim3(grid_for((int64_t)T * n_nodes * S * fo)), dim3(256), 0, static_cast<hipStream_t>(stream), n_nodes,
                       T, S, fo, g_y, ld_gy, scale, row_scale, g_z);
    if (g_bias) {
        const int rows_per_block = 128;
        hipLaunchKernelGGL(combine_bias_grad, dim3((unsigned)((n_nodes + rows_per_block - 1) / rows_per_block)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), n_nodes, T * fo, g_y, ld_gy, row_scale, g_bias, rows_per_block);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
