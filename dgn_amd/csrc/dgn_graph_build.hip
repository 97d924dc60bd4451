// Graph batch preparation on the device (include/dgn_hip.h: dgn_graph_build*): the edge list a (batched) DGL graph hands
// over -- src / dst in edge-id order -- becomes the CSR-by-destination the sweep reads, its transposed view for the atomic-free
// backward and the row windows of the window-local scatter.  The reference lets DGL bucket the destinations by in-degree on
// EVERY update_all call (realworld_benchmark/nets/dgn_layer.py:115,186,264; dgl.batch in data/molecules.py:229); here a batch
// is prepared once, by a handful of kernels enqueued on the caller's stream with no host synchronisation inside (the caller
// reads three integers back when it needs them).  Sorting and scans: hipCUB (rocPRIM) device primitives.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>

#include "dgn_common.hpp"

namespace dgn {
namespace {

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline unsigned blocks(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }
inline int bits_for(int64_t n) { int b = 1; while (((int64_t)1 << b) < n && b < 31) ++b; return b; }

__global__ void gb_prepare(int64_t E, const int64_t* __restrict__ dst, int32_t* __restrict__ key, int32_t* __restrict__ val,
                           int32_t* __restrict__ deg) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int d = (int)dst[e];
    key[e] = d;
    val[e] = (int)e;
    atomicAdd(deg + d, 1);
}
__global__ void gb_gather(int64_t E, const int64_t* __restrict__ src, const int32_t* __restrict__ perm, int32_t* __restrict__ src_csr,
                          int64_t* __restrict__ eid) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const int e = perm[k];
    src_csr[k] = (int)src[e];
    eid[k] = e;
}
__global__ void gb_node_stats(int64_t N, const int32_t* __restrict__ deg, float* __restrict__ log_deg, int64_t* __restrict__ deg64,
                              int32_t* __restrict__ stats, int32_t hub_threshold) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int d = 0;
    if (i < N) {
        d = deg[i];
        log_deg[i] = (float)log((double)(d + 1));       // scalers.py:13: np.log of a python int, in double, then fp32
        if (deg64) deg64[i] = d;
    }
    // block maximum first: one atomic per wave instead of one per node
    int m = d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(stats + 0, m);
    if (i < N && d > hub_threshold) atomicAdd(stats + 1, 1);
}
__global__ void gb_iota_hist(int64_t E, const int32_t* __restrict__ key, int32_t* __restrict__ val, int32_t* __restrict__ hist) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    val[e] = (int)e;
    atomicAdd(hist + key[e], 1);
}
__global__ void gb_invert(int64_t E, const int32_t* __restrict__ order, int32_t* __restrict__ pos) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E) pos[order[k]] = (int)k;
}

// ---- windows ----
__global__ void gw_span(int64_t E, const int32_t* __restrict__ src_csr, const int32_t* __restrict__ dst_csr, int32_t* __restrict__ span) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E) return;
    const int s = src_csr[j], d = dst_csr[j];
    if (s == d) return;
    atomicAdd(span + min(s, d), 1);       // the edge spans the cuts after rows min .. max - 1
    atomicAdd(span + max(s, d), -1);
}
__global__ void gw_last_closed(int64_t N, const int32_t* __restrict__ prefix, int32_t* __restrict__ last_closed, int R0) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N && prefix[r] == 0) atomicMax(last_closed + r / R0, (int)r);
}
__global__ void gw_cuts(int64_t nb, int64_t N, const int32_t* __restrict__ last_closed, int32_t* __restrict__ cut1, int R0) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const bool has = last_closed[b] >= 0, prev_has = b == 0 || last_closed[b - 1] >= 0;
    const int bin_end = (int)min((b + 1) * (int64_t)R0 - 1, N - 1);
    int cut = has ? last_closed[b] : (!prev_has ? bin_end : -1);
    if (b == nb - 1) cut = (int)N - 1;
    cut1[b] = cut + 1;
}
__global__ void gw_info(int64_t nb, const int32_t* __restrict__ win_ptr, const int32_t* __restrict__ indptr, const int32_t* __restrict__ csc_ptr,
                        int32_t* __restrict__ info, int32_t* __restrict__ win_of_row) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int r0 = win_ptr[b], r1 = win_ptr[b + 1];
    int32_t* o = info + b * 8;
    o[0] = r0; o[1] = r1 - r0; o[2] = indptr[r0]; o[3] = indptr[r1]; o[4] = csc_ptr[r0]; o[5] = csc_ptr[r1]; o[6] = 0; o[7] = 0;
    for (int r = r0; r < r1; ++r) win_of_row[r] = (int)b;
}
__global__ void gw_remote(int64_t E, const int32_t* __restrict__ order, const int32_t* __restrict__ src_csr, const int32_t* __restrict__ dst_csr,
                          const int32_t* __restrict__ win_of_row, const int32_t* __restrict__ win_ptr, const int32_t* __restrict__ csc_ptr,
                          int ecap, uint8_t* __restrict__ flag, int32_t* __restrict__ rem_cnt, int32_t* __restrict__ stats) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const int j = order[k];
    const int u = src_csr[j], wu = win_of_row[u];
    const bool local = win_of_row[dst_csr[j]] == wu && (int)k - csc_ptr[win_ptr[wu]] < ecap;
    flag[k] = local ? 0 : 1;
    if (!local) {
        atomicAdd(rem_cnt + u, 1);
        atomicAdd(stats + 2, 1);
    }
}
__global__ void gb_iota(int64_t n, int32_t* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int)i;
}

struct MaxOp {
    __host__ __device__ int operator()(int a, int b) const { return a > b ? a : b; }
};

// temp storage the hipCUB calls of this file need, for n items
size_t cub_bytes(int64_t n) {
    size_t a = 0, b = 0, c = 0, d = 0;
    int32_t* p = nullptr;
    uint8_t* f = nullptr;
    hipcub::DeviceRadixSort::SortPairs(nullptr, a, p, p, p, p, (int)n);
    hipcub::DeviceScan::ExclusiveSum(nullptr, b, p, p, (int)n);
    hipcub::DeviceScan::InclusiveScan(nullptr, c, p, p, MaxOp(), (int)n);
    hipcub::DeviceSelect::Flagged(nullptr, d, p, f, p, p, (int)n);
    return up256(std::max(std::max(a, b), std::max(c, d))) + 256;
}

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" size_t dgn_graph_build_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    const int64_t m = (n_nodes + 2 > n_edges ? n_nodes + 2 : n_edges) + 1;
    // keys in / values in / values out / flags + scratch per node, + the primitives' own storage
    return cub_bytes(m) + 4 * up256((size_t)(n_edges + 1) * 4) + 4 * up256((size_t)(n_nodes + 2) * 4) + up256((size_t)n_edges + 1);
}

extern "C" int dgn_graph_build(int64_t n_nodes, int64_t n_edges, const int64_t* src, const int64_t* dst, int32_t* indptr,
                               int32_t* src_csr, int32_t* dst_csr, int64_t* eid, float* log_deg, int64_t* in_degree, int32_t* stats,
                               int32_t hub_threshold, void* ws, size_t ws_bytes, void* stream_) {
    const char* fn = "dgn_graph_build";
    if (n_nodes < 0 || n_edges < 0 || n_nodes >= INT32_MAX - 1 || n_edges >= INT32_MAX - 1) { set_error("%s: sizes outside the int32 CSR range", fn); return DGN_ERR_INVALID; }
    if (!indptr || !log_deg || !stats || (n_edges > 0 && (!src || !dst || !src_csr || !dst_csr || !eid))) { set_error("%s: null array", fn); return DGN_ERR_INVALID; }
    if (!ws || ws_bytes < dgn_graph_build_workspace_bytes(n_nodes, n_edges)) { set_error("%s: workspace too small", fn); return DGN_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    char* w = static_cast<char*>(ws);
    const size_t eb = up256((size_t)(n_edges + 1) * 4), nbytes = up256((size_t)(n_nodes + 2) * 4);
    int32_t* key = reinterpret_cast<int32_t*>(w); w += eb;
    int32_t* val = reinterpret_cast<int32_t*>(w); w += eb;
    int32_t* perm = reinterpret_cast<int32_t*>(w); w += eb;
    w += eb;
    int32_t* deg = reinterpret_cast<int32_t*>(w); w += nbytes;
    w += 3 * nbytes;
    w += up256((size_t)n_edges + 1);
    void* cub = w;
    size_t cub_sz = ws_bytes - (size_t)(w - static_cast<char*>(ws));
    DGN_HIP_CHECK(hipMemsetAsync(deg, 0, (size_t)(n_nodes + 1) * 4, st));
    DGN_HIP_CHECK(hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), st));
    if (n_edges > 0) hipLaunchKernelGGL(gb_prepare, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, dst, key, val, deg);
    DGN_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(cub, cub_sz, deg, indptr, (int)(n_nodes + 1), st));
    if (n_edges > 0) {
        // stable: the slots of a destination keep ascending edge id (the mailbox order the oracle encodes)
        DGN_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(cub, cub_sz, key, dst_csr, val, perm, (int)n_edges, 0, bits_for(n_nodes), st));
        hipLaunchKernelGGL(gb_gather, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, src, perm, src_csr, eid);
    }
    if (n_nodes > 0) hipLaunchKernelGGL(gb_node_stats, dim3(blocks(n_nodes)), dim3(256), 0, st, n_nodes, deg, log_deg, in_degree, stats, hub_threshold);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_graph_build_csc(int64_t n_nodes, int64_t n_edges, const int32_t* src_csr, int32_t* csc_ptr, int32_t* csc_pos,
                                   int32_t* csc_order, void* ws, size_t ws_bytes, void* stream_) {
    const char* fn = "dgn_graph_build_csc";
    if (!csc_ptr || (n_edges > 0 && (!src_csr || !csc_pos || !csc_order))) { set_error("%s: null array", fn); return DGN_ERR_INVALID; }
    if (!ws || ws_bytes < dgn_graph_build_workspace_bytes(n_nodes, n_edges)) { set_error("%s: workspace too small", fn); return DGN_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    char* w = static_cast<char*>(ws);
    const size_t eb = up256((size_t)(n_edges + 1) * 4), nbytes = up256((size_t)(n_nodes + 2) * 4);
    int32_t* val = reinterpret_cast<int32_t*>(w); w += eb;
    int32_t* key_out = reinterpret_cast<int32_t*>(w); w += eb;
    w += 2 * eb;
    int32_t* hist = reinterpret_cast<int32_t*>(w); w += nbytes;
    w += 3 * nbytes;
    w += up256((size_t)n_edges + 1);
    void* cub = w;
    size_t cub_sz = ws_bytes - (size_t)(w - static_cast<char*>(ws));
    DGN_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)(n_nodes + 1) * 4, st));
    if (n_edges > 0) hipLaunchKernelGGL(gb_iota_hist, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, src_csr, val, hist);
    DGN_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(cub, cub_sz, hist, csc_ptr, (int)(n_nodes + 1), st));
    if (n_edges > 0) {
        DGN_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(cub, cub_sz, src_csr, key_out, val, csc_order, (int)n_edges, 0, bits_for(n_nodes), st));
        hipLaunchKernelGGL(gb_invert, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, csc_order, csc_pos);
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_graph_build_windows(int64_t n_nodes, int64_t n_edges, const int32_t* indptr, const int32_t* src_csr,
                                       const int32_t* dst_csr, const int32_t* csc_ptr, const int32_t* csc_order, int32_t bin_rows,
                                       int32_t ecap, int32_t* win_ptr, int32_t* win_info, int32_t* rem_ptr, int32_t* rem_idx,
                                       int32_t* stats, void* ws, size_t ws_bytes, void* stream_) {
    const char* fn = "dgn_graph_build_windows";
    if (n_nodes < 1 || bin_rows < 1 || ecap < 1 || !indptr || !csc_ptr || !win_ptr || !win_info || !rem_ptr || !rem_idx || !stats ||
        (n_edges > 0 && (!src_csr || !dst_csr || !csc_order))) { set_error("%s: bad argument", fn); return DGN_ERR_INVALID; }
    if (!ws || ws_bytes < dgn_graph_build_workspace_bytes(n_nodes, n_edges)) { set_error("%s: workspace too small", fn); return DGN_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int64_t nb = (n_nodes + bin_rows - 1) / bin_rows;
    char* w = static_cast<char*>(ws);
    const size_t eb = up256((size_t)(n_edges + 1) * 4), nbytes = up256((size_t)(n_nodes + 2) * 4);
    int32_t* iota = reinterpret_cast<int32_t*>(w); w += eb;
    w += 3 * eb;
    int32_t* span = reinterpret_cast<int32_t*>(w); w += nbytes;
    int32_t* prefix = reinterpret_cast<int32_t*>(w); w += nbytes;
    int32_t* binbuf = reinterpret_cast<int32_t*>(w); w += nbytes;          // last_closed, then cut + 1   (nb <= n_nodes)
    int32_t* win_of_row = reinterpret_cast<int32_t*>(w); w += nbytes;
    uint8_t* flag = reinterpret_cast<uint8_t*>(w); w += up256((size_t)n_edges + 1);
    void* cub = w;
    size_t cub_sz = ws_bytes - (size_t)(w - static_cast<char*>(ws));
    int32_t* rem_cnt = span;                                                    // (span is dead once `prefix` exists)
    DGN_HIP_CHECK(hipMemsetAsync(span, 0, (size_t)(n_nodes + 1) * 4, st));
    DGN_HIP_CHECK(hipMemsetAsync(binbuf, 0xff, (size_t)nb * 4, st));           // -1
    DGN_HIP_CHECK(hipMemsetAsync(stats + 2, 0, sizeof(int32_t), st));
    if (n_edges > 0) hipLaunchKernelGGL(gw_span, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, src_csr, dst_csr, span);
    DGN_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(cub, cub_sz, span, prefix, (int)n_nodes, st));
    hipLaunchKernelGGL(gw_last_closed, dim3(blocks(n_nodes)), dim3(256), 0, st, n_nodes, prefix, binbuf, bin_rows);
    int32_t* cut1 = prefix;                                                     // (prefix is dead after gw_last_closed) -- different stream order: same stream, safe
    hipLaunchKernelGGL(gw_cuts, dim3(blocks(nb)), dim3(256), 0, st, nb, n_nodes, binbuf, cut1, bin_rows);
    DGN_HIP_CHECK(hipMemsetAsync(win_ptr, 0, sizeof(int32_t), st));
    DGN_HIP_CHECK(hipcub::DeviceScan::InclusiveScan(cub, cub_sz, cut1, win_ptr + 1, MaxOp(), (int)nb, st));
    hipLaunchKernelGGL(gw_info, dim3(blocks(nb)), dim3(256), 0, st, nb, win_ptr, indptr, csc_ptr, win_info, win_of_row);
    DGN_HIP_CHECK(hipMemsetAsync(rem_cnt, 0, (size_t)(n_nodes + 1) * 4, st));
    if (n_edges > 0) {
        hipLaunchKernelGGL(gw_remote, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, csc_order, src_csr, dst_csr, win_of_row, win_ptr, csc_ptr,
                           ecap, flag, rem_cnt, stats);
        hipLaunchKernelGGL(gb_iota, dim3(blocks(n_edges)), dim3(256), 0, st, n_edges, iota);
    }
    DGN_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(cub, cub_sz, rem_cnt, rem_ptr, (int)(n_nodes + 1), st));
    if (n_edges > 0)
        DGN_HIP_CHECK(hipcub::DeviceSelect::Flagged(cub, cub_sz, iota, flag, rem_idx, stats + 3, (int)n_edges, st));
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
