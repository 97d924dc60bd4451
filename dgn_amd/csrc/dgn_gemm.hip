// Host side of the wide tall-skinny GEMMs (include/dgn_hip.h: dgn_gemm_*), kernels in dgn_gemm_kernels.hpp.
#include "dgn_gemm_kernels.hpp"

#include <algorithm>
#include <cstdlib>

namespace dgn {
namespace gemm {
namespace {

int n_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

template <int NT>
hipError_t launch_gemm_nt(const GemmParams& p, dim3 grid, hipStream_t st, int wkn) {
    const size_t lds = (size_t)2 * NT * 16 * kKS * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ts_gemm<NT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ts_gemm<NT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (wkn) hipLaunchKernelGGL((ts_gemm<NT, 1>), grid, dim3(kWave * kWaves), lds, st, p);
    else hipLaunchKernelGGL((ts_gemm<NT, 0>), grid, dim3(kWave * kWaves), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_gemm(int nt, const GemmParams& p, dim3 grid, hipStream_t st, int wkn) {
    switch (nt) {
#define DGN_CASE(N) case N: return launch_gemm_nt<N>(p, grid, st, wkn);
        DGN_CASE(1) DGN_CASE(2) DGN_CASE(3) DGN_CASE(4) DGN_CASE(5) DGN_CASE(6) DGN_CASE(7) DGN_CASE(8)
        DGN_CASE(9) DGN_CASE(10) DGN_CASE(11) DGN_CASE(12) DGN_CASE(13) DGN_CASE(14) DGN_CASE(15) DGN_CASE(16)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}

template <int KT>
hipError_t launch_wgrad_kt(const WgradParams& p, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ts_gemm_wgrad<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL((ts_gemm_wgrad<KT>), grid, dim3(kWave * kWgWaves), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_wgrad(int kt, const WgradParams& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (kt) {
#define DGN_CASE(K) case K: return launch_wgrad_kt<K>(p, grid, lds, st);
        DGN_CASE(1) DGN_CASE(2) DGN_CASE(3) DGN_CASE(4) DGN_CASE(5) DGN_CASE(6) DGN_CASE(7) DGN_CASE(8) DGN_CASE(9) DGN_CASE(10) DGN_CASE(11)
        DGN_CASE(12) DGN_CASE(13) DGN_CASE(14) DGN_CASE(15) DGN_CASE(16)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}

struct WgPlan { int k_slices, k_slice, kt, nt, slots; size_t lds; };
WgPlan wgrad_plan(int64_t M, int k, int n) {
    WgPlan w{};
    w.nt = (n + 15) / 16;
    w.k_slices = (k + 16 * kMaxKT - 1) / (16 * kMaxKT);
    const int per = (k + w.k_slices - 1) / w.k_slices;
    w.kt = (per + 15) / 16;
    w.k_slice = w.kt * 16;
    w.k_slices = (k + w.k_slice - 1) / w.k_slice;
    const int64_t n_strips = (M + 15) / 16;
    w.slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_cus(), n_strips));
    w.lds = (size_t)2 * 16 * ((w.nt * 16 + 4) + (w.kt * 16 + 4)) * sizeof(float);
    return w;
}

// tile_wgrad: blocks of at most 8 x 8 tiles of 32 columns, as even as the widths allow; one workgroup per CU (LDS), the row strips dealt
// round-robin to `slots` workgroups per block
struct TwPlan { int nb_tiles, kb_tiles, n_blocks, k_blocks, slots; size_t lds; };
TwPlan tile_wgrad_plan(int64_t M, int kk, int n) {
    TwPlan t{};
    const int nt = (n + 31) / 32, kt = (kk + 31) / 32;
    t.n_blocks = (nt + 7) / 8;
    t.nb_tiles = (nt + t.n_blocks - 1) / t.n_blocks;
    t.k_blocks = (kt + 7) / 8;
    t.kb_tiles = (kt + t.k_blocks - 1) / t.k_blocks;
    const int64_t n_strips = (M + kTwRows - 1) / kTwRows;
    t.slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_strips, n_cus() / (t.n_blocks * t.k_blocks)));
    t.lds = (size_t)2 * kTwRows * (tw_stride(t.nb_tiles * 32) + tw_stride(t.kb_tiles * 32)) * sizeof(float);
    return t;
}

bool use_tile_wgrad(int n) {
    const int64_t o = option(OPT_TILE_WGRAD);
    return o >= 0 ? o != 0 : n > 16 * kWgWaves;
}

}  // namespace
}  // namespace gemm
}  // namespace dgn

using namespace dgn;
using namespace dgn::gemm;

// (widths of at least 4: the operand prefetches are branch-free 16-byte loads clamped into the row)
extern "C" int dgn_gemm_supported(int32_t k, int32_t n) { return k >= 4 && n >= 4 && k <= 4096 && n <= 4096; }

extern "C" int dgn_gemm_forward(int64_t n_rows, int32_t k, int32_t n, const float* a, int64_t lda, const float* w, int64_t ldw,
                                int32_t w_is_kn, const float* bias, float* c, int64_t ldc, void* stream) {
    const char* fn = "dgn_gemm_forward";
    if (n_rows < 0 || !dgn_gemm_supported(k, n)) { set_error("%s: widths outside 4..4096 (k=%d n=%d)", fn, k, n); return DGN_ERR_INVALID; }
    if (n_rows == 0) return DGN_OK;
    if (!a || !w || !c || lda < k || ldc < n || ldw < (w_is_kn ? n : k)) { set_error("%s: null operand or row stride smaller than the row", fn); return DGN_ERR_INVALID; }
    GemmParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.A = a; p.lda = lda; p.W = w; p.ldw = ldw; p.bias = bias; p.C = c; p.ldc = ldc;
    // 256-row tiles with both operands through LDS: wide outputs of nn.Linear-layout weights on many rows
    const int64_t tile_opt = option(OPT_TILE_GEMM);
    static const char* min_env = getenv("DGN_TILE_GEMM_MIN_ROWS");
    static const int64_t tile_min_rows = min_env ? atoll(min_env) : 131072;
    const bool tile = tile_opt >= 0 ? tile_opt != 0 : (n >= 64 && n_rows >= tile_min_rows);     // (fewer rows: too few 256-row tiles to fill the CUs -- measured slower at 52 k rows)
    if (tile && !w_is_kn) {
        // column tiles of 16 NQ (NQ <= 8): the split with the least padded columns, fewer tiles on a tie
        // (NQ = 8 would need 265 registers: one wave per SIMD)
        int best_nq = 7, best_tiles = (n + 111) / 112, best_pad = best_tiles * 112 - n;
        for (int nq = 7; nq >= 4; --nq) {
            const int tiles = (n + 16 * nq - 1) / (16 * nq), pad = tiles * 16 * nq - n;
            if (pad < best_pad) { best_nq = nq; best_tiles = tiles; best_pad = pad; }
        }
        p.n_slice = 16 * best_nq;
        // one workgroup per resident slot (two per CU: 256 registers per lane, 51-59 KB of LDS each), each with an equal range of
        // rows: 2150 tiles over 512 slots used to run as five rounds of which the last was a fifth full
        const int per_cu = 2;
        const int64_t slots_x = std::max<int64_t>(1, (int64_t)n_cus() * per_cu / best_tiles);
        p.rows_per_block = std::max<int64_t>(64, (((n_rows + slots_x - 1) / slots_x) + 63) / 64 * 64);
        const dim3 grid((unsigned)((n_rows + p.rows_per_block - 1) / p.rows_per_block), (unsigned)best_tiles);
        hipStream_t st = static_cast<hipStream_t>(stream);
        switch (best_nq) {
            case 4: hipLaunchKernelGGL(tile_gemm<4>, grid, dim3(256), 0, st, p); break;
            case 5: hipLaunchKernelGGL(tile_gemm<5>, grid, dim3(256), 0, st, p); break;
            case 6: hipLaunchKernelGGL(tile_gemm<6>, grid, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL(tile_gemm<7>, grid, dim3(256), 0, st, p); break;
        }
        DGN_HIP_CHECK(hipGetLastError());
        return DGN_OK;
    }
    int slices = (n + 16 * kMaxNT - 1) / (16 * kMaxNT);
    {   // Small batches: 2 970 rows are 24 row blocks -- 24 workgroups carrying the whole product on 24 of 256 CUs (23 us for 0.15 GFLOP on
        // the shipped ZINC layer at batch 128).  More, narrower column slices until the grid covers the chip; a slice re-reads the
        // block's A rows from L2, which at this size is nothing.
        const int64_t n_blocks0 = (n_rows + 16 * kWaves - 1) / (16 * kWaves);
        const int max_slices = (n + 15) / 16;
        const int64_t want = (int64_t)n_cus() / std::max<int64_t>(1, n_blocks0);
        static const bool no_split = getenv("DGN_GEMM_NO_COL_SPLIT") != nullptr;
        if (!no_split && want > slices) slices = (int)std::min<int64_t>(want, max_slices);
    }
    const int nt = ((n + slices - 1) / slices + 15) / 16;
    p.n_slice = nt * 16;
    const int gy = (n + p.n_slice - 1) / p.n_slice;
    const int64_t n_blocks = (n_rows + 16 * kWaves - 1) / (16 * kWaves);
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(n_blocks, (int64_t)n_cus() * 2 / gy + 1));
    DGN_HIP_CHECK(launch_gemm(nt, p, dim3(gx, gy), static_cast<hipStream_t>(stream), w_is_kn));
    return DGN_OK;
}

extern "C" size_t dgn_gemm_wgrad_workspace_bytes(int64_t n_rows, int32_t k, int32_t n) {
    if (n_rows <= 0 || !dgn_gemm_supported(k, n)) return 0;
    // dgn_gemm_wgrad plans with kk = k (no bias gradient) or k + 1 (bias gradient as a ones column), and neither plan is monotone in
    // kk (k = 256, n = 192: kk = 256 needs 50 MB of partials, kk = 257 needs 31 MB): the query covers BOTH
    size_t bytes = 0;
    for (int kk = k; kk <= k + 1; ++kk) {
        const TwPlan t = tile_wgrad_plan(n_rows, kk, n);
        bytes = std::max(bytes, (size_t)t.n_blocks * t.k_blocks * t.slots * (t.nb_tiles * 32) * (t.kb_tiles * 32) * sizeof(float));
        if (n <= 16 * kWgWaves) {
            const WgPlan w = wgrad_plan(n_rows, kk, n);
            bytes = std::max(bytes, (size_t)w.k_slices * w.slots * (w.nt * 16) * (w.kt * 16) * sizeof(float));
        }
    }
    return bytes;
}

extern "C" int dgn_gemm_wgrad(int64_t n_rows, int32_t k, int32_t n, const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw,
                              int64_t lddw, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "dgn_gemm_wgrad";
    if (n_rows < 0 || !dgn_gemm_supported(k, n)) { set_error("%s: widths outside 4..4096 (k=%d n=%d)", fn, k, n); return DGN_ERR_INVALID; }
    if (!dw || lddw < k) { set_error("%s: null output", fn); return DGN_ERR_INVALID; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_rows == 0) {
        if (zero_rows_async(dw, n, k, lddw, st)) return DGN_ERR_HIP;
        return dbias ? zero_rows_async(dbias, 1, n, n, st) : DGN_OK;
    }
    if (!g || !x || ldg < n || ldx < k) { set_error("%s: null operand or row stride smaller than the row", fn); return DGN_ERR_INVALID; }
    const size_t need = dgn_gemm_wgrad_workspace_bytes(n_rows, k, n);
    if (!ws || ws_bytes < need) { set_error("%s: workspace too small (%zu < %zu)", fn, ws_bytes, need); return DGN_ERR_WORKSPACE; }
    // n <= 256: the strip kernel (16 waves, one n tile each; measured 75-80 TFLOP/s on the posttrans shapes, where the tile kernel's
    // 2 x 4 wave grid of 32 x 32 tiles leaves SIMDs unevenly loaded: 0.347 vs 0.259 ms at n = 225, k = 152).  Wider outputs: the tile
    // kernel (v_mfma_f32_32x32x2_f32, whole accumulator blocks per workgroup).  Both carry the bias gradient as a column of ones.
    if (use_tile_wgrad(n) || n > 16 * kWgWaves) {
        TileWgParams p{};
        p.M = n_rows; p.n = n; p.k = k; p.kk = dbias ? k + 1 : k;
        p.G = g; p.ldg = ldg; p.X = x; p.ldx = ldx; p.part = static_cast<float*>(ws);
        const TwPlan t = tile_wgrad_plan(n_rows, p.kk, n);
        p.slots = t.slots; p.n_blocks = t.n_blocks; p.k_blocks = t.k_blocks; p.nb_tiles = t.nb_tiles; p.kb_tiles = t.kb_tiles;
        static bool attr = false;
        if (!attr) {
            DGN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        hipLaunchKernelGGL(tile_wgrad, dim3(t.slots, t.n_blocks * t.k_blocks), dim3(kWave * kTwWaves), t.lds, st, p);
        const int64_t total = (int64_t)n * p.kk;
        hipLaunchKernelGGL(tile_wgrad_finalize, dim3((unsigned)((total + 63) / 64)), dim3(64 * 16), 0, st, n, k, p.kk, t.slots, t.k_blocks,
                           t.nb_tiles * 32, t.kb_tiles * 32, p.part, dw, lddw, dbias);
        DGN_HIP_CHECK(hipGetLastError());
        return DGN_OK;
    }
    const int kk = dbias ? k + 1 : k;
    const WgPlan w = wgrad_plan(n_rows, kk, n);
    WgradParams p{};
    p.M = n_rows; p.n = n; p.k = k; p.kk = kk; p.G = g; p.ldg = ldg; p.X = x; p.ldx = ldx; p.part = static_cast<float*>(ws);
    p.k_slice = w.k_slice; p.slots = w.slots;
    DGN_HIP_CHECK(launch_wgrad(w.kt, p, dim3(w.slots, w.k_slices), w.lds, st));
    const int64_t total = (int64_t)n * kk;
    hipLaunchKernelGGL(ts_gemm_wgrad_finalize, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n, k, kk, w.k_slice, w.slots, w.nt * 16,
                       w.kt * 16, p.part, dw, lddw, dbias);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
