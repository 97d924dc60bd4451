// Host side of the degree-class posttrans products (include/dgn_hip.h: dgn_dc_*), kernels in dgn_dc_kernels.hpp.
#include "dgn_dc_kernels.hpp"

#include <algorithm>
#include <cstdlib>

namespace dgn {
namespace dc {
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

bool check_classes(const char* fn, const DgnDegreeClasses* d) {
    if (!d || d->n_units < 0 || (d->n_units > 0 && (!d->vperm || !d->unit_class || !d->present || !d->scale))) {
        set_error("%s: incomplete DgnDegreeClasses", fn);
        return false;
    }
    return true;
}

struct WgPlan { int ntn, kt, k_slice, k_slices, kpad, slots; int64_t units_per_block; size_t lds, part_floats; };
WgPlan wgrad_plan(int64_t n_units, int k, int n) {
    WgPlan w{};
    w.ntn = (n + 15) / 16;
    // k tiles per wave: at most 4 (n <= 80), 3 (n <= 96), 2 (n <= 128) -- the accumulators of NTN x KT tiles must fit 256 registers
    const int kt_max = w.ntn <= 5 ? 4 : (w.ntn == 6 ? 3 : 2), slice_max = 16 * kWgWaves * kt_max;
    w.k_slices = (k + slice_max - 1) / slice_max;
    const int per = (k + w.k_slices - 1) / w.k_slices;
    w.kt = ((per + 15) / 16 + kWgWaves - 1) / kWgWaves;
    w.k_slice = 16 * kWgWaves * w.kt;
    w.k_slices = (k + w.k_slice - 1) / w.k_slice;
    w.kpad = (std::min(w.k_slice, k) + 15) / 16 * 16;
    const int64_t want = std::max<int64_t>(1, n_cus() / w.k_slices);
    w.units_per_block = std::max<int64_t>(1, (n_units + want - 1) / want);
    w.slots = (int)std::max<int64_t>(1, (n_units + w.units_per_block - 1) / w.units_per_block);
    w.lds = (size_t)2 * 16 * ((w.ntn * 16 + 4) + (w.kt * 128 + 4)) * sizeof(float);
    w.part_floats = (size_t)w.k_slices * (w.slots + kClasses) * (w.ntn * 16) * w.kpad;
    return w;
}

template <int NTN, int KT>
hipError_t launch_wgrad_t(const DcWgradParams& p, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dc_wgrad<NTN, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL((dc_wgrad<NTN, KT>), grid, dim3(kWave * kWgWaves), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_wgrad(int ntn, int kt, const DcWgradParams& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (ntn * 4 + kt) {      // kt in 1..4, see wgrad_plan
#define DGN_CASE(N, K) case N * 4 + K: return launch_wgrad_t<N, K>(p, grid, lds, st);
        DGN_CASE(1, 1) DGN_CASE(1, 2) DGN_CASE(1, 3) DGN_CASE(1, 4) DGN_CASE(2, 1) DGN_CASE(2, 2) DGN_CASE(2, 3) DGN_CASE(2, 4)
        DGN_CASE(3, 1) DGN_CASE(3, 2) DGN_CASE(3, 3) DGN_CASE(3, 4) DGN_CASE(4, 1) DGN_CASE(4, 2) DGN_CASE(4, 3) DGN_CASE(4, 4)
        DGN_CASE(5, 1) DGN_CASE(5, 2) DGN_CASE(5, 3) DGN_CASE(5, 4) DGN_CASE(6, 1) DGN_CASE(6, 2) DGN_CASE(6, 3)
        DGN_CASE(7, 1) DGN_CASE(7, 2) DGN_CASE(8, 1) DGN_CASE(8, 2)
#undef DGN_CASE
    }
    return hipErrorInvalidValue;
}

}  // namespace
}  // namespace dc
}  // namespace dgn

using namespace dgn;
using namespace dgn::dc;

// widths: k, n >= 4 (16-byte clamped operand loads); the weight gradient keeps n <= 128 output rows in one wave's accumulators
extern "C" int dgn_dc_supported(int32_t k, int32_t n) { return k >= 4 && n >= 4 && k <= 4096 && n <= 4096; }
extern "C" int dgn_dc_wgrad_supported(int32_t k, int32_t n) { return k >= 4 && n >= 4 && k <= 4096 && n <= 128; }

namespace {
bool check_layout(const char* fn, const DgnDcLayout* lay, int k) {
    if (!lay) return true;
    if (lay->n_agg < 1 || lay->f_pad < lay->f_in || lay->f_in < 1 || (lay->h_off != 0 && lay->h_off != lay->f_in) ||
        k != (lay->n_agg + (lay->h_off ? 1 : 0)) * lay->f_pad) {
        set_error("%s: DgnDcLayout does not describe %d columns", fn, k);
        return false;
    }
    return true;
}
}  // namespace

extern "C" int dgn_dc_fold(const DgnDegreeClasses* d, int32_t S, int32_t n, int32_t k, int32_t towers, const float* wf, const DgnDcLayout* layout, float* wc,
                           float* wct, void* stream) {
    const char* fn = "dgn_dc_fold";
    if (!check_classes(fn, d)) return DGN_ERR_INVALID;
    if (S < 1 || S > 3 || n < 1 || k < 1 || towers < 1 || towers > 64 || !wf || !wc || !wct || (layout && towers != 1)) {
        set_error("%s: bad shape or null buffer", fn);
        return DGN_ERR_INVALID;
    }
    if (!check_layout(fn, layout, k)) return DGN_ERR_INVALID;
    if (d->n_units == 0) return DGN_OK;
    const DgnDcLayout none{};
    hipLaunchKernelGGL(dc_fold, dim3((unsigned)(((int64_t)n * k + 255) / 256), kClasses * towers), dim3(256), 0, static_cast<hipStream_t>(stream), S, n, k,
                       towers, d->present, d->scale, wf, wc, wct, layout ? *layout : none, layout ? 1 : 0);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" int dgn_dc_gemm(const DgnDegreeClasses* d, int32_t k, int32_t n, int32_t towers, const float* a, int64_t lda, int64_t a_tower, const float* w,
                           int64_t ldw, int64_t class_stride, int64_t w_tower, const float* bias, const float* row_scale, float* c, int64_t ldc,
                           int64_t c_tower, int32_t stream_out, void* stream) {
    return dgn::dc::gemm_stats(d, k, n, towers, a, lda, a_tower, w, ldw, class_stride, w_tower, bias, row_scale, c, ldc, c_tower, stream_out, nullptr, 0, nullptr,
                               stream);
}

// (the most slots gemm_stats writes: row ranges of dc_gemm -- at most two per CU, rounded up to the XCD count -- or units of dc_gemm_small -- at most four per CU)
size_t dgn::dc::gemm_stats_bytes(int32_t n) { return (size_t)2 * n * ((size_t)n_cus() * 4 + kXcds) * sizeof(double); }

// dgn_dc_gemm with BatchNorm's column partials of C riding in the epilogue (DcGemmParams.bn_part; one tower): *slots = the G bn_finalize reads
int dgn::dc::gemm_stats(const DgnDegreeClasses* d, int32_t k, int32_t n, int32_t towers, const float* a, int64_t lda, int64_t a_tower, const float* w,
                        int64_t ldw, int64_t class_stride, int64_t w_tower, const float* bias, const float* row_scale, float* c, int64_t ldc,
                        int64_t c_tower, int32_t stream_out, double* part, size_t part_bytes, int* slots, void* stream) {
    const char* fn = "dgn_dc_gemm";
    if (part && (towers != 1 || !slots || part_bytes < gemm_stats_bytes(n))) { set_error("%s: statistics need one tower and gemm_stats_bytes()", fn); return DGN_ERR_INVALID; }
    if (!check_classes(fn, d)) return DGN_ERR_INVALID;
    if (!dgn_dc_supported(k, n) || towers < 1 || towers > 64) { set_error("%s: widths outside 4..4096 (k=%d n=%d) or towers outside 1..64", fn, k, n); return DGN_ERR_INVALID; }
    if (d->n_units == 0) return DGN_OK;
    if (!a || !w || !c || lda < k || ldc < n || ldw < k) { set_error("%s: null operand or row stride smaller than the row", fn); return DGN_ERR_INVALID; }
    DcGemmParams p{};
    p.n_units = d->n_units; p.vperm = d->vperm; p.unit_class = d->unit_class; p.k = k; p.n = n; p.A = a; p.lda = lda; p.W = w; p.ldw = ldw;
    p.class_stride = class_stride; p.bias = bias; p.row_scale = row_scale; p.C = c; p.ldc = ldc; p.stream_out = stream_out;
    p.a_tower = a_tower; p.w_tower = w_tower; p.c_tower = c_tower; p.bias_tower = n;
    // column tiles of 16 NQ (NQ <= 7): the fewest padded columns, a tile's fixed cost (prologue, operand re-reads) priced at 16 columns
    int best_nq = 7, best_tiles = (n + 111) / 112, best_cost = best_tiles * (112 + 16);
    for (int nq = 6; nq >= 1; --nq) {
        const int tiles = (n + 16 * nq - 1) / (16 * nq), cost = tiles * (16 * nq + 16);
        if (cost < best_cost) { best_nq = nq; best_tiles = tiles; best_cost = cost; }
    }
    // 113 .. 128 columns (hidden 128: C5): ONE tile of 128 columns -- the A rows are fetched and staged once instead of twice, 128 MFMAs
    // per k chunk and wave behind 12 LDS reads instead of 64 behind 8 (DGN_DC_NQ8=0: two tiles of 64)
    static const bool nq8 = !(getenv("DGN_DC_NQ8") && atoi(getenv("DGN_DC_NQ8")) == 0);
    if (nq8 && n > 112 && n <= 128) { best_nq = 8; best_tiles = 1; }
    p.n_slice = 16 * best_nq;
    // one workgroup per resident slot (two per CU), each with an equal range of units
    const int64_t slots_x = std::max<int64_t>(1, (int64_t)n_cus() * 2 / (best_tiles * towers));
    p.units_per_block = std::max<int64_t>(1, (d->n_units + slots_x - 1) / slots_x);
    const int64_t ranges = (d->n_units + p.units_per_block - 1) / p.units_per_block;
    static const bool no_xcd = getenv("DGN_DC_NO_XCD") != nullptr;
    p.col_tiles = (best_tiles > 1 && !no_xcd) ? best_tiles : 1;
    const dim3 grid = p.col_tiles > 1 ? dim3((unsigned)((ranges + kXcds - 1) / kXcds * kXcds * best_tiles), 1, (unsigned)towers)
                                      : dim3((unsigned)ranges, (unsigned)best_tiles, (unsigned)towers);
    hipStream_t st = static_cast<hipStream_t>(stream);
    p.bn_part = best_nq <= 6 ? part : nullptr; p.bn_F = n;      // (wider tiles: no statistics, *slots = 0 -- the caller runs bn_stats)
    p.bn_G = (int)(p.col_tiles > 1 ? grid.x / best_tiles : ranges);
    if (part) *slots = p.bn_part ? p.bn_G : 0;
    // few units: one unit per workgroup, four workgroups per CU (dc_gemm_small) while every (unit, column tile) is resident at once
    // (few ROWS is the criterion -- at most four units per CU --: column tiles multiply the workgroups of both kernels alike.  HIV batch 2048,
    //  816 units, captured step 0.348 -> 0.338 ms with the forward and the input-gradient product on it; DGN_DC_SMALL=0: off)
    static const bool small_on = !(getenv("DGN_DC_SMALL") && atoi(getenv("DGN_DC_SMALL")) == 0);
    if (small_on && best_nq <= 6 && d->n_units * towers <= (int64_t)n_cus() * 4) {
        const dim3 gs((unsigned)d->n_units, (unsigned)best_tiles, (unsigned)towers);
        p.col_tiles = 1;
        p.bn_G = (int)d->n_units;
        if (part) *slots = p.bn_G;
        switch (best_nq) {
            case 1: hipLaunchKernelGGL(dc_gemm_small<1>, gs, dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL(dc_gemm_small<2>, gs, dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL(dc_gemm_small<3>, gs, dim3(256), 0, st, p); break;
            case 4: hipLaunchKernelGGL(dc_gemm_small<4>, gs, dim3(256), 0, st, p); break;
            case 5: hipLaunchKernelGGL(dc_gemm_small<5>, gs, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL(dc_gemm_small<6>, gs, dim3(256), 0, st, p); break;
        }
        DGN_HIP_CHECK(hipGetLastError());
        return DGN_OK;
    }
    switch (best_nq) {
        case 1: hipLaunchKernelGGL(dc_gemm<1>, grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(dc_gemm<2>, grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(dc_gemm<3>, grid, dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(dc_gemm<4>, grid, dim3(256), 0, st, p); break;
        case 5: hipLaunchKernelGGL(dc_gemm<5>, grid, dim3(256), 0, st, p); break;
        case 6: hipLaunchKernelGGL(dc_gemm<6>, grid, dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL(dc_gemm<8>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(dc_gemm<7>, grid, dim3(256), 0, st, p); break;
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

extern "C" size_t dgn_dc_wgrad_workspace_bytes(int64_t n_units, int32_t k, int32_t n) {
    if (n_units <= 0 || !dgn_dc_wgrad_supported(k, n)) return 0;
    const WgPlan w = wgrad_plan(n_units, k, n);
    return w.part_floats * sizeof(float) + (size_t)w.slots * sizeof(uint32_t) + 256;
}

extern "C" int dgn_dc_wgrad(const DgnDegreeClasses* d, int32_t S, int32_t k, int32_t n, const float* g, int64_t ldg, const float* x, int64_t ldx,
                            float* g_wf, int64_t ldw, const DgnDcLayout* layout, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "dgn_dc_wgrad";
    if (!check_classes(fn, d)) return DGN_ERR_INVALID;
    if (!dgn_dc_wgrad_supported(k, n) || S < 1 || S > 3) { set_error("%s: unsupported widths (k=%d n=%d S=%d)", fn, k, n, S); return DGN_ERR_INVALID; }
    if (!g_wf || (!layout && ldw < k)) { set_error("%s: null output", fn); return DGN_ERR_INVALID; }
    if (!check_layout(fn, layout, k)) return DGN_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->n_units == 0) return layout ? zero_rows_async(g_wf, n, layout->ld, layout->ld, st) : zero_rows_async(g_wf, (int64_t)S * n, k, ldw, st);
    if (!g || !x || ldg < n || ldx < k) { set_error("%s: null operand or row stride smaller than the row", fn); return DGN_ERR_INVALID; }
    const size_t need = dgn_dc_wgrad_workspace_bytes(d->n_units, k, n);
    if (!ws || ws_bytes < need) { set_error("%s: workspace too small (%zu < %zu)", fn, ws_bytes, need); return DGN_ERR_WORKSPACE; }
    const WgPlan w = wgrad_plan(d->n_units, k, n);
    DcWgradParams p{};
    p.n_units = d->n_units; p.vperm = d->vperm; p.unit_class = d->unit_class; p.n = n; p.k = k; p.G = g; p.ldg = ldg; p.X = x; p.ldx = ldx;
    p.part = static_cast<float*>(ws);
    p.run_mask = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + ((w.part_floats * sizeof(float) + 255) & ~(size_t)255));
    p.k_slice = w.k_slice; p.kpad = w.kpad; p.slots = w.slots; p.units_per_block = w.units_per_block;
    const DgnDcLayout none{};
    const DgnDcLayout lay = layout ? *layout : none;
    const int hl = layout ? 1 : 0;
    DGN_HIP_CHECK(launch_wgrad(w.ntn, w.kt, p, dim3(w.slots, w.k_slices), w.lds, st));
    const int64_t total = (int64_t)n * k;
    const dim3 fgrid((unsigned)((total + 63) / 64));
    switch (S) {
        case 1: hipLaunchKernelGGL(dc_wgrad_finalize<1>, fgrid, dim3(64 * 16), 0, st, n, k, w.k_slice, w.kpad, w.ntn * 16, w.slots, p.run_mask, d->scale, p.part, g_wf, ldw, lay, hl); break;
        case 2: hipLaunchKernelGGL(dc_wgrad_finalize<2>, fgrid, dim3(64 * 16), 0, st, n, k, w.k_slice, w.kpad, w.ntn * 16, w.slots, p.run_mask, d->scale, p.part, g_wf, ldw, lay, hl); break;
        default: hipLaunchKernelGGL(dc_wgrad_finalize<3>, fgrid, dim3(64 * 16), 0, st, n, k, w.k_slice, w.kpad, w.ntn * 16, w.slots, p.run_mask, d->scale, p.part, g_wf, ldw, lay, hl); break;
    }
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}
