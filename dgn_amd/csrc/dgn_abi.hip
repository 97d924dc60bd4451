// Error plumbing and version of libdgn_hip.so.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "dgn_common.hpp"

namespace dgn {
namespace {
thread_local char g_err[512] = "";
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
__global__ void zero_rows_kernel(int64_t rows, int64_t width, int64_t ld, float* __restrict__ p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * width) return;
    const int64_t r = i / width;
    p[r * ld + (i - r * width)] = 0.f;
}
}  // namespace

// rows x width floats := 0 (row stride ld) by a KERNEL: hipMemsetAsync nodes did not replay reliably under stream capture on this
// ROCm (a replayed step then accumulated on stale values), and every zero-fill of this library must be capture-safe
int zero_rows_async(float* p, int64_t rows, int64_t width, int64_t ld, hipStream_t stream) {
    if (!p || rows <= 0 || width <= 0) return DGN_OK;
    const int64_t n = rows * width;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, width, ld, p);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return DGN_ERR_HIP;
}
}  // namespace dgn

namespace dgn {
namespace {
struct OptDef { const char* name; const char* env; int64_t def; bool presence; };      // presence: the variable's existence means 1
const OptDef kOpts[OPT_COUNT] = {
    {"blk_lds_kb", "DGN_BLK_LDS_KB", 13, false},                 // LDS of a wave's block in agg_bwd_block (KB)
    {"blk_min_nodes", "DGN_BLK_MIN_NODES", 131072, false},       // batches from this many nodes on take agg_bwd_block
    {"bwd_rows_per_wave", "DGN_BWD_ROWS_PER_WAVE", 4, false},    // <= 1: the staged backward sweep with one row per wave
    {"tile_gemm", "DGN_TILE_GEMM", -1, false},                   // 0 / 1: force the strip / tile product (-1: by shape)
    {"tile_wgrad", "DGN_TILE_WGRAD", -1, false},                 // 0 / 1: force the strip / tile weight gradient (-1: by shape)
    {"no_zmask", "DGN_NO_ZMASK", 0, true},                       // towers layer: keep the mixing network's pre-activation instead of its sign mask
    {"linear_small_min_waves", "DGN_LINEAR_SMALL_MIN_WAVES", 8, false},
    {"graph_bwd_tiles", "DGN_GRAPH_BWD_TILES", 0, false},        // feature tiles of the graph backward (0: by batch size and LDS)
    {"odd_direct", "DGN_ODD_DIRECT", 1, false},                  // simple layers at odd hidden sizes: the sweep reads the un-padded rows (0: padded copy)
    {"bn_from_wgrad", "DGN_BN_FROM_WGRAD", 1, false},            // towers layer backward: BatchNorm's column sums derived from the mixing weight gradient (0: their own pass)
    {"mix_bwd_fused", "DGN_MIX_BWD_FUSED", 1, false},            // towers layer backward: mixing weight gradient straight from g_out + mask, BatchNorm's backward in the input-gradient product's epilogue (0: round 5's sequence)
    {"blk_lds_pad_kb", "DGN_BLK_LDS_PAD_KB", 0, false},          // experiments: unused LDS added to agg_bwd_block's allocation (fewer resident waves per CU: the occupancy what-if of profiles/NOTES.md)
    {"lin_wreg", "DGN_LIN_WREG", 3, false},                     // streaming posttrans products up to 18 tiles with the weights register-resident, 12 waves per CU: bit 0 the combine-epilogue product, bit 1 the expanded-operand product (0: weights re-read from LDS per strip, 16 waves)
    {"bd_bwd_fused", "DGN_BD_BWD_FUSED", 1, false},             // towers layer backward: the block-diagonal pretrans product's input gradient and weight gradient in one pass over d(P|Q) (0: two kernels)
    {"bn_stats_fused", "DGN_BN_STATS_FUSED", 1, false},         // towers layer forward: BatchNorm's column sums ride in the posttrans product's combine epilogue (fp64 LDS cells; 0: bn_stats, a pass of its own over y0)
};
std::atomic<int64_t> g_opt[OPT_COUNT];
std::once_flag g_opt_once;
void init_options() {
    for (int i = 0; i < OPT_COUNT; ++i) {
        const char* e = getenv(kOpts[i].env);
        g_opt[i].store(e ? (kOpts[i].presence ? 1 : atoll(e)) : kOpts[i].def, std::memory_order_relaxed);
    }
    if (getenv("DGN_NO_MIX_FUSED")) g_opt[OPT_NO_ZMASK].store(1, std::memory_order_relaxed);
}
int find_option(const char* name) {
    if (name)
        for (int i = 0; i < OPT_COUNT; ++i)
            if (!strcmp(name, kOpts[i].name)) return i;
    return -1;
}
}  // namespace
int64_t option(Opt o) {
    std::call_once(g_opt_once, init_options);
    return g_opt[o].load(std::memory_order_relaxed);
}
}  // namespace dgn

extern "C" int dgn_set_option(const char* name, int64_t value) {
    const int i = dgn::find_option(name);
    if (i < 0) { dgn::set_error("dgn_set_option: unknown option '%s'", name ? name : "(null)"); return DGN_ERR_INVALID; }
    std::call_once(dgn::g_opt_once, dgn::init_options);
    dgn::g_opt[i].store(value, std::memory_order_relaxed);
    return DGN_OK;
}
extern "C" int64_t dgn_get_option(const char* name) {
    const int i = dgn::find_option(name);
    if (i < 0) { dgn::set_error("dgn_get_option: unknown option '%s'", name ? name : "(null)"); return INT64_MIN; }
    return dgn::option(static_cast<dgn::Opt>(i));
}

extern "C" int dgn_abi_version(void) { return DGN_ABI_VERSION; }
extern "C" size_t dgn_sizeof(const char* name) {
#define DGN_SZ(T) if (name && !strcmp(name, #T)) return sizeof(T);
    DGN_SZ(DgnGraph) DGN_SZ(DgnChannel) DGN_SZ(DgnAggSpec) DGN_SZ(DgnMsg) DGN_SZ(DgnMsgGrad) DGN_SZ(DgnBnGrad) DGN_SZ(DgnTowersLayer)
    DGN_SZ(DgnTowersGrads) DGN_SZ(DgnDegreeClasses) DGN_SZ(DgnDcLayout) DGN_SZ(DgnDenseLayer) DGN_SZ(DgnDenseGrads) DGN_SZ(DgnBlockTable)
    DGN_SZ(DgnBlockLayer) DGN_SZ(DgnBlockGrads)
#undef DGN_SZ
    return 0;
}
extern "C" const char* dgn_last_error(void) { return dgn::g_err; }
