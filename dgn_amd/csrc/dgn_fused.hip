// Host side of the fused layer kernels (include/dgn_hip.h: dgn_layer_fused_*): validation, configuration dispatch, launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dgn_fused_kernels.hpp"

namespace dgn {
// defined in dgn_agg.hip
int agg_validate_and_fill(AggParams& p, const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                          const float* log_deg);
int agg_backward_prepare(AggParams& p, const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                         const float* log_deg, const float* g_out, int64_t ld_gout, bool lds_gout, const DgnMsgGrad* grads, void* ws,
                         size_t ws_bytes, void* stream_, float** tab_part_out, const unsigned char* aux);
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

template <class C, class O>
int launch_fwd(const FusedParams& p, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        DGN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_fwd_fused<C, O>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
        attr = true;
    }
    const int per_cu = std::max(1, (int)(160 * 1024 / lds));                       // workgroups of one CU alternate between sweep and MFMA phases
    const unsigned grid = (unsigned)std::min<int64_t>(p.n_iters, (int64_t)n_cus() * per_cu);
    hipLaunchKernelGGL((layer_fwd_fused<C, O>), dim3(grid), dim3(kWave * kFusedWaves), lds, st, p);
    DGN_HIP_CHECK(hipGetLastError());
    return DGN_OK;
}

// The fused kernels exist for the aggregator lists of the reference's configs (dgn_agg_hot.hpp) only: with the list baked in
// (StaticOps) the sweep half stays under the 128 registers a 16-wave workgroup leaves per lane; the generic list decoder does not
// (it spilled in every accumulator configuration).  Any other list runs sweep and posttrans as separate kernels.
bool is_hot(const AggParams& a) {
    static const bool no_hot = getenv("DGN_NO_HOT") != nullptr;
    if (no_hot) return false;
#define DGN_HOT(NA, OPS, CHS, NS, SCS, N, S, A)                                                                          \
    if (a.n_agg == NA && a.op_pack == OPS && a.ch_pack == CHS && a.n_scalers == NS && a.scaler_pack == SCS && a.agg_total == NA && \
        a.agg_offset == 0 && a.n_ch == N)                                                                                \
        return NS == 1;
#include "dgn_agg_hot.hpp"
#undef DGN_HOT
    return false;
}

int dispatch_fwd(const FusedParams& p, size_t lds, hipStream_t st) {
    const AggParams& a = p.a;
#define DGN_HOT(NA, OPS, CHS, NS, SCS, N, S, A)                                                                          \
    if (a.n_agg == NA && a.op_pack == OPS && a.ch_pack == CHS && a.n_scalers == NS && a.scaler_pack == SCS && a.agg_total == NA && \
        a.agg_offset == 0 && a.n_ch == N) {                                                                              \
        if constexpr (NS == 1) return launch_fwd<Cfg<2, N, S, A>, StaticOps<NA, OPS, CHS, NS, SCS>>(p, lds, st);         \
        else return DGN_ERR_INVALID;          /* (scalers are folded behind posttrans: lists with in-sweep scalers never come here) */ \
    }
#include "dgn_agg_hot.hpp"
#undef DGN_HOT
    set_error("no fused kernel for this aggregator list");
    return DGN_ERR_INVALID;
}


bool al8(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; }

}  // namespace
}  // namespace dgn

using namespace dgn;

extern "C" int dgn_layer_fused_supported(const DgnGraph* g, const DgnAggSpec* spec, int64_t F, int32_t n_scalers, int32_t f_out) {
    if (!g || !spec || F < 2 || (F & 1) || F > 2 * kWave || spec->n_towers < 1 || F % spec->n_towers) return 0;
    const int Ft = (int)(F / spec->n_towers);
    const int a_total = spec->agg_total > 0 ? spec->agg_total : spec->n_agg;
    const int K = a_total * Ft, n = n_scalers * f_out, nq = (n + 15) / 16;
    if ((Ft & 1) || K > 16 * kFusedKB || (K & 3) || n_scalers < 1 || n_scalers > 3 || f_out < 2 || (f_out & 1) || spec->n_towers * nq > kFusedWaves * kFusedUnits) return 0;
    if (spec->n_scalers != 1 || spec->scaler[0] != DGN_SCALE_IDENTITY || (spec->agg_total > 0 && (spec->agg_offset != 0 || spec->n_agg != spec->agg_total))) return 0;
    if (g->n_hub > 0 || g->n_src > 0 || g->max_in_degree <= 0 || g->max_in_degree > kWave) return 0;
    AggParams a{};
    a.n_towers = spec->n_towers; a.agg_total = a_total; a.Ft = Ft;
    a.n_agg = spec->n_agg; a.n_scalers = 1; a.n_ch = spec->n_ch; a.agg_offset = 0;
    for (int i = 0; i < spec->n_agg; ++i) {
        const int op = spec->agg_op[i];
        a.op_pack |= (uint64_t)op << (4 * i);
        a.ch_pack |= (uint64_t)((op >= DGN_AGG_DIR_AV && op <= DGN_AGG_DIR_DX_NO_ABS) ? spec->agg_ch[i] : 0) << (3 * i);
    }
    return is_hot(a) && fused_lds_floats(a, nq) * sizeof(float) <= 160 * 1024;
}

extern "C" int dgn_layer_fused_forward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                                       const float* log_deg, const float* weight, int64_t ldw, int64_t stride_w, int32_t n_scalers,
                                       int32_t f_out, const float* scale, const float* bias, const float* row_scale, float* y,
                                       int64_t ld_y, void* stream) {
    const char* fn = "dgn_layer_fused_forward";
    FusedParams p{};
    int rc = agg_validate_and_fill(p.a, g, spec, msg, w, ld_w, log_deg);
    if (rc) return rc;
    if (g->n_nodes == 0) return DGN_OK;
    if (!dgn_layer_fused_supported(g, spec, msg->F, n_scalers, f_out)) { set_error("%s: configuration outside the fused kernel's domain", fn); return DGN_ERR_INVALID; }
    if (!weight || !y || (n_scalers > 1 && !scale)) { set_error("%s: null operand", fn); return DGN_ERR_INVALID; }
    if (msg->m_edge && msg->x_src) { set_error("%s: messages with two gathered parts are not taken", fn); return DGN_ERR_INVALID; }
    auto ok2 = [&](const float* q, int64_t ld) { return !q || (al8(q) && (ld & 1) == 0); };
    if (!ok2(msg->x_src, msg->ld_src) || !ok2(msg->x_dst, msg->ld_dst) || !ok2(msg->m_edge, msg->ld_edge) || !ok2(msg->x_in, msg->ld_in) ||
        !ok2(y, ld_y) || !ok2(bias, 0) || ld_y < (int64_t)spec->n_towers * f_out) {
        set_error("%s: operands must be 8-byte aligned with even row strides", fn);
        return DGN_ERR_INVALID;
    }
    p.a.tower_stride = 0;
    p.W = weight; p.ldw = ldw; p.sW = stride_w;
    p.sc = n_scalers > 1 ? scale : nullptr; p.rs = row_scale; p.cb = bias;
    p.S = n_scalers; p.fo = f_out; p.nq = (n_scalers * f_out + 15) / 16;
    p.Y = y; p.ldy = ld_y;
    p.n_iters = (g->n_nodes + kFusedRows - 1) / kFusedRows;
    { static const char* e = getenv("DGN_FUSED_DBG"); p.dbg = e ? atoi(e) : 0; }
    const size_t lds = fused_lds_floats(p.a, p.nq) * sizeof(float);
    return dispatch_fwd(p, lds, static_cast<hipStream_t>(stream));
}
