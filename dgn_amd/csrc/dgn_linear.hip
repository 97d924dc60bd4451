// Host side of the tall-skinny fp32 Linear (include/dgn_hip.h, dgn_linear_*) and the plain ts_linear instantiations; the
// kernels are in dgn_linear_kernels.hpp, the other kernel families in dgn_linear_{combine,expand,wgrad}.hip.
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

hipError_t launch_linear_plain(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st) {
    return launch_linear_grid<kPlain>(nt, kb, p, threads, lds, st);
}

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

bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace
}  // namespace lin
}  // namespace dgn

using namespace dgn;
using namespace dgn::lin;



extern "C" int dgn_linear_supported(int32_t k, int32_t n, int32_t wgrad) {
    const bool ok = k >= 2 && n >= 2 && k % 2 == 0 && n % 2 == 0 && k <= 16 * kMaxTiles && n <= 16 * kMaxTiles;
    return ok && (!wgrad || ((n + 15) / 16) * ((k + 15) / 16) <= kMaxWgradTiles);
}

static int launch_linear(const char* fn, LinParams& p, void* stream) {
    p.kp = lds_stride(p.k);
    const int NT = (p.n + 15) / 16, KB = (p.k + 15) / 16;
    const bool bn = p.bn_mean != nullptr && p.bnb_gz == nullptr, actm = p.act_z != nullptr, addm = p.add1 != nullptr;
    const bool mixm = p.out2 != nullptr, maskm = p.act_mask != nullptr, bnbm = p.bnb_gz != nullptr;
    const int mode = bnbm ? kActMaskBnb : (mixm ? kMixFwd : (bn ? kBnPlain : (actm ? kActPlain : (maskm ? kActMask : (addm ? kAddPlain : kPlain)))));
    const size_t w_bytes = ((size_t)NT * 16 * p.kp + 2 * NT * 16 + (bnbm ? 5 * NT * 16 : (bn ? 4 * KB * 16 : (actm ? KB * 16 : 0)))) * 4;
    // (LinParams.bn_part: 4 fp64 cells per lane and epilogue trip, per wave, behind everything else)
    const size_t stat_bytes = p.bn_part ? (size_t)4 * ((kStrip * (p.fo >> 1) + 63) / 64) * 64 * sizeof(double) : 0;
    const size_t strip_bytes = ((size_t)strip_floats(p.k) + kStrip * p.n + kFacFloats) * 4 + stat_bytes;
    int waves = linear_threads(NT, KB, mode) / 64;
    p.wreg = (p.ex.gy || p.S > 0) && !bnbm && linear_wreg_ok(NT, KB, p.ex.gy ? kExpand : kCombine) && (option(OPT_LIN_WREG) & (p.ex.gy ? 2 : 1)) ? 1 : 0;      // (bit 0: combine epilogue, bit 1: expanded operand)
    if (p.wreg) waves = std::min(waves, 4);
    while (waves > 1 && w_bytes + waves * strip_bytes > (size_t)kLdsBudget) waves = waves == 12 ? 8 : waves / 2;
    const int64_t n_strips = (p.M + kStrip - 1) / kStrip;
    {   // Small batches: 2 970 rows are 186 strips -- twelve 16-wave workgroups on twelve of 256 CUs.  Fewer waves per workgroup until
        // the strips cover the chip (every workgroup stages the weights itself: 20 KB from L2).
        static const bool keep = getenv("DGN_LINEAR_NO_SMALL") != nullptr;
        const int min_waves = (int)option(OPT_LINEAR_SMALL_MIN_WAVES);           // (4: no gain, 2 and 1: slower -- the weights are staged by too few threads)
        while (!keep && waves > min_waves && n_strips * p.T < (int64_t)n_cus() * waves) waves = waves == 12 ? 8 : waves / 2;
    }
    const size_t wl_bytes = (size_t)NT * 16 * p.kp * 4;
    size_t lds = p.wreg ? w_bytes - wl_bytes + std::max(wl_bytes, waves * (strip_bytes - stat_bytes)) : w_bytes + waves * (strip_bytes - stat_bytes);
    if (p.bn_part) {
        lds = (lds + 7) & ~(size_t)7;
        p.st_off = (int)(lds / 4);
        lds += waves * stat_bytes;
    }
    if (lds > (size_t)kLdsBudget) { set_error("%s: weights do not fit in LDS", fn); return -1; }
    const int per_cu = std::max(1, std::min({(int)(kLdsBudget / lds), 32 / waves, p.wreg ? 3 : 32}));      // (WREG: 3 waves per SIMD by registers)
    int groups = std::max(1, n_cus() * per_cu / p.T);
    groups = (int)std::min<int64_t>(groups, (n_strips + waves - 1) / waves);
    p.groups = groups;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const hipError_t e = p.ex.gy ? launch_linear_expand(NT, KB, p, waves * 64, lds, st)
                       : bnbm    ? launch_linear_bnb(NT, KB, p, waves * 64, lds, st)
                       : p.S > 0 ? launch_linear_combine(NT, KB, p, waves * 64, lds, st)
                       : mixm    ? launch_linear_mix(NT, KB, p, waves * 64, lds, st)
                       : bn      ? launch_linear_bn(NT, KB, p, waves * 64, lds, st)
                       : actm    ? launch_linear_act(NT, KB, p, waves * 64, lds, st)
                       : maskm   ? launch_linear_actm(NT, KB, p, waves * 64, lds, st)
                       : addm    ? launch_linear_add(NT, KB, p, waves * 64, lds, st)
                                 : launch_linear_plain(NT, KB, p, waves * 64, lds, st);
    DGN_HIP_CHECK(e);
    return 0;
}

extern "C" int dgn_linear_forward(int64_t n_rows, int32_t k, int32_t n, int32_t batch, const float* a, int64_t lda,
                                  int64_t stride_a, const float* w, int64_t ldw, int64_t stride_w, int32_t w_is_kn,
                                  const float* bias, int64_t stride_bias, float* c, int64_t ldc, int64_t stride_c,
                                  void* stream) {
    const char* fn = "dgn_linear_forward";
    if (n_rows < 0 || batch < 1 || !dgn_linear_supported(k, n, 0)) { set_error("%s: need even k, n in [2, 160] (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!a || !w || !c) { set_error("%s: null operand", fn); return -1; }
    if (lda != k || ldc != n || (stride_a & 1) || (stride_c & 1) || !aligned8(a) || !aligned8(c)) {
        set_error("%s: A and C must have dense rows (lda == k, ldc == n) and 8-byte aligned batch entries", fn);
        return -1;
    }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = batch;
    p.A = a; p.sA = stride_a;
    p.W = w; p.ldw = ldw; p.sW = stride_w; p.w_kn = w_is_kn;
    p.bias = bias; p.sBias = stride_bias;
    p.C = c; p.sC = stride_c;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_forward_bn(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, int32_t w_is_kn,
                                     const float* bias, float* c, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                     const float* bn_beta, void* stream) {
    const char* fn = "dgn_linear_forward_bn";
    if (n_rows < 0 || !dgn_linear_supported(k, n, 0)) { set_error("%s: need even k, n in [2, 160] (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!a || !w || !c || !bn_mean || !bn_invstd) { set_error("%s: null operand", fn); return -1; }
    if (!aligned8(a) || !aligned8(c)) { set_error("%s: A and C must be 8-byte aligned (dense rows)", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = a; p.W = w; p.ldw = ldw; p.w_kn = w_is_kn;
    p.bias = bias;
    p.C = c;
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_gamma = bn_gamma; p.bn_beta = bn_beta;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_add_supported(int32_t k, int32_t n) {
    return dgn_linear_supported(k, n, 0) && linear_add_shape_ok((n + 15) / 16, (k + 15) / 16);
}

extern "C" int dgn_linear_forward_add(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, int32_t w_is_kn,
                                      const float* add1, const float* add2, float* c, void* stream) {
    const char* fn = "dgn_linear_forward_add";
    if (n_rows < 0 || !dgn_linear_add_supported(k, n)) { set_error("%s: widths outside the supported set (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!a || !w || !c || !add1) { set_error("%s: null operand", fn); return -1; }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!aligned8(a) || !al16(c) || !al16(add1) || (add2 && !al16(add2))) { set_error("%s: a must be 8-byte, c / add1 / add2 16-byte aligned (dense rows)", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = a; p.W = w; p.ldw = ldw; p.w_kn = w_is_kn;
    p.C = c;
    p.add1 = add1; p.add2 = add2;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_forward_bn_act(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, const float* bn_mean,
                                         const float* bn_invstd, const float* bn_gamma, const float* bn_beta, const float* act_bias, int32_t act,
                                         float slope, const float* residual, float* z_out, float* out, void* stream) {
    const char* fn = "dgn_linear_forward_bn_act";
    if (n_rows < 0 || !dgn_linear_add_supported(k, n)) { set_error("%s: widths outside the supported set (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!a || !w || !z_out || !out || !bn_mean || !bn_invstd) { set_error("%s: null operand", fn); return -1; }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!aligned8(a) || !al16(z_out) || !al16(out) || (residual && !al16(residual))) { set_error("%s: a must be 8-byte, z_out / out / residual 16-byte aligned", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = a; p.W = w; p.ldw = ldw; p.w_kn = 0;
    p.C = z_out;
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_gamma = bn_gamma; p.bn_beta = bn_beta;
    p.ep_bias = act_bias; p.act_kind = act; p.act_slope = slope; p.add1 = residual; p.out2 = out;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_act_supported(int32_t k, int32_t n) {
    return dgn_linear_supported(k, n, 0) && linear_act_shape_ok((n + 15) / 16, (k + 15) / 16);
}

extern "C" int dgn_linear_forward_act(int64_t n_rows, int32_t k, int32_t n, const float* g, const float* z, const float* act_bias, int32_t act,
                                      float slope, const float* w, int64_t ldw, int32_t w_is_kn, float* c, float* gz_out, void* stream) {
    const char* fn = "dgn_linear_forward_act";
    if (n_rows < 0 || !dgn_linear_act_supported(k, n)) { set_error("%s: need even k, n in [2, 160] (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!g || !z || !w || !c) { set_error("%s: null operand", fn); return -1; }
    if (!aligned8(g) || !aligned8(z) || !aligned8(c) || (gz_out && (reinterpret_cast<uintptr_t>(gz_out) & 15))) { set_error("%s: operands must be 8-byte aligned (dense rows), gz_out 16-byte aligned", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = g; p.W = w; p.ldw = ldw; p.w_kn = w_is_kn;
    p.C = c;
    p.act_z = z; p.act_bias = act_bias; p.act_kind = act; p.act_slope = slope; p.gz_out = gz_out;
    return launch_linear(fn, p, stream);
}

extern "C" size_t dgn_linear_act_mask_bytes(int64_t n_rows, int32_t n) { return n_rows > 0 && n > 0 ? (((size_t)n_rows * n / 2) + 15) & ~(size_t)15 : 0; }

extern "C" int dgn_linear_forward_bn_act_mask(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, const float* bn_mean,
                                              const float* bn_invstd, const float* bn_gamma, const float* bn_beta, const float* act_bias, int32_t act,
                                              float slope, const float* residual, unsigned char* zmask_out, float* out, void* stream) {
    const char* fn = "dgn_linear_forward_bn_act_mask";
    if (n_rows < 0 || !dgn_linear_add_supported(k, n)) { set_error("%s: widths outside the supported set (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!a || !w || !zmask_out || !out || !bn_mean || !bn_invstd) { set_error("%s: null operand", fn); return -1; }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!aligned8(a) || !al16(zmask_out) || !al16(out) || (residual && !al16(residual))) { set_error("%s: a must be 8-byte, zmask_out / out / residual 16-byte aligned", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = a; p.W = w; p.ldw = ldw; p.w_kn = 0;
    p.C = nullptr; p.zmask_out = zmask_out;
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_gamma = bn_gamma; p.bn_beta = bn_beta;
    p.ep_bias = act_bias; p.act_kind = act; p.act_slope = slope; p.add1 = residual; p.out2 = out;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_forward_act_mask(int64_t n_rows, int32_t k, int32_t n, const float* g, const unsigned char* zmask, int32_t act, float slope,
                                           const float* w, int64_t ldw, int32_t w_is_kn, float* c, float* gz_out, void* stream) {
    const char* fn = "dgn_linear_forward_act_mask";
    if (n_rows < 0 || !dgn_linear_act_supported(k, n)) { set_error("%s: need even k, n in [2, 160] (k=%d n=%d)", fn, k, n); return -1; }
    if (n_rows == 0) return 0;
    if (!g || !zmask || !w || !c) { set_error("%s: null operand", fn); return -1; }
    if (!aligned8(g) || (reinterpret_cast<uintptr_t>(zmask) & 1) || !aligned8(c) || (gz_out && (reinterpret_cast<uintptr_t>(gz_out) & 15))) { set_error("%s: g / c must be 8-byte aligned (dense rows), zmask 2-byte, gz_out 16-byte aligned", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = g; p.W = w; p.ldw = ldw; p.w_kn = w_is_kn;
    p.C = c;
    p.act_mask = zmask; p.act_kind = act; p.act_slope = slope; p.gz_out = gz_out;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_bnb_supported(int32_t k, int32_t n) {
    return dgn_linear_act_supported(k, n) && dgn_linear_add_supported(k, n) && linear_bnb_shape_ok((n + 15) / 16, (k + 15) / 16);
}

extern "C" int dgn_linear_forward_act_mask_bnb(int64_t n_rows, int32_t k, int32_t n, const float* g, const unsigned char* zmask, int32_t act, float slope,
                                               const float* w, int64_t ldw, int32_t w_is_kn, const float* y, const float* bn_mean, const float* bn_invstd,
                                               const float* bn_gamma, const float* sums, const float* row_scale, int32_t f_out, float* gz,
                                               int64_t stride_gz, void* stream) {
    const char* fn = "dgn_linear_forward_act_mask_bnb";
    if (n_rows < 0 || !dgn_linear_bnb_supported(k, n) || f_out < 2 || (f_out & 1) || n % f_out != 0 || n / f_out > 15) {
        set_error("%s: widths outside the supported set (k=%d n=%d f_out=%d: even f_out dividing n, at most 15 towers)", fn, k, n, f_out);
        return -1;
    }
    if (n_rows == 0) return 0;
    if (!g || !zmask || !w || !y || !bn_mean || !bn_invstd || !sums || !gz) { set_error("%s: null operand", fn); return -1; }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!aligned8(g) || (reinterpret_cast<uintptr_t>(zmask) & 1) || !al16(y) || !aligned8(gz) || (stride_gz & 1)) {
        set_error("%s: g 8-byte, zmask 2-byte, y 16-byte, gz 8-byte aligned (dense rows, even tower stride)", fn);
        return -1;
    }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = 1;
    p.A = g; p.W = w; p.ldw = ldw; p.w_kn = w_is_kn;
    p.act_mask = zmask; p.act_kind = act; p.act_slope = slope;
    p.bnb_y = y; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_gamma = bn_gamma; p.bnb_sums = sums; p.rs = row_scale; p.fo = f_out;
    p.bnb_gz = gz; p.bnb_sT = stride_gz;
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_combine_forward(int64_t n_rows, int32_t k, int32_t n_towers, int32_t n_scalers, int32_t f_out,
                                          const float* a, int64_t stride_a, const float* w, int64_t ldw, int64_t stride_w,
                                          const float* scale, const float* bias, const float* row_scale, float* y, int64_t ld_y,
                                          void* stream) {
    const char* fn = "dgn_linear_combine_forward";
    const int n = n_scalers * f_out;
    if (n_rows < 0 || n_towers < 1 || n_scalers < 1 || n_scalers > 3 || f_out < 1 || !dgn_linear_supported(k, n, 0)) {
        set_error("%s: need even k and n_scalers * f_out in [2, 160], at most 3 scalers (k=%d n=%d)", fn, k, n);
        return -1;
    }
    if (n_rows == 0) return 0;
    if (!a || !w || !y || (n_scalers > 1 && !scale)) { set_error("%s: null operand", fn); return -1; }
    if ((f_out & 1) || (ld_y & 1) || !aligned8(y)) { set_error("%s: f_out and ld_y must be even, y 8-byte aligned", fn); return -1; }
    if ((stride_a & 1) || !aligned8(a) || ld_y < (int64_t)n_towers * f_out) { set_error("%s: A entries must be 8-byte aligned, y rows n_towers * f_out wide", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = n_towers;
    p.A = a; p.sA = stride_a;
    p.W = w; p.ldw = ldw; p.sW = stride_w; p.w_kn = 0;
    p.S = n_scalers; p.fo = f_out; p.sc = scale; p.rs = row_scale; p.cb = bias; p.Y = y; p.ldy = ld_y;
    return launch_linear(fn, p, stream);
}

size_t dgn::lin::combine_forward_stats_bytes(int32_t n_towers, int32_t f_out) {
    return (size_t)2 * n_towers * f_out * std::max(1, n_cus() * 32) * sizeof(double);       // (groups * T <= CUs x the most workgroups a CU holds)
}

int dgn::lin::combine_forward_stats(int64_t n_rows, int32_t k, int32_t n_towers, int32_t n_scalers, int32_t f_out, const float* a, int64_t stride_a,
                                    const float* w, int64_t ldw, int64_t stride_w, const float* scale, const float* bias, const float* row_scale,
                                    float* y, int64_t ld_y, double* part, size_t part_bytes, int* groups, void* stream) {
    const char* fn = "combine_forward_stats";
    const int n = n_scalers * f_out;
    if (n_rows <= 0 || n_towers < 1 || n_scalers < 1 || n_scalers > 3 || f_out < 2 || (f_out & 1) || !dgn_linear_supported(k, n, 0) || kStrip * (f_out >> 1) > 256 ||
        !a || !w || !y || (n_scalers > 1 && !scale) || (ld_y & 1) || !aligned8(y) || (stride_a & 1) || !aligned8(a) || ld_y != (int64_t)n_towers * f_out || !part ||
        part_bytes < combine_forward_stats_bytes(n_towers, f_out))
        return 1;
    LinParams p{};
    p.M = n_rows; p.k = k; p.n = n; p.T = n_towers;
    p.A = a; p.sA = stride_a;
    p.W = w; p.ldw = ldw; p.sW = stride_w; p.w_kn = 0;
    p.S = n_scalers; p.fo = f_out; p.sc = scale; p.rs = row_scale; p.cb = bias; p.Y = y; p.ldy = ld_y;
    p.bn_part = part; p.bn_F = n_towers * f_out;
    const int rc = launch_linear(fn, p, stream);
    if (rc == 0) *groups = p.groups;
    return rc;
}

static int wgrad_groups(int64_t n_rows, int32_t k, int32_t n, int32_t batch) {
    const int64_t n_strips = (n_rows + kStrip - 1) / kStrip;
    const int per_cu = ((n + 15) / 16) * ((k + 15) / 16) <= 25 ? 2 : 1;          // registers allow two waves per SIMD
    int groups = std::max(1, n_cus() * per_cu / batch);
    return (int)std::max<int64_t>(1, std::min<int64_t>(groups, (n_strips + 3) / 4));
}

extern "C" size_t dgn_linear_wgrad_workspace_bytes(int64_t n_rows, int32_t k, int32_t n, int32_t batch) {
    if (n_rows <= 0 || batch < 1 || !dgn_linear_supported(k, n, 1)) return 0;
    const size_t NT = (n + 15) / 16, KT = (k + 15) / 16;
    return (size_t)batch * wgrad_groups(n_rows, k, n, batch) * NT * 16 * KT * 16 * sizeof(float);
}

static int launch_wgrad(const char* fn, WgParams& p, float* dw, int64_t lddw, int64_t stride_dw, float* dbias, int64_t stride_dbias,
                        void* ws, size_t ws_bytes, void* stream, float* pick = nullptr, int pick_off = 0, int pick_w = 0) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = p.n, k = p.k, batch = p.T;
    const size_t need = dgn_linear_wgrad_workspace_bytes(p.M, k, n, batch);
    if (!ws || ws_bytes < need) { set_error("%s: workspace too small (%zu < %zu)", fn, ws_bytes, need); return -1; }
    p.part = static_cast<float*>(ws);
    p.groups = wgrad_groups(p.M, k, n, batch);
    p.ones = dbias != nullptr;
    const int NT = (n + 15) / 16, KT = (k + 15) / 16;
    size_t lds = std::max((size_t)4 * wgrad_wave_floats(n, k, p.g_mask != nullptr), (size_t)NT * 16 * KT * 16) * 4;
    if (p.bn_mean) {
        p.bn_off = (int)(lds / 4);
        lds += (size_t)4 * KT * 16 * 4;
    }
    const hipError_t e = p.ex.gy ? launch_wgrad_expand(NT, KT, p, lds, st) : (p.g_mask ? launch_wgrad_gmask(NT, KT, p, lds, st) : launch_wgrad_plain(NT, KT, p, lds, st));
    DGN_HIP_CHECK(e);
    const int64_t total = (int64_t)batch * n * (dbias ? k + 1 : k);
    hipLaunchKernelGGL(ts_wgrad_finalize, dim3((unsigned)((total + 63) / 64)), dim3(64 * kFinWaves), 0, st, batch, n, k, p.groups,
                       NT * 16, KT * 16, p.part, dw, lddw, stride_dw, dbias, stride_dbias, dbias ? pick : nullptr, pick_off, pick_w);
    DGN_HIP_CHECK(hipGetLastError());
    return 0;
}

static int zero_wgrad(float* dw, int64_t lddw, int64_t stride_dw, float* dbias, int64_t stride_dbias, int k, int n, int batch,
                      hipStream_t st) {
    for (int t = 0; t < batch; ++t) {
        if (zero_rows_async(dw + t * stride_dw, n, k, lddw, st)) return DGN_ERR_HIP;
        if (dbias && zero_rows_async(dbias + t * stride_dbias, 1, n, n, st)) return DGN_ERR_HIP;
    }
    return 0;
}

extern "C" int dgn_linear_wgrad(int64_t n_rows, int32_t k, int32_t n, int32_t batch, const float* g, int64_t ldg,
                                int64_t stride_g, const float* x, int64_t ldx, int64_t stride_x, float* dw, int64_t lddw,
                                int64_t stride_dw, float* dbias, int64_t stride_dbias, void* ws, size_t ws_bytes,
                                void* stream) {
    const char* fn = "dgn_linear_wgrad";
    if (dbias && k % 16 == 0) { set_error("%s: the bias gradient rides in X's padding column (k %% 16 != 0)", fn); return -1; }
    if (n_rows < 0 || batch < 1 || !dgn_linear_supported(k, n, 1)) { set_error("%s: need even k, n in [2, 160] and at most 45 tiles (k=%d n=%d)", fn, k, n); return -1; }
    if (!dw) { set_error("%s: null output", fn); return -1; }
    if (n_rows == 0) return zero_wgrad(dw, lddw, stride_dw, dbias, stride_dbias, k, n, batch, static_cast<hipStream_t>(stream));
    if (!g || !x) { set_error("%s: null operand", fn); return -1; }
    if (ldg != n || ldx != k || (stride_g & 1) || (stride_x & 1) || !aligned8(g) || !aligned8(x)) {
        set_error("%s: G and X must have dense rows (ldg == n, ldx == k) and 8-byte aligned batch entries", fn);
        return -1;
    }
    WgParams p{};
    p.M = n_rows; p.n = n; p.k = k; p.T = batch;
    p.G = g; p.sG = stride_g;
    p.X = x; p.sX = stride_x;
    return launch_wgrad(fn, p, dw, lddw, stride_dw, dbias, stride_dbias, ws, ws_bytes, stream);
}

extern "C" int dgn_linear_wgrad_bn(int64_t n_rows, int32_t k, int32_t n, const float* g, const float* x, float* dw, int64_t lddw, float* dbias,
                                   const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta, void* ws,
                                   size_t ws_bytes, void* stream) {
    const char* fn = "dgn_linear_wgrad_bn";
    if (dbias && k % 16 == 0) { set_error("%s: the bias gradient rides in X's padding column (k %% 16 != 0)", fn); return -1; }
    if (n_rows < 0 || !dgn_linear_supported(k, n, 1)) { set_error("%s: need even k, n in [2, 160] and at most 45 tiles (k=%d n=%d)", fn, k, n); return -1; }
    if (!dw) { set_error("%s: null output", fn); return -1; }
    if (n_rows == 0) return zero_wgrad(dw, lddw, 0, dbias, 0, k, n, 1, static_cast<hipStream_t>(stream));
    if (!g || !x || !bn_mean || !bn_invstd || !aligned8(g) || !aligned8(x)) { set_error("%s: null or misaligned operand", fn); return -1; }
    WgParams p{};
    p.M = n_rows; p.n = n; p.k = k; p.T = 1;
    p.G = g; p.X = x;
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_gamma = bn_gamma; p.bn_beta = bn_beta;
    return launch_wgrad(fn, p, dw, lddw, 0, dbias, 0, ws, ws_bytes, stream);
}

extern "C" int dgn_linear_wgrad_bn_act_mask(int64_t n_rows, int32_t k, int32_t n, const float* g, const unsigned char* zmask, int32_t act, float slope,
                                            const float* x, float* dw, int64_t lddw, float* dbias, const float* bn_mean, const float* bn_invstd,
                                            const float* bn_gamma, const float* bn_beta, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "dgn_linear_wgrad_bn_act_mask";
    if (dbias && k % 16 == 0) { set_error("%s: the bias gradient rides in X's padding column (k %% 16 != 0)", fn); return -1; }
    if (n_rows < 0 || !dgn_linear_supported(k, n, 1)) { set_error("%s: need even k, n in [2, 160] and at most 45 tiles (k=%d n=%d)", fn, k, n); return -1; }
    if (!dw) { set_error("%s: null output", fn); return -1; }
    if (n_rows == 0) return zero_wgrad(dw, lddw, 0, dbias, 0, k, n, 1, static_cast<hipStream_t>(stream));
    if (!g || !zmask || !x || !bn_mean || !bn_invstd || !aligned8(g) || !aligned8(x) || (reinterpret_cast<uintptr_t>(zmask) & 15)) {
        set_error("%s: null or misaligned operand (g, x 8-byte; zmask 16-byte: its strips travel as 16-byte direct-to-LDS pieces)", fn);
        return -1;
    }
    WgParams p{};
    p.M = n_rows; p.n = n; p.k = k; p.T = 1;
    p.G = g; p.X = x;
    p.g_mask = zmask; p.act_kind = act; p.act_slope = slope;
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_gamma = bn_gamma; p.bn_beta = bn_beta;
    return launch_wgrad(fn, p, dw, lddw, 0, dbias, 0, ws, ws_bytes, stream);
}

static bool expand_ok(const char* fn, int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, const float* gy,
                      int64_t stride_gy, const float* scale) {
    if (n_rows < 0 || n_towers < 1 || n_scalers < 1 || n_scalers > 3 || f_out < 2 || (f_out & 1)) {
        set_error("%s: need 1..3 scalers and an even f_out", fn);
        return false;
    }
    if (n_rows > 0 && (!gy || (n_scalers > 1 && !scale) || (stride_gy & 1) || !aligned8(gy))) {
        set_error("%s: gy must be [n_towers][n_rows][f_out] with 8-byte aligned towers; scale is required for more than one scaler", fn);
        return false;
    }
    return true;
}

extern "C" int dgn_linear_combine_backward_input(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k,
                                                 const float* gy, int64_t stride_gy, const float* scale, const float* w, int64_t ldw,
                                                 int64_t stride_w, float* g_a, int64_t stride_ga, void* stream) {
    const char* fn = "dgn_linear_combine_backward_input";
    const int red = n_scalers * f_out;               // reduction width = rows of w[t]
    if (!expand_ok(fn, n_rows, n_towers, n_scalers, f_out, gy, stride_gy, scale)) return -1;
    if (!dgn_linear_supported(red, k, 0)) { set_error("%s: need even widths in [2, 160] (S*f_out=%d k=%d)", fn, red, k); return -1; }
    if (n_rows == 0) return 0;
    if (!w || !g_a || (stride_ga & 1) || !aligned8(g_a)) { set_error("%s: null or misaligned operand", fn); return -1; }
    LinParams p{};
    p.M = n_rows; p.k = red; p.n = k; p.T = n_towers;
    p.W = w; p.ldw = ldw; p.sW = stride_w; p.w_kn = 1;
    p.C = g_a; p.sC = stride_ga;
    if (n_scalers == 1) { p.A = gy; p.sA = stride_gy; }      // nothing to expand: gy is the operand
    else { p.ex.gy = gy; p.ex.sT = stride_gy; p.ex.sc = scale; p.ex.S = n_scalers; p.ex.fo = f_out; }
    return launch_linear(fn, p, stream);
}

extern "C" int dgn_linear_combine_backward_weight(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k,
                                                  const float* gy, int64_t stride_gy, const float* scale, const float* a,
                                                  int64_t stride_a, float* dw, int64_t lddw, int64_t stride_dw, void* ws,
                                                  size_t ws_bytes, void* stream) {
    return dgn_linear_combine_backward_weight_bias(n_rows, n_towers, n_scalers, f_out, k, gy, stride_gy, scale, a, stride_a, dw, lddw, stride_dw,
                                                   nullptr, ws, ws_bytes, stream);
}

extern "C" int dgn_linear_combine_backward_weight_bias(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k,
                                                       const float* gy, int64_t stride_gy, const float* scale, const float* a,
                                                       int64_t stride_a, float* dw, int64_t lddw, int64_t stride_dw, float* g_sum,
                                                       void* ws, size_t ws_bytes, void* stream) {
    return dgn::lin::combine_backward_weight_bias_pick(n_rows, n_towers, n_scalers, f_out, k, gy, stride_gy, scale, a, stride_a, dw, lddw, stride_dw, g_sum,
                                                       nullptr, 0, ws, ws_bytes, stream);
}

// ... with scaler slot `pick_slot`'s block of the column sums also written densely to pick [T][f_out] by the finalize kernel (library-internal:
// dgn_towers_layer_backward's d b_post)
int dgn::lin::combine_backward_weight_bias_pick(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k, const float* gy,
                                                int64_t stride_gy, const float* scale, const float* a, int64_t stride_a, float* dw, int64_t lddw,
                                                int64_t stride_dw, float* g_sum, float* pick, int32_t pick_slot, void* ws, size_t ws_bytes,
                                                void* stream) {
    const char* fn = "dgn_linear_combine_backward_weight";
    if (g_sum && k % 16 == 0) { set_error("%s: the column sums ride in A's padding column (k %% 16 != 0)", fn); return -1; }
    const int n = n_scalers * f_out;
    if (!expand_ok(fn, n_rows, n_towers, n_scalers, f_out, gy, stride_gy, scale)) return -1;
    if (!dgn_linear_supported(k, n, 1)) { set_error("%s: need even widths in [2, 160] and at most 45 tiles (k=%d S*f_out=%d)", fn, k, n); return -1; }
    if (!dw) { set_error("%s: null output", fn); return -1; }
    if (n_rows == 0) {
        if (pick && zero_rows_async(pick, 1, (int64_t)n_towers * f_out, (int64_t)n_towers * f_out, static_cast<hipStream_t>(stream))) return DGN_ERR_HIP;
        return zero_wgrad(dw, lddw, stride_dw, g_sum, n, k, n, n_towers, static_cast<hipStream_t>(stream));
    }
    if (!a || (stride_a & 1) || !aligned8(a)) { set_error("%s: null or misaligned operand", fn); return -1; }
    WgParams p{};
    p.M = n_rows; p.n = n; p.k = k; p.T = n_towers;
    p.X = a; p.sX = stride_a;
    if (n_scalers == 1) { p.G = gy; p.sG = stride_gy; }
    else { p.ex.gy = gy; p.ex.sT = stride_gy; p.ex.sc = scale; p.ex.S = n_scalers; p.ex.fo = f_out; }
    if (pick && (!g_sum || pick_slot < 0 || pick_slot >= n_scalers)) { set_error("%s: pick needs g_sum and a scaler slot", fn); return -1; }
    return launch_wgrad(fn, p, dw, lddw, stride_dw, g_sum, n, ws, ws_bytes, stream, pick, pick_slot * f_out, f_out);
}
