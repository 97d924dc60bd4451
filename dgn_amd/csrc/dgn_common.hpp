// Shared host/device helpers of libdgn_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgn_hip.h"

namespace dgn {

constexpr int kWave = 64;
#ifndef DGN_WAVES_PER_BLOCK
#define DGN_WAVES_PER_BLOCK 1
#endif
constexpr int kWavesPerBlock = DGN_WAVES_PER_BLOCK;   // one wave per workgroup: a finished row frees its wave slot at once (+9 % on power-law rows)
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kXcds = 8;                         // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

void set_error(const char* fmt, ...);
// Library options (dgn_set_option / dgn_get_option of the C ABI): process-wide switches the tests and experiments flip.  Each is
// initialised ONCE from its environment variable when the library first looks (no getenv on any launch path) and changed only through
// the setter.  -1 = "auto" where the library has a rule of its own.
enum Opt { OPT_BLK_LDS_KB, OPT_BLK_MIN_NODES, OPT_BWD_ROWS_PER_WAVE, OPT_TILE_GEMM, OPT_TILE_WGRAD, OPT_NO_ZMASK, OPT_LINEAR_SMALL_MIN_WAVES, OPT_GRAPH_BWD_TILES, OPT_ODD_DIRECT, OPT_BN_FROM_WGRAD, OPT_MIX_BWD_FUSED, OPT_BLK_LDS_PAD_KB, OPT_LIN_WREG, OPT_BD_BWD_FUSED, OPT_BN_STATS_FUSED, OPT_COUNT };
int64_t option(Opt o);
int hip_fail(hipError_t e, const char* what);
int zero_rows_async(float* p, int64_t rows, int64_t width, int64_t ld, hipStream_t stream);   // capture-safe zero fill (dgn_abi.hip)
// dgn_scale_combine_backward with the bias gradient WRITTEN instead of accumulated (set_bias != 0): the whole-layer calls (dgn_combine.hip)
int scale_combine_backward_impl(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* g_y, int64_t ld_gy, const float* scale,
                                const float* row_scale, float* g_z, float* g_bias, void* ws, size_t ws_bytes, const DgnBnGrad* bn,
                                void* stream, int set_bias);

// dgn_scale_combine_forward with BatchNorm's column partials of y riding in the pass (dgn_combine.hip)
int scale_combine_forward_stats(int64_t n_nodes, int32_t T, int32_t S, int32_t fo, const float* z, const float* scale, const float* bias,
                                const float* row_scale, float* y, int64_t ld_y, double* part, size_t part_bytes, int* slots, void* stream);
// dgn_bn_tail_forward + the BatchNorm modules' num_batches_tracked counters (dgn_bn_tail.hip)
int bn_tail_forward_nbt(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, int32_t training, int32_t relu, const float* residual, float* y,
                        float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, const int64_t* n_valid, int64_t* nbt, int32_t n_nbt,
                        void* stream);

// BatchNorm's finalize kernel alone (mean / invstd / running statistics / counters from G slots of column partials laid out as bn_stats
// leaves them: part[(q F + c) G + g]) -- for a producer that computed the partials itself (lin::combine_forward_stats)
// ... followed by the apply pass (dgn_bn_tail_forward's training mode without its statistics kernel)
int bn_tail_forward_from_partials(int64_t n_rows, int32_t F, int32_t G, const double* part, const float* x, int64_t ld, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, int32_t relu, const float* residual, float* y,
                                  float* save_mean, float* save_invstd, int64_t* nbt, int32_t n_nbt, void* stream);
int bn_finalize_launch(int64_t n_rows, int32_t F, int32_t G, const double* part, float* running_mean, float* running_var, float momentum, float eps,
                       float* save_mean, float* save_invstd, int64_t* nbt, int32_t n_nbt, void* stream);

// input gradient + weight gradient of the towers' block-diagonal pretrans product in one pass (dgn_linear_bd.hip: bd_backward_both)
namespace dc {
// dgn_dc_gemm with BatchNorm's column partials of C riding in the epilogue (dgn_dc.hip); part == NULL: dgn_dc_gemm itself
size_t gemm_stats_bytes(int32_t n);
int gemm_stats(const DgnDegreeClasses* d, int32_t k, int32_t n, int32_t towers, const float* a, int64_t lda, int64_t a_tower, const float* w, int64_t ldw,
               int64_t class_stride, int64_t w_tower, const float* bias, const float* row_scale, float* c, int64_t ldc, int64_t c_tower, int32_t stream_out,
               double* part, size_t part_bytes, int* slots, void* stream);
}
namespace lin {
// dgn_linear_combine_forward with BatchNorm's column partials of y riding in the epilogue (LinParams.bn_part); *groups = the number of slots
// written.  DGN_ERR_UNSUPPORTED-like return 1 (nothing launched, no error set) where the shape has no such instance: the caller runs the two passes.
size_t combine_forward_stats_bytes(int32_t n_towers, int32_t f_out);
int combine_forward_stats(int64_t n_rows, int32_t k, int32_t n_towers, int32_t n_scalers, int32_t f_out, const float* a, int64_t stride_a,
                          const float* w, int64_t ldw, int64_t stride_w, const float* scale, const float* bias, const float* row_scale, float* y,
                          int64_t ld_y, double* part, size_t part_bytes, int* groups, void* stream);
int combine_backward_weight_bias_pick(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k, const float* gy,
                                      int64_t stride_gy, const float* scale, const float* a, int64_t stride_a, float* dw, int64_t lddw,
                                      int64_t stride_dw, float* g_sum, float* pick, int32_t pick_slot, void* ws, size_t ws_bytes, void* stream);
int bd_backward_both_launch(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* g, const float* w, int64_t ldw, const float* x,
                            const float* add1, const float* add2, float* g_h, float* dw, int64_t lddw, float* dbias, void* ws, size_t ws_bytes,
                            void* stream);
}

#define DGN_HIP_CHECK(expr)                                         \
    do {                                                            \
        hipError_t _e = (expr);                                     \
        if (_e != hipSuccess) return ::dgn::hip_fail(_e, #expr);    \
    } while (0)

// ---- device helpers --------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float bcast_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (observed, speed only).
// Give every XCD one contiguous eighth of the logical work so that neighbouring rows -- whose
// sources overlap in batched small graphs -- share one L2.  Returns -1 for padding blocks.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n_logical) {
#ifdef DGN_EXP_NOREMAP
    return b < n_logical ? b : -1;
#endif
    int64_t per = (n_logical + kXcds - 1) / kXcds;
    int64_t logical = (b % kXcds) * per + b / kXcds;
    return logical < n_logical ? logical : -1;
}
inline int64_t xcd_grid(int64_t n_logical) { return ((n_logical + kXcds - 1) / kXcds) * kXcds; }

template <int VEC>
__device__ __forceinline__ void ldv(float (&d)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    } else if constexpr (VEC == 2) {
        float2 t = *reinterpret_cast<const float2*>(p);
        d[0] = t.x; d[1] = t.y;
    } else {
        d[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
    } else {
        *p = d[0];
    }
}
// Streaming ("nontemporal", `global_store ... nt`) stores for tensors that are written once and read by a LATER kernel: they do not
// allocate in L2 on the way out.  Measured on the forward sweep's 470 MB of aggregate rows: 0.210 -> 0.155 ms.  DGN_NO_NT_STORES
// (compile time) turns them into plain stores.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
// fire-and-forget fp64 add to an LDS cell (ds_add_f64: no register for a result, nothing to wait for)
__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    __hip_atomic_fetch_add((__attribute__((address_space(3))) double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void st_stream(float4* p, const float4& v) {
#ifdef DGN_NO_NT_STORES
    *p = v;
#else
    nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
#endif
}
__device__ __forceinline__ void st_stream(float2* p, const float2& v) {
#ifdef DGN_NO_NT_STORES
    *p = v;
#else
    nt_f2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f2*>(p));
#endif
}
__device__ __forceinline__ void st_stream(float* p, float v) {
#ifdef DGN_NO_NT_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
template <int VEC>
__device__ __forceinline__ void stv_stream(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 4) st_stream(reinterpret_cast<float4*>(p), make_float4(d[0], d[1], d[2], d[3]));
    else if constexpr (VEC == 2) st_stream(reinterpret_cast<float2*>(p), make_float2(d[0], d[1]));
    else st_stream(p, d[0]);
}
template <int VEC>
__device__ __forceinline__ void ldvi(int (&d)[VEC], const int* p) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = p[i];
}
template <int VEC>
__device__ __forceinline__ void stvi(int* p, const int (&d)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = d[i];
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    // hardware global_atomic_add_f32 (no CAS loop); result unused
    __builtin_amdgcn_global_atomic_fadd_f32(p, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}

}  // namespace dgn
