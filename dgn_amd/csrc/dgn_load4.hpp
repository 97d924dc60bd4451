// Branch-free 16-byte operand loads shared by the GEMM kernels (dgn_gemm_kernels.hpp, dgn_dc_kernels.hpp).
#pragma once
#include <hip/hip_runtime.h>

namespace dgn {
namespace gemm {

using f4 = __attribute__((ext_vector_type(4))) float;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte vector at 4-byte alignment (odd row strides)

// Four consecutive floats row[col .. col + 3] of a row that has `limit` (>= 4) columns, zero past the end, WITHOUT a branch: one
// 16-byte load from an address clamped into the row, the wanted window picked with selects.  A load inside a branch is followed by
// s_waitcnt vmcnt(0) at the branch's end (measured in these kernels' ISA: every operand prefetch was a synchronous memory round trip
// in front of the chunk's MFMAs instead of travelling during them); `live` = false gives zeros (rows past the matrix: the caller
// clamps the row pointer).
// Split in two so that NOTHING consumes the load where it is issued: load4_raw (the load + the window's shift) goes with the prefetch,
// load4_window (the selects) with the commit after the MFMAs -- selects right behind the load would again put a wait in front of the
// MFMAs.
struct Raw4 { f4 raw; int sh; };
__device__ __forceinline__ Raw4 load4_raw(const float* row, int col, int limit, bool live = true) {
    const int cc = max(0, min(col, limit - 4));
    Raw4 r;
    r.raw = *reinterpret_cast<const f4u*>(row + cc);
    r.sh = live ? col - cc : 4;                  // 0..3 inside the row, >= 4: nothing of the window exists
    return r;
}
// The window of a raw load.  Almost every piece of a tall matrix is a whole aligned window (sh == 0 in every lane: only the last k chunk
// of a ragged width and rows past the matrix differ), so a WAVE-UNIFORM test returns the raw registers untouched; the general case is
// written with integer masks -- as nested ?: the compiler turned it into a nest of exec-mask branches (~40 scalar / vector instructions
// and a dozen s_cbranch per window, six windows per chunk behind every MFMA block: found in the ISA of dc_gemm, round 3).
__device__ __forceinline__ f4 load4_window(const Raw4& r) {
    const f4 raw = r.raw;
    const int sh = r.sh;
    if (__builtin_amdgcn_ballot_w64(sh != 0) == 0) return raw;
    const unsigned e0 = 0u - (unsigned)(sh == 0), e1 = 0u - (unsigned)(sh == 1), e2 = 0u - (unsigned)(sh == 2), e3 = 0u - (unsigned)(sh == 3);
    const unsigned b0 = __float_as_uint(raw[0]), b1 = __float_as_uint(raw[1]), b2 = __float_as_uint(raw[2]), b3 = __float_as_uint(raw[3]);
    f4 v;
    v[0] = __uint_as_float((b0 & e0) | (b1 & e1) | (b2 & e2) | (b3 & e3));
    v[1] = __uint_as_float((b1 & e0) | (b2 & e1) | (b3 & e2));
    v[2] = __uint_as_float((b2 & e0) | (b3 & e1));
    v[3] = __uint_as_float(b3 & e0);
    return v;
}
__device__ __forceinline__ f4 load4_nb(const float* row, int col, int limit, bool live = true) { return load4_window(load4_raw(row, col, limit, live)); }

}  // namespace gemm
}  // namespace dgn
