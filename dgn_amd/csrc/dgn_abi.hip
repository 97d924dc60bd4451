// Error plumbing and version of libdgn_hip.so.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

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

extern "C" int dgn_abi_version(void) { return DGN_ABI_VERSION; }
extern "C" const char* dgn_last_error(void) { return dgn::g_err; }
