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

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return DGN_ERR_HIP;
}
}  // namespace dgn

extern "C" int dgn_abi_version(void) { return DGN_ABI_VERSION; }
extern "C" const char* dgn_last_error(void) { return dgn::g_err; }
