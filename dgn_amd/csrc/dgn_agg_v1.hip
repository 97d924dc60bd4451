// Instantiations of the aggregation kernels for VEC = 1 floats per lane (one TU per VEC keeps the
// build parallel: 18 accumulator configurations x 7 kernels each).
#include "dgn_agg_kernels.hpp"

namespace dgn {
int launch_agg_v1(const AggParams& p, unsigned tiles, hipStream_t stream, bool backward) {
    return backward ? launch_vec<1, true>(p, tiles, stream) : launch_vec<1, false>(p, tiles, stream);
}
}  // namespace dgn
