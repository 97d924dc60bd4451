// Instantiations of the block backward (dgn_agg_block.hpp) for VEC = 2 floats per lane, hot lists only.
#include "dgn_agg_block.hpp"
#include "dgn_agg_graph.hpp"

namespace dgn {
int launch_agg_block_v2(const AggParams& p, int gap, hipStream_t stream) { return launch_block_vec<2>(p, gap, stream); }
int launch_agg_graph_v2(const AggParams& p, hipStream_t stream) { return launch_graph_v2(p, stream); }
}  // namespace dgn
