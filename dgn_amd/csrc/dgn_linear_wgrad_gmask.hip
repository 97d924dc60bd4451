// Instantiations of one kernel family of dgn_linear_kernels.hpp (own translation unit: they compile in parallel): the weight gradient
// whose G operand is masked by an activation's derivative while it is staged (WgParams.g_mask, round 6).
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

hipError_t launch_wgrad_gmask(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st) {
    return launch_wgrad_grid<false, true>(nt, kt, p, lds, st);
}

}  // namespace lin
}  // namespace dgn
