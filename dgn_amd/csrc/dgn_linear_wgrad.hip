// Instantiations of one kernel family of dgn_linear_kernels.hpp (own translation unit: they compile in parallel).
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

hipError_t launch_wgrad_plain(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st) {
    return launch_wgrad_grid<false>(nt, kt, p, lds, st);
}

}  // namespace lin
}  // namespace dgn
