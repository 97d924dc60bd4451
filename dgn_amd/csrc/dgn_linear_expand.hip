// Instantiations of one kernel family of dgn_linear_kernels.hpp (own translation unit: they compile in parallel).
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

hipError_t launch_linear_expand(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st) {
    return launch_linear_grid<kExpand>(nt, kb, p, threads, lds, st);
}
hipError_t launch_wgrad_expand(int nt, int kt, const WgParams& p, size_t lds, hipStream_t st) {
    return launch_wgrad_grid<true>(nt, kt, p, lds, st);
}

}  // namespace lin
}  // namespace dgn
