// Instantiations of one kernel family of dgn_linear_kernels.hpp (own translation unit: they compile in parallel): the mixing network's
// input gradient with BatchNorm's backward and the graph norm in its epilogue, tower-major output (kActMaskBnb, round 6).
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

hipError_t launch_linear_bnb(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st) {
    return launch_linear_grid<kActMaskBnb>(nt, kb, p, threads, lds, st);
}

}  // namespace lin
}  // namespace dgn
