// Instantiations of one kernel family of dgn_linear_kernels.hpp (own translation unit: they compile in parallel).
#include "dgn_linear_kernels.hpp"

namespace dgn {
namespace lin {

hipError_t launch_linear_actm(int nt, int kb, const LinParams& p, int threads, size_t lds, hipStream_t st) {
    return launch_linear_grid<kActMask>(nt, kb, p, threads, lds, st);
}

}  // namespace lin
}  // namespace dgn
