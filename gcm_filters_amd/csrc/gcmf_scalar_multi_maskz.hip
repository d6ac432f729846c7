// k_scalar_multi instantiations of one stencil kind (see gcmf_scalar_multi.hip)
#include "gcmf_scalar_multi_impl.hpp"

namespace gcmf {
int launch_multi_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_multi_kind<K_MASKZ>(pl, a, s); }
}  // namespace gcmf
