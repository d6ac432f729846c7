// k_ring instantiations for K_MASKZ (see gcmf_ring_impl.hpp); one translation unit per stencil kind so that they compile in parallel
#include "gcmf_ring_impl.hpp"

namespace gcmf {
int launch_ring_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ring_kind<K_MASKZ>(pl, a, s); }
}  // namespace gcmf
