// k_ring instantiations for K_FLUX (see gcmf_ring_impl.hpp); one translation unit per stencil kind so that they compile in parallel
#include "gcmf_ring_impl.hpp"

namespace gcmf {
int launch_ring_flux(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ring_kind<K_FLUX>(pl, a, s); }
}  // namespace gcmf
