// k_ringc (backward / Clenshaw evaluation, gcmf_ringc_impl.hpp) instantiations for K_FLUX; one translation unit per stencil kind
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ringc_kind<K_FLUX>(pl, a, s); }
}  // namespace gcmf
