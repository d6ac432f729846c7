// k_ringc (backward / Clenshaw evaluation, gcmf_ringc_impl.hpp) instantiations for K_MASKZ; one translation unit per stencil kind
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ringc_kind<K_MASKZ>(pl, a, s); }
}  // namespace gcmf
