// k_ringc (backward / Clenshaw evaluation, gcmf_ringc_impl.hpp) instantiations for K_REG; one translation unit per stencil kind
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_reg(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ringc_kind<K_REG>(pl, a, s); }
}  // namespace gcmf
