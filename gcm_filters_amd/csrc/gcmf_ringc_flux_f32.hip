// k_ringc<float, K_FLUX, ...>: the f32-state instantiations of gcmf_ringc_flux.hip, in their own translation unit
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ringc_kind_f32<K_FLUX>(pl, a, s); }
}  // namespace gcmf
