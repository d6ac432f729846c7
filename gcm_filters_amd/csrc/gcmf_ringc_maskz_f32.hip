// k_ringc<float, K_MASKZ, ...>: the f32-state instantiations of gcmf_ringc_maskz.hip, in their own translation unit
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_maskz_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ringc_kind_f32<K_MASKZ>(pl, a, s); }
}  // namespace gcmf
