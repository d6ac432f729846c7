// k_ringcr: the K_REG backward kernel with the level ramp, for short strips (gcmf_ringc_impl.hpp); its own translation unit so that it
// compiles beside the others
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_reg_ramp(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) { return launch_ringc_kind<K_REG, true>(pl, a, s); }
}  // namespace gcmf
