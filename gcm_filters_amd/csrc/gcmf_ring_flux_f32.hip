// k_ring for the flux kinds with f32 state (round 5: the default evaluation of f32 scalar fields is the forward recurrence again, and
// IRREGULAR / POP / MOM5 model output is f32 more often than not).  TWO cells per lane: with four (16-byte accesses) the rings need 510
// registers and the kernel lost to k_flux_multi2 (round 3: 405 against 485 G); with two they need what the f64 kernel needs.
#include "gcmf_ring_impl.hpp"

namespace gcmf {
int launch_ring_flux_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (a.fb_is_f32) return launch_ring_k<float, float, K_FLUX, 2>(pl, a, s);
  return launch_ring_k<float, double, K_FLUX, 2>(pl, a, s);
}
}  // namespace gcmf
