// k_ringcs<float> at seven and eight levels (see gcmf_ringc_flux_slab_f32.hip)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_slab_f32b(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 7: return a.first ? launch_ringc_sf<float, K_FLUX, 7, true, true>(pl, a, s) : launch_ringc_sf<float, K_FLUX, 7, false, true>(pl, a, s);
    case 8:
      if (a.first) break;
      return launch_ringc_sf<float, K_FLUX, 8, false, true>(pl, a, s);
  }
  set_error("k_ringcs<float>: depth %d%s is not offered", a.S, a.first ? " as a first launch" : "");
  return GCMF_ERR_INVALID_ARG;
}
}  // namespace gcmf
