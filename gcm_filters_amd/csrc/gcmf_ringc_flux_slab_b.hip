// k_ringcs<double> at seven and eight levels (see gcmf_ringc_flux_slab.hip)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_slab_b(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 7: return a.first ? launch_ringc_sf<double, K_FLUX, 7, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 7, false, true>(pl, a, s);
    case 8: return a.first ? launch_ringc_sf<double, K_FLUX, 8, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 8, false, true>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}
}  // namespace gcmf
