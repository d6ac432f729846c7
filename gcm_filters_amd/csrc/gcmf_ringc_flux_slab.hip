// k_ringcs: the flux-kind backward kernel with early exits, for row slabs without a tripole seam (gcmf_ringc_impl.hpp); its own
// translation unit so that it compiles beside the others
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_slab(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype != GCMF_F64) return launch_ringc_flux_slab_f32(pl, a, s);   // (its own translation unit: gcmf_ringc_flux_slab_f32.hip)
  switch (a.S) {
    case 5: return a.first ? launch_ringc_sf<double, K_FLUX, 5, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 5, false, true>(pl, a, s);
    case 6: return a.first ? launch_ringc_sf<double, K_FLUX, 6, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 6, false, true>(pl, a, s);
    case 7: return a.first ? launch_ringc_sf<double, K_FLUX, 7, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 7, false, true>(pl, a, s);
    case 8: return a.first ? launch_ringc_sf<double, K_FLUX, 8, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 8, false, true>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}
}  // namespace gcmf
