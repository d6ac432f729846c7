// k_ringcs: the flux-kind backward kernel with early exits, for row slabs without a tripole seam (gcmf_ringc_impl.hpp): f64 state, five and
// six levels; seven and eight: gcmf_ringc_flux_slab_b.hip; f32 state: gcmf_ringc_flux_slab_f32[b].hip (translation units that compile side by side)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_slab_b(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_flux_slab(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype != GCMF_F64) return launch_ringc_flux_slab_f32(pl, a, s);
  switch (a.S) {
    case 5: return a.first ? launch_ringc_sf<double, K_FLUX, 5, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 5, false, true>(pl, a, s);
    case 6: return a.first ? launch_ringc_sf<double, K_FLUX, 6, true, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 6, false, true>(pl, a, s);
  }
  return launch_ringc_flux_slab_b(pl, a, s);
}
}  // namespace gcmf
