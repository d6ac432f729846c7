// k_ringcs<float>: the f32-state flux-kind backward kernel with early exits (gcmf_ringc_impl.hpp, gcmf_ringc_flux_slab.hip), five and six
// levels; seven and eight: gcmf_ringc_flux_slab_f32b.hip (translation units that compile side by side)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_slab_f32b(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_flux_slab_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  // four cells per lane; first launches of at most seven levels, as k_ringc<float> (eight spill there)
  switch (a.S) {
    case 5: return a.first ? launch_ringc_sf<float, K_FLUX, 5, true, true>(pl, a, s) : launch_ringc_sf<float, K_FLUX, 5, false, true>(pl, a, s);
    case 6: return a.first ? launch_ringc_sf<float, K_FLUX, 6, true, true>(pl, a, s) : launch_ringc_sf<float, K_FLUX, 6, false, true>(pl, a, s);
  }
  return launch_ringc_flux_slab_f32b(pl, a, s);
}
}  // namespace gcmf
