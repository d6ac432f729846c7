// k_ringcs<float>: the f32-state flux-kind backward kernel with early exits (gcmf_ringc_impl.hpp, gcmf_ringc_flux_slab.hip); its own translation
// unit so that it compiles beside the f64 one
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux_slab_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  // four cells per lane; first launches of at most seven levels, as k_ringc<float> (eight spill there)
  switch (a.S) {
    case 5: return a.first ? launch_ringc_sf<float, K_FLUX, 5, true, true>(pl, a, s) : launch_ringc_sf<float, K_FLUX, 5, false, true>(pl, a, s);
    case 6: return a.first ? launch_ringc_sf<float, K_FLUX, 6, true, true>(pl, a, s) : launch_ringc_sf<float, K_FLUX, 6, false, true>(pl, a, s);
    case 7: return a.first ? launch_ringc_sf<float, K_FLUX, 7, true, true>(pl, a, s) : launch_ringc_sf<float, K_FLUX, 7, false, true>(pl, a, s);
    case 8:
      if (a.first) break;
      return launch_ringc_sf<float, K_FLUX, 8, false, true>(pl, a, s);
  }
  set_error("k_ringcs<float>: depth %d%s is not offered", a.S, a.first ? " as a first launch" : "");
  return GCMF_ERR_INVALID_ARG;
}
}  // namespace gcmf
