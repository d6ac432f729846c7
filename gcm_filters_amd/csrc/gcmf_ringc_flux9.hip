// k_ringc<double, K_FLUX, 9, *>: NINE levels per launch for whole f64 flux-form grids without a tripole seam (round 5; its own translation
// unit: the nine-level body compiles for a minute).  The ring period R = 12 already admits S + D = 9 + 3 rows; a level costs 16 registers
// (418 of 512); BASELINE config 3's 63-level polynomial becomes 7 passes over HBM instead of 8.
// (The early-exit form k_ringcs<double, 9> was measured too: a strip of the BASELINE grid would march 100 rows instead of 108, but at 486
// registers the launch takes 143-149 us against 132-135 us: 570-578 G against 596-599 G on the same box.  Not built.)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_flux9(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype != GCMF_F64 || a.S != 9) return GCMF_ERR_INVALID_ARG;
  return a.first ? launch_ringc_sf<double, K_FLUX, 9, true>(pl, a, s) : launch_ringc_sf<double, K_FLUX, 9, false>(pl, a, s);
}
}  // namespace gcmf
