// k_ringcz<double> at five and six levels (see gcmf_ringc_zip.hip)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_zip_d(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 5: return a.first ? launch_ringc_zip_sf<double, 5, true>(pl, a, s) : launch_ringc_zip_sf<double, 5, false>(pl, a, s);
    case 6: return a.first ? launch_ringc_zip_sf<double, 6, true>(pl, a, s) : launch_ringc_zip_sf<double, 6, false>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}
}  // namespace gcmf
