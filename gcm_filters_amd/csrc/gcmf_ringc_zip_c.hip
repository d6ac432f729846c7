// k_ringcz<double> at seven levels (see gcmf_ringc_zip.hip)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_zip_d(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_zip_c(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (a.S == 7) return a.first ? launch_ringc_zip_sf<double, 7, true>(pl, a, s) : launch_ringc_zip_sf<double, 7, false>(pl, a, s);
  return launch_ringc_zip_d(pl, a, s);
}
}  // namespace gcmf
