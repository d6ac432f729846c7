// k_ringcz<double> at eight levels (see gcmf_ringc_zip.hip)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_zip_c(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_zip_b(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (a.S == 8) return a.first ? launch_ringc_zip_sf<double, 8, true>(pl, a, s) : launch_ringc_zip_sf<double, 8, false>(pl, a, s);
  return launch_ringc_zip_c(pl, a, s);
}
}  // namespace gcmf
