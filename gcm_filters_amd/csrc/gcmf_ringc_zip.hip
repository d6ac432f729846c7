// k_ringcz<double>: pairs of strips zipped at a shared seam (gcmf_ringc_impl.hpp), nine levels (with early exits and in whole ring periods); eight, seven, six and five: gcmf_ringc_zip_{b,c,d}.hip
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_zip_b(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_zip(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype != GCMF_F64 || pl->kind != K_FLUX) return GCMF_ERR_INVALID_ARG;
  switch (a.S) {
    case 9: return a.first ? launch_ringc_zip_sf<double, 9, true>(pl, a, s) : launch_ringc_zip_sf<double, 9, false>(pl, a, s);
  }
  return launch_ringc_zip_b(pl, a, s);
}
int ringc_zip_march(const gcmf_plan *pl, const MultiArgs &a, int *pairs) {
  if (!pl->ringc_zip || pl->d.dtype != GCMF_F64 || pl->kind != K_FLUX || (pl->g.fold && !pl->alone_now) || pl->strip_rows > 0 || a.S < 5 || a.S > 9) return 0;
  const int M = (a.S + 1) / 2 * 2, WI = 128 - 2 * M;
  int march = 0;
  const int np = ringc_zip_pairs((pl->g.nx + WI - 1) / WI, a.nbatch, a.row_hi - a.row_lo, a.S, &march);
  if (pairs) *pairs = np;
  return np >= 1 ? march : 0;
}
// The tripole seam inside the launch (round 6): whole-launch conditions for k_ringcz's fold strips (gcmf_ringc_impl.hpp) -- the f64 flux
// kind evaluated backwards, rows up to the seam, a lane's two cells on one side of the row's centre, no packed batch.
bool ringc_zip_fold_ok(const gcmf_plan *pl, const MultiArgs &a) {
  return pl->ringc_zip && pl->zip_fold && pl->g.fold && a.row_hi == pl->g.rows && pl->d.dtype == GCMF_F64 && pl->kind == K_FLUX && pl->strip_rows <= 0 &&
         a.S >= 5 && a.S <= 9 && (pl->g.nx % 4) == 0 && pl->g.nx >= 256 && a.row_hi - a.row_lo >= 24 && !(a.nbatch > 1 && pl->pack_batch) && a.nbatch <= 64;
}
}  // namespace gcmf
