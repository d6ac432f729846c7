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
  if (!(pl->ringc_zip && pl->zip_fold && pl->g.fold && a.row_hi == pl->g.rows && pl->d.dtype == GCMF_F64 && pl->kind == K_FLUX && pl->strip_rows <= 0 &&
        a.S >= 5 && a.S <= 9 && (pl->g.nx % 4) == 0 && pl->g.nx >= 256 && a.row_hi - a.row_lo >= 24 && a.nbatch <= 64))
    return false;
  if (a.nbatch <= 1 || !pl->pack_batch) return true;
  // a batch that may be packed: the fold strips (gridDim.y = the batch) against the packed column + k_fold_band -- in rows marched, as
  // launch_ringc weighs it (1080 x 1440 POP, 8 fields: 1167 -> 995 us; 16 fields stay packed: 2126 against 2613 us)
  const int M = (a.S + 1) / 2 * 2, WI = 128 - 2 * M;
  const long long nwx = (pl->g.nx + WI - 1) / WI, nrows = a.row_hi - a.row_lo, nfw = (pl->g.nx / 2 + WI - 1) / WI;
  long long npmax = 0, k = 1;
  for (; k <= 8 && npmax < 1; ++k) {
    const long long cap = 256 * k / std::max<long long>(1, std::min<long long>(a.nbatch, 256 * k));
    npmax = 2 * cap > nfw ? (2 * cap - nfw) / nwx : 0;
  }
  if (npmax < 1) return false;
  const long long rounds = k - 1, np = std::max(1LL, std::min(npmax, (nrows - a.S) / 4));
  const long long fold_rows = std::max<long long>(a.S, (nrows + 2 * np) / (2 * np + 1)), H = std::max(fold_rows, (nrows - fold_rows + 2 * np - 1) / (2 * np));
  const double zip = (double)(rounds * ringc_zip_rows(H + a.S + 1, a.S, nullptr)) * (1.0 + 0.04 * (rounds - 1));
  return zip <= ringc_batch_cost(nwx, a.nbatch, nrows - a.S, std::min(a.S, 8), 12, true);
}
}  // namespace gcmf
