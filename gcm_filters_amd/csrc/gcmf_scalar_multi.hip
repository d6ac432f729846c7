// Temporally blocked scalar kernels: dispatch.  The kernel template lives in gcmf_scalar_multi_impl.hpp and is
// instantiated per stencil kind in gcmf_scalar_multi_{reg,mask,maskz,flux}.hip (four translation units compile in
// parallel; one unit took over two minutes).
#include "gcmf_internal.hpp"

#include <cstdlib>

namespace gcmf {

int launch_multi_reg(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_multi_mask(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_multi_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_multi_flux(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);

bool multi_supported(const gcmf_plan *pl, int S) {
  if (pl->ncomp != 1) return false;
  if (S < 2 || S > 8) return false;
  const int vec = pl->d.dtype == GCMF_F64 ? 2 : 4;
  if (pl->g.nx % vec) return false;
  // the tripole seam couples mirrored columns: its top S rows are advanced by single steps (advance_multi)
  if (pl->g.fold && pl->g.rows < 3 * S + 2) return false;
  if (pl->g.rows < S + 2) return false;  // the march wraps row indices with one conditional add (needs |r| < rows)
  return true;
}

int launch_scalar_multi(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  // deep flux launches: the two-rows-per-iteration kernel (gcmf_flux_multi2.hip); GCMF_FLUX2=0 keeps the one-row form
  static const bool flux2 = !(getenv("GCMF_FLUX2") && atoi(getenv("GCMF_FLUX2")) == 0);
  if (flux2 && flux_multi2_supported(pl, a.S)) return launch_flux_multi2(pl, a, s);
  switch (pl->kind) {
    case K_REG: return launch_multi_reg(pl, a, s);
    case K_MASK: return a.land_zero ? launch_multi_maskz(pl, a, s) : launch_multi_mask(pl, a, s);
    case K_FLUX: return launch_multi_flux(pl, a, s);
  }
  set_error("launch_scalar_multi: plan is not a scalar kind");
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
