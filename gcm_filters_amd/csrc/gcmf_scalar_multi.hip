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
int launch_ring_reg(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ring_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ring_flux(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);

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

// S in 5..8; the first launch of a filter only if the caller fixes up the isolated cells afterwards (ring_first); land-mask kinds only once land is kept out of
// the state, a row of zeros at hand for closed boundaries, and fbar NOT accumulated in place: a strip that meets a NaN /
// inf is redone from its inputs, which its own stores must not have touched
bool ring_supported(const gcmf_plan *pl, const MultiArgs &a) {
  if (!pl->ring || !pl->zero_row || a.S < 5 || a.S > 8 || a.fb_in == a.fb_out) return false;
  if (a.first) {
    // the first launch of a filter: land is zeroed as it is loaded (the caller says it will fix those cells up: ring_first),
    // which needs the plan's byte plane
    if (!a.ring_first) return false;
    if (pl->kind == K_MASK && !(pl->lbits && pl->n_land > 0)) return false;  // MASKZ stencil: land must really be zero
    if (pl->kind == K_FLUX && pl->n_land > 0 && !pl->lbits) return false;
  } else if (pl->kind == K_MASK) {
    return a.land_zero != 0;
  }
  if (pl->kind == K_MASK) return true;
  if (pl->kind == K_FLUX) return pl->d.dtype == GCMF_F64 || pl->ring_flux_f32;  // (f32: two cells per lane, gcmf_ring_flux_f32.hip)
  return pl->kind == K_REG;
}

int launch_scalar_multi(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  // deep flux launches: the two-rows-per-iteration kernel (gcmf_flux_multi2.hip); GCMF_FLUX2=0 keeps the one-row form
  static const bool flux2 = !(getenv("GCMF_FLUX2") && atoi(getenv("GCMF_FLUX2")) == 0);
  // deep launches after the first: the static-ring kernels (gcmf_ring_impl.hpp); GCMF_RING=0 keeps the general ones
  if (ring_supported(pl, a)) {
    switch (pl->kind) {
      case K_REG: return launch_ring_reg(pl, a, s);
      case K_MASK: return launch_ring_maskz(pl, a, s);
      case K_FLUX: return launch_ring_flux(pl, a, s);
    }
  }
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
