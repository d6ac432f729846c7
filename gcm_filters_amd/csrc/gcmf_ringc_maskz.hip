// k_ringc (backward / Clenshaw evaluation, gcmf_ringc_impl.hpp) instantiations for K_MASKZ, f64 state; the f32 ones: gcmf_ringc_maskz_f32.hip
// (one translation unit per stencil kind and state type: they compile side by side)
#include "gcmf_ringc_impl.hpp"

namespace gcmf {
int launch_ringc_maskz_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  return pl->d.dtype != GCMF_F64 ? launch_ringc_maskz_f32(pl, a, s) : launch_ringc_kind_f64<K_MASKZ>(pl, a, s);
}
}  // namespace gcmf
