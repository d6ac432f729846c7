// Private to the gcmf_api*.hip translation units (round 6: gcmf_api.hip was one file of 1 600 lines): what they share.
//   gcmf_api.hip          plan lifetime, gcmf_apply / gcmf_laplacian (the whole-polynomial drivers, host paths)
//   gcmf_api_blocks.hip   the row-slab building blocks and drivers (gcmf_cheb_*, gcmf_slab_apply_backward*, land helpers, resident levels)
//   gcmf_api_options.hip  tuning, named options, instrumentation and the per-launch timing events
#pragma once
#include "gcmf_internal.hpp"

namespace gcmf {
// gcmf_api.hip
int step_dispatch(gcmf_plan *pl, const StepArgs &a, hipStream_t s);
int ensure_dev_p(gcmf_plan *pl, const double *p, int n_steps, hipStream_t s);
// gcmf_api_options.hip: an event pair around every launch of the dominant kernel (gcmf_set_timing(plan, 2))
int dom_begin(gcmf_plan *pl, hipStream_t s);
int dom_end(gcmf_plan *pl, hipStream_t s);
int dom_collect(gcmf_plan *pl);
// gcmf_api_blocks.hip
bool land_ok(const gcmf_plan *pl, int n_steps);
bool ringc9_ok(const gcmf_plan *pl);
int clenshaw_cut(const gcmf_plan *pl, int n_steps, int *depths, int max_depths, bool f32_asked = false, int64_t nbatch = 1);   // (nbatch: a lone field on a cache-resident grid may be cut into shallower launches)
bool ptr_al16(const void *p);
int vec_backward_next_depth(const gcmf_plan *pl, int64_t nbatch, int left, int smax);
}  // namespace gcmf
