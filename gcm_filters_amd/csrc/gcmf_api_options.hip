// libgcmf C ABI, part 3: tuning, named options (A/B switches), instrumentation (which kernel / which path ran, launch timing) and the
// event pairs gcmf_apply puts around the launches of the dominant kernel.  See include/gcmf.h for the contract.
#include "gcmf_api_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace gcmf {

// gcmf_set_timing(plan, 2): bracket a blocked launch with its own event pair on the stream it runs on
int dom_begin(gcmf_plan *pl, hipStream_t s) {
  if (!pl->timing_detail) return GCMF_OK;
  if ((size_t)pl->dom_used + 2 > pl->dom_ev.size())
    for (int q = 0; q < 2; ++q) {
      hipEvent_t e;
      GCMF_HIP(hipEventCreate(&e));
      pl->dom_ev.push_back(e);
    }
  GCMF_HIP(hipEventRecord(pl->dom_ev[pl->dom_used], s));
  return GCMF_OK;
}

int dom_end(gcmf_plan *pl, hipStream_t s) {
  if (!pl->timing_detail) return GCMF_OK;
  GCMF_HIP(hipEventRecord(pl->dom_ev[pl->dom_used + 1], s));
  pl->dom_name.resize(pl->dom_ev.size() / 2);
  pl->dom_name[pl->dom_used / 2] = pl->last_launched;
  pl->dom_used += 2;
  return GCMF_OK;
}

int dom_collect(gcmf_plan *pl) {
  pl->dom_ms = pl->dom_min = pl->dom_max = 0.f;
  pl->dom_n = 0;
  // only the launches of the dominant kernel (the one gcmf_last_kernel reports): the first launch of a filter and the
  // remainder launch run other instantiations
  for (int q = 0; q + 1 < pl->dom_used; q += 2) {
    float ms = 0.f;
    GCMF_HIP(hipEventSynchronize(pl->dom_ev[q + 1]));
    if (!pl->last_kernel.empty() && pl->dom_name[q / 2] != pl->last_kernel) continue;
    GCMF_HIP(hipEventElapsedTime(&ms, pl->dom_ev[q], pl->dom_ev[q + 1]));
    pl->dom_ms += ms;
    pl->dom_min = pl->dom_n ? std::min(pl->dom_min, ms) : ms;
    pl->dom_max = std::max(pl->dom_max, ms);
    ++pl->dom_n;
  }
  pl->dom_used = 0;
  return GCMF_OK;
}

}  // namespace gcmf

using namespace gcmf;

extern "C" {

int gcmf_set_timing(gcmf_plan *pl, int enabled) {
  if (!pl) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(pl->mu);
  pl->timing = enabled != 0;
  pl->timing_detail = enabled == 2;
  pl->dom_used = 0;
  return GCMF_OK;
}
int gcmf_last_kernel_timing(const gcmf_plan *pl, float *ms_sum, int *n_launches, float *ms_min, float *ms_max) {
  if (!pl) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(const_cast<gcmf_plan *>(pl)->mu);   // dom_* are written by a running gcmf_apply
  if (ms_sum) *ms_sum = pl->dom_ms;
  if (n_launches) *n_launches = pl->dom_n;
  if (ms_min) *ms_min = pl->dom_min;
  if (ms_max) *ms_max = pl->dom_max;
  return GCMF_OK;
}
int gcmf_last_timing(const gcmf_plan *pl, float *ms_total, int *n_launches) {
  if (!pl) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(const_cast<gcmf_plan *>(pl)->mu);
  if (ms_total) *ms_total = pl->last_ms;
  if (n_launches) *n_launches = pl->last_launches;
  return GCMF_OK;
}
int gcmf_last_kernel(gcmf_plan *pl, char *buf, int n) {
  if (!pl || !buf || n < 1) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(pl->mu);
  snprintf(buf, (size_t)n, "%s", pl->last_kernel.c_str());
  pl->last_kernel.clear();
  pl->last_kernel_weight = 0;
  return GCMF_OK;
}
int gcmf_last_kernel_geometry(gcmf_plan *pl, char *buf, int n) {
  if (!pl || !buf || n < 1) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(pl->mu);
  snprintf(buf, (size_t)n, "%s", pl->last_geom.c_str());
  return GCMF_OK;
}
int gcmf_ring_fallbacks(gcmf_plan *pl, int64_t *count) {
  if (!pl || !count) return GCMF_ERR_INVALID_ARG;
  *count = 0;
  if (!pl->ring_nfb) return GCMF_OK;
  unsigned n = 0;
  std::lock_guard<std::mutex> lk(pl->mu);          // not while an apply of this plan is enqueueing
  GCMF_HIP(hipSetDevice(pl->d.device));            // the plan's device, not whichever is current in this thread
  GCMF_HIP(hipDeviceSynchronize());                // callers may run the plan on any stream of that device
  GCMF_HIP(hipMemcpy(&n, pl->ring_nfb, sizeof n, hipMemcpyDeviceToHost));
  GCMF_HIP(hipMemset(pl->ring_nfb, 0, sizeof n));
  *count = n;
  return GCMF_OK;
}
int gcmf_set_tuning(gcmf_plan *pl, int rows_per_wave, int xcd_remap, int multi_s) {
  if (!pl) return GCMF_ERR_INVALID_ARG;
  if (rows_per_wave > 0) pl->rows_per_wave = rows_per_wave;
  if (xcd_remap >= 0) {
    pl->xcd_remap = xcd_remap & 1;
    if ((xcd_remap >> 1) & 3) pl->zigzag = ((xcd_remap >> 1) & 3) - 1;
  }
  if (multi_s > 0) {
    pl->multi_s = multi_s & 0xFF;               // low byte: steps per pass
    pl->strip_rows = (multi_s >> 8) & 0xFFFF;   // bits 8..23: rows per strip (0 = auto)
    pl->prefetch_rows = (multi_s >> 24) & 0xF;  // bits 24..27: operand rows in flight per wave (0 = default)
    if ((multi_s >> 28) & 3) pl->clenshaw = ((multi_s >> 28) & 3) - 1;  // bits 28..29: backward evaluation 1 = off, 2 = flux kinds, 3 = all
  }
  return GCMF_OK;
}

int gcmf_set_option(gcmf_plan *pl, const char *name, int value) {
  if (!pl || !name) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(pl->mu);
  const std::string n(name);
  if (n == "cgrid_ring") pl->cgrid_ring = value;
  else if (n == "cgrid_ring_smax") pl->cgrid_ring_smax = value;
  else if (n == "cgrid_ring_hmax") pl->cgrid_ring_hmax = value;
  else if (n == "cgrid_ring_ncarry") pl->cgrid_ring_ncarry = value;
  else if (n == "pack_batch") pl->pack_batch = value;
  else if (n == "single_launch") pl->single_launch = value;
  else if (n == "ringc9") pl->ringc9 = value;
  else if (n == "ringc_zip") pl->ringc_zip = value;
  else if (n == "ringc_smax") pl->ringc_smax = value;
  else if (n == "band_seq_cells") pl->band_seq_cells = value;
  else if (n == "zip_fold") pl->zip_fold = value;
  else if (n == "slab_nines") pl->slab_nines = value;
  else if (n == "clenshaw_f32") pl->clenshaw_f32 = value;
  else if (n == "ring_flux_f32") pl->ring_flux_f32 = value;
  else {
    set_error("gcmf_set_option: unknown option '%s'", name);
    return GCMF_ERR_INVALID_ARG;
  }
  return GCMF_OK;
}

// ---- the on-chip (resident) kernel, gcmf_resident.hip --------------------------------------------------------------------------
// Whether L levels of the backward evaluation with output rows [row_lo, row_hi) of this plan can run in ONE resident launch (f64 scalar
// plans whose rows [row_lo - L, row_hi + L) fit the register files + LDS of the chip, no tripole seam in that range, L <= 64).
int gcmf_plan_last_path(const gcmf_plan *pl, int *path, int64_t *counts) {
  if (!pl) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(const_cast<gcmf_plan *>(pl)->mu);
  if (path) *path = pl->last_path;
  if (counts)
    for (int k = 0; k < 5; ++k) counts[k] = pl->path_count[k];
  return GCMF_OK;
}

int gcmf_resident_status(int device, int *state, uint64_t *failures) {
  unsigned long long nf = 0;
  int st = GCMF_RESIDENT_OFF;
  resident_status(device, &st, &nf);
  if (state) *state = st;
  if (failures) *failures = (uint64_t)nf;
  return GCMF_OK;
}

}  // extern "C"
