// libgcmf C ABI: plan lifetime, the whole-polynomial apply loop and the per-step building blocks.
// See include/gcmf.h for the contract and the reference interfaces each entry point replaces.
#include "gcmf_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>

namespace gcmf {

static thread_local std::string g_err = "";

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

size_t dtype_size(int dtype) { return dtype == GCMF_F64 ? 8 : 4; }

struct GridInfo {
  int nplanes, ncomp, dimensional, tripolar, area_weighted, kind;
};
static bool grid_info(int gt, GridInfo &gi) {
  switch (gt) {
    case GCMF_REGULAR: gi = {0, 1, 0, 0, 0, K_REG}; return true;
    case GCMF_REGULAR_AREA_WEIGHTED: gi = {1, 1, 0, 0, 1, K_REG}; return true;
    case GCMF_REGULAR_WITH_LAND: gi = {1, 1, 0, 0, 0, K_MASK}; return true;
    case GCMF_REGULAR_WITH_LAND_AREA_WEIGHTED: gi = {2, 1, 0, 0, 1, K_MASK}; return true;
    case GCMF_IRREGULAR_WITH_LAND: gi = {8, 1, 1, 0, 0, K_FLUX}; return true;
    case GCMF_MOM5U: gi = {6, 1, 1, 0, 0, K_FLUX}; return true;
    case GCMF_MOM5T: gi = {6, 1, 1, 0, 0, K_FLUX}; return true;
    case GCMF_TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED: gi = {2, 1, 0, 1, 1, K_MASK}; return true;
    case GCMF_TRIPOLAR_POP_WITH_LAND: gi = {6, 1, 1, 1, 0, K_FLUX}; return true;
    case GCMF_VECTOR_C_GRID: gi = {14, 2, 1, 0, 0, K_CGRID}; return true;
    case GCMF_VECTOR_B_GRID: gi = {8, 2, 1, 0, 0, K_BGRID}; return true;
  }
  return false;
}

static int step_dispatch(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  return pl->ncomp == 1 ? launch_scalar_step(pl, a, s) : launch_vector_step(pl, a, s);
}

static int ensure_work(gcmf_plan *pl, size_t bytes) {
  if (bytes <= pl->work_bytes) return GCMF_OK;
  if (pl->work) {
    GCMF_HIP(hipFree(pl->work));
    pl->work = nullptr;
    pl->work_bytes = 0;
  }
  GCMF_HIP(hipMalloc(&pl->work, bytes));
  pl->work_bytes = bytes;
  return GCMF_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Page-locked caller ranges, process wide and reference counted: two plans (two dask threads) may stream the same host
// array at once, and the first to finish must not unregister it under the other's transfers.
static std::mutex g_reg_mu;
static std::map<const void *, std::pair<size_t, int>> g_reg;
static bool host_register(const void *p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.find(p);
  if (it != g_reg.end()) {
    if (it->second.first < bytes) return false;  // a shorter range is locked: leave this call on the pageable path
    ++it->second.second;
    return true;
  }
  if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  g_reg[p] = {bytes, 1};
  return true;
}
static void host_unregister(const void *p) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.find(p);
  if (it == g_reg.end()) return;
  if (--it->second.second == 0) {
    (void)hipHostUnregister(const_cast<void *>(p));
    g_reg.erase(it);
  }
}

// p[0..n_steps] on the device for k_land_fix; uploaded only when it changed (a pageable upload stalls the host behind
// the stream)
static int ensure_dev_p(gcmf_plan *pl, const double *p, int n_steps, hipStream_t s) {
  const size_t n = (size_t)n_steps + 1;
  if (pl->dev_p_n < n) {
    if (pl->dev_p) GCMF_HIP(hipFree(pl->dev_p));
    pl->dev_p = nullptr;
    pl->dev_p_n = 0;
    pl->host_p.clear();
    GCMF_HIP(hipMalloc((void **)&pl->dev_p, n * sizeof(double)));
    pl->dev_p_n = n;
  }
  if (pl->host_p.size() != n || memcmp(pl->host_p.data(), p, n * sizeof(double)) != 0) {
    pl->host_p.assign(p, p + n);
    GCMF_HIP(hipStreamSynchronize(s));  // nothing may still read the old coefficients
    GCMF_HIP(hipMemcpy(pl->dev_p, pl->host_p.data(), n * sizeof(double), hipMemcpyHostToDevice));
  }
  return GCMF_OK;
}

// gcmf_set_timing(plan, 2): bracket a blocked launch with its own event pair on the stream it runs on
static int dom_begin(gcmf_plan *pl, hipStream_t s) {
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
static int dom_end(gcmf_plan *pl, hipStream_t s) {
  if (!pl->timing_detail) return GCMF_OK;
  GCMF_HIP(hipEventRecord(pl->dom_ev[pl->dom_used + 1], s));
  pl->dom_name.resize(pl->dom_ev.size() / 2);
  pl->dom_name[pl->dom_used / 2] = pl->last_launched;
  pl->dom_used += 2;
  return GCMF_OK;
}
static int dom_collect(gcmf_plan *pl) {
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

// One temporally blocked advance of S steps on rows [row_lo, row_hi) of a scalar plan.
//
// Tripolar grids: the fold couples column i of the top row with column nx-1-i, i.e. with a DIFFERENT wave of the
// strip-marching kernel, which therefore stops S rows below the seam; the top S rows ("band") are advanced by k_fold_band.
static int launch_ringc(gcmf_plan *pl, const MultiArgs &m, hipStream_t s) {
  switch (pl->kind) {
    case K_REG: return launch_ringc_reg(pl, m, s);
    case K_MASK: return launch_ringc_maskz(pl, m, s);
    case K_FLUX:
      // Nothing has to fit beside the waves where there is no tripole seam -> the early-exit form (k_ringcs) wherever it shortens the march:
      // the plain form marches whole 12-row ring periods, the early-exit form leaves after every fourth row (and costs ~60 registers:
      // 1.4 % per launch in f64, ~8 % in f32).  1024 lone waves on 1080 x 1440 f64 cells own 14-row strips: 32 rows marched instead of 36,
      // and every SIMD has a wave (330 -> 364 G cell-steps/s, tools/measure_midsize.py); an 8-way slab 28 instead of 36; BASELINE-size
      // f64 grids 96 either way (-> k_ringc), BASELINE-size f32 grids 52 instead of 60 (+4 %), 1080 x 1440 f32 24 either way (-> k_ringc,
      // the early-exit form measured 9 % slower there).
      if (!pl->g.fold && m.S <= 8) {
        const bool f64 = pl->d.dtype == GCMF_F64;
        const int wi = f64 ? 112 : 240;   // useful columns of a window (f32: four cells per lane)
        const long long nrows = m.row_hi - m.row_lo;
        const long long nwx = (pl->g.nx + wi - 1) / wi, want = strips_per_column(nwx * std::max<long long>(1, m.nbatch), nrows, m.S, 12);
        const long long H0 = std::min(nrows, std::max(4LL, pl->strip_rows > 0 ? (long long)pl->strip_rows : (nrows + want - 1) / want));
        const long long need = H0 + 2 * m.S;
        const long long rows_xe = std::max(12LL, (need + 3) / 4 * 4), rows_pad = (need + 11) / 12 * 12;
        if (H0 < pl->ringc_xe_rows && rows_xe * 100 <= rows_pad * (f64 ? 95 : 90)) return launch_ringc_flux_slab(pl, m, s);
      }
      if (m.S == 9) return launch_ringc_flux9(pl, m, s);
      return launch_ringc_flux(pl, m, s);
    default: break;
  }
  set_error("k_ringc: plan is not a scalar kind");
  return GCMF_ERR_INVALID_ARG;
}

int advance_multi(gcmf_plan *pl, const MultiArgs &m, hipStream_t s, int *launches, bool backward) {
  const Geom &g = pl->g;
  const int rows = g.rows, S = m.S;
  const bool band = g.fold && m.row_hi == rows;
  int rc;
  auto blocked = [&](const MultiArgs &a) { return backward ? launch_ringc(pl, a, s) : launch_scalar_multi(pl, a, s); };
  if (!band) {
    if ((rc = dom_begin(pl, s))) return rc;
    if ((rc = blocked(m))) return rc;
    if ((rc = dom_end(pl, s))) return rc;
    if (launches) ++*launches;
    return GCMF_OK;
  }
  const int blo = rows - S;  // first band row
  // k_fold_band reads rows [rows - 2S, rows) of the input planes (valid: the caller's ghost zone covers [row_lo - S, ...)) and owns
  // [rows - S, rows); the blocked launch gets [row_lo, rows - S), possibly nothing
  if (m.row_lo > blo || rows < 2 * S) {
    set_error("advance_multi: row range [%d, %d) too short for the tripole band of %d rows", m.row_lo, m.row_hi, S);
    return GCMF_ERR_INVALID_ARG;
  }
  MultiArgs mm = m;
  mm.row_hi = blo;
  if (!fold_band_supported(pl, m)) {
    set_error("advance_multi: k_fold_band does not cover this plan / depth %d / batch %lld", S, (long long)m.nbatch);
    return GCMF_ERR_UNSUPPORTED;
  }
  // The seam rows in ONE launch (k_fold_band, gcmf_foldband.hip) on a side stream beside the blocked launch: neither reads what
  // the other writes (the band reads rows >= rows - 2S of the input planes, the two write disjoint rows of the output planes).
  if (!pl->side) {
    // both streams are on this device and nothing between fork and join is read by the host or a peer: no system-scope
    // fence on these events (agent scope orders the two queues; measured +2 % on config 4: the fork / join packets are
    // the only cost the seam has left, ~6 us per launch)
    const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
    GCMF_HIP(hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking));
    GCMF_HIP(hipEventCreateWithFlags(&pl->ev_fork, evf));
    GCMF_HIP(hipEventCreateWithFlags(&pl->ev_join, evf));
  }
  GCMF_HIP(hipEventRecord(pl->ev_fork, s));
  GCMF_HIP(hipStreamWaitEvent(pl->side, pl->ev_fork, 0));
  if ((rc = launch_fold_band(pl, m, backward, pl->side))) return rc;
  if (launches) ++*launches;
  GCMF_HIP(hipEventRecord(pl->ev_join, pl->side));
  if (mm.row_hi > mm.row_lo) {
    if ((rc = dom_begin(pl, s))) return rc;
    if ((rc = blocked(mm))) return rc;
    if ((rc = dom_end(pl, s))) return rc;
    if (launches) ++*launches;
  }
  GCMF_HIP(hipStreamWaitEvent(s, pl->ev_join, 0));
  return GCMF_OK;
}

}  // namespace gcmf

using namespace gcmf;

extern "C" {

const char *gcmf_last_error(void) { return g_err.c_str(); }
int gcmf_version(void) { return GCMF_VERSION; }

int gcmf_grid_nplanes(int gt) {
  GridInfo gi;
  return grid_info(gt, gi) ? gi.nplanes : -1;
}
int gcmf_grid_ncomp(int gt) {
  GridInfo gi;
  return grid_info(gt, gi) ? gi.ncomp : -1;
}
int gcmf_grid_is_dimensional(int gt) {
  GridInfo gi;
  return grid_info(gt, gi) ? gi.dimensional : -1;
}
int gcmf_grid_is_tripolar(int gt) {
  GridInfo gi;
  return grid_info(gt, gi) ? gi.tripolar : -1;
}

void gcmf_plan_destroy(gcmf_plan *pl) {
  if (!pl) return;
  (void)hipSetDevice(pl->d.device);
  if (pl->stream) (void)hipStreamSynchronize(pl->stream);
  for (void *p : pl->owned) (void)hipFree(p);
  for (hipEvent_t e : pl->dom_ev) (void)hipEventDestroy(e);
  if (pl->work) (void)hipFree(pl->work);
  if (pl->side) { (void)hipStreamSynchronize(pl->side); (void)hipStreamDestroy(pl->side); }
  if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
  if (pl->ev_join) (void)hipEventDestroy(pl->ev_join);
  if (pl->ev0) (void)hipEventDestroy(pl->ev0);
  if (pl->ev1) (void)hipEventDestroy(pl->ev1);
  if (pl->ev_busy) (void)hipEventDestroy(pl->ev_busy);
  if (pl->s_in) { (void)hipStreamSynchronize(pl->s_in); (void)hipStreamDestroy(pl->s_in); }
  if (pl->s_out) { (void)hipStreamSynchronize(pl->s_out); (void)hipStreamDestroy(pl->s_out); }
  for (int q = 0; q < 2; ++q) {
    if (pl->ev_in[q]) (void)hipEventDestroy(pl->ev_in[q]);
    if (pl->ev_cmp[q]) (void)hipEventDestroy(pl->ev_cmp[q]);
    if (pl->ev_out[q]) (void)hipEventDestroy(pl->ev_out[q]);
  }
  if (pl->stage) (void)hipFree(pl->stage);
  if (pl->dev_p) (void)hipFree(pl->dev_p);
  resident_free(pl);
  if (pl->stream) (void)hipStreamDestroy(pl->stream);
  delete pl;
}

int gcmf_plan_create(const gcmf_plan_desc *desc, const void *const *planes, int nplanes, gcmf_plan **out) {
  if (!desc || !out) {
    set_error("gcmf_plan_create: null argument");
    return GCMF_ERR_INVALID_ARG;
  }
  *out = nullptr;
  GridInfo gi;
  if (!grid_info(desc->grid_type, gi)) {
    set_error("gcmf_plan_create: unknown grid_type %d", desc->grid_type);
    return GCMF_ERR_INVALID_ARG;
  }
  if (nplanes != gi.nplanes || (nplanes > 0 && !planes)) {
    set_error("gcmf_plan_create: grid type %d needs %d grid planes, got %d", desc->grid_type, gi.nplanes, nplanes);
    return GCMF_ERR_INVALID_ARG;
  }
  if (desc->dtype != GCMF_F32 && desc->dtype != GCMF_F64) {
    set_error("gcmf_plan_create: bad dtype %d", desc->dtype);
    return GCMF_ERR_INVALID_ARG;
  }
  if (desc->ny < 1 || desc->nx < 1 || desc->ny > (1 << 30) || desc->nx > (1 << 30) ||
      desc->ny * desc->nx > (int64_t)2000000000) {
    set_error("gcmf_plan_create: bad grid shape (%lld, %lld)", (long long)desc->ny, (long long)desc->nx);
    return GCMF_ERR_INVALID_ARG;
  }
  if (desc->row_begin < 0 || desc->row_end > desc->ny || desc->row_begin >= desc->row_end || desc->halo < 0) {
    set_error("gcmf_plan_create: bad row slab [%lld, %lld) of %lld rows", (long long)desc->row_begin,
              (long long)desc->row_end, (long long)desc->ny);
    return GCMF_ERR_INVALID_ARG;
  }
  for (int k = 0; k < nplanes; ++k)
    if (!planes[k]) {
      set_error("gcmf_plan_create: grid plane %d is NULL", k);
      return GCMF_ERR_INVALID_ARG;
    }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("no HIP device available (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return GCMF_ERR_NO_DEVICE;
  }
  GCMF_HIP(hipSetDevice(desc->device));

  gcmf_plan *pl = new gcmf_plan();
  pl->d = *desc;
  pl->kind = gi.kind;
  pl->ncomp = gi.ncomp;
  pl->tripolar = gi.tripolar;
  pl->area_weighted = gi.area_weighted;
  pl->dimensional = gi.dimensional;
  const bool ring_of_one = (desc->flags & GCMF_PLAN_SELF_RING) != 0;
  if (ring_of_one && (gi.tripolar || desc->row_begin != 0 || desc->row_end != desc->ny)) {
    delete pl;
    set_error("gcmf_plan_create: GCMF_PLAN_SELF_RING needs a non-tripolar plan covering the whole grid");
    return GCMF_ERR_INVALID_ARG;
  }
  pl->full = (desc->row_begin == 0 && desc->row_end == desc->ny) && !ring_of_one;
  int64_t gs = 0, gn = 0;
  if (!pl->full) {
    if (desc->halo < 1) {
      delete pl;
      set_error("gcmf_plan_create: a partial row slab needs halo >= 1");
      return GCMF_ERR_INVALID_ARG;
    }
    gs = (pl->tripolar && desc->row_begin == 0) ? 0 : desc->halo;
    gn = (pl->tripolar && desc->row_end == desc->ny) ? 0 : desc->halo;
  }
  pl->rows_owned = desc->row_end - desc->row_begin;
  pl->rows_alloc = gs + pl->rows_owned + gn;
  pl->first_owned = gs;
  Geom &g = pl->g;
  g.nx = (int)desc->nx;
  g.rows = (int)pl->rows_alloc;
  g.south_wrap = pl->full && !pl->tripolar;
  g.north_wrap = pl->full && !pl->tripolar;
  g.fold = pl->tripolar && desc->row_end == desc->ny;
  g.area_weighted = pl->area_weighted;

  auto fail = [&](int rc) {
    gcmf_plan_destroy(pl);
    return rc;
  };
#define PLAN_HIP(call)                                                                 \
  do {                                                                                 \
    hipError_t e2_ = (call);                                                           \
    if (e2_ != hipSuccess) {                                                           \
      set_error("%s failed: %s", #call, hipGetErrorString(e2_));                       \
      return fail(GCMF_ERR_HIP);                                                       \
    }                                                                                  \
  } while (0)
  if (const char *e = getenv("GCMF_CGRID_TILE")) pl->cgrid_tile = atoi(e);
  if (const char *e = getenv("GCMF_CGRID_RING")) pl->cgrid_ring = atoi(e);
  if (const char *e = getenv("GCMF_CGRID_RING_SMAX")) pl->cgrid_ring_smax = atoi(e);
  if (const char *e = getenv("GCMF_HOST_CHUNK_MB")) pl->host_chunk_bytes = (size_t)(atof(e) * 1048576.0);
  if (const char *e = getenv("GCMF_HOST_REGISTER")) pl->host_register = atoi(e);
  if (const char *e = getenv("GCMF_ZERO_LAND")) pl->zero_land = atoi(e);
  if (const char *e = getenv("GCMF_RING")) pl->ring = atoi(e);
  if (const char *e = getenv("GCMF_ZIGZAG")) pl->zigzag = atoi(e);
  if (const char *e = getenv("GCMF_RINGC_XE_ROWS")) pl->ringc_xe_rows = atoi(e);
  if (const char *e = getenv("GCMF_CLENSHAW")) pl->clenshaw = atoi(e);
  if (const char *e = getenv("GCMF_RINGC9")) pl->ringc9 = atoi(e);
  if (const char *e = getenv("GCMF_CLENSHAW_F32")) pl->clenshaw_f32 = atoi(e);
  PLAN_HIP(hipStreamCreateWithFlags(&pl->stream, hipStreamNonBlocking));
  PLAN_HIP(hipEventCreate(&pl->ev0));
  PLAN_HIP(hipEventCreate(&pl->ev1));
  PLAN_HIP(hipEventCreateWithFlags(&pl->ev_busy, hipEventDisableTiming));

  // stage the raw grid planes on the device (temporaries), fold them, free the temporaries
  const size_t plane_bytes = (size_t)desc->ny * desc->nx * dtype_size(desc->dtype);
  std::vector<const void *> dplanes(nplanes, nullptr);
  std::vector<void *> staged;
  int rc = GCMF_OK;
  for (int k = 0; k < nplanes && rc == GCMF_OK; ++k) {
    if (desc->planes_on_device) {
      dplanes[k] = planes[k];
      continue;
    }
    int dup = -1;  // the same host array passed twice (e.g. wet_mask_t is wet_mask_q) is uploaded once
    for (int q = 0; q < k; ++q)
      if (planes[q] == planes[k]) dup = q;
    if (dup >= 0) {
      dplanes[k] = dplanes[dup];
      continue;
    }
    void *p = nullptr;
    hipError_t e3 = hipMalloc(&p, plane_bytes);
    if (e3 == hipSuccess) {
      staged.push_back(p);
      e3 = hipMemcpyAsync(p, planes[k], plane_bytes, hipMemcpyHostToDevice, pl->stream);
    }
    if (e3 != hipSuccess) {
      set_error("staging grid plane %d failed: %s", k, hipGetErrorString(e3));
      rc = GCMF_ERR_HIP;
    }
    dplanes[k] = p;
  }
  if (rc == GCMF_OK) rc = precompute(pl, dplanes.data(), desc->planes_on_device ? nullptr : planes);
  if (rc == GCMF_OK && pl->ncomp == 1) {  // a row of zeros for k_ring
    void *z = nullptr;
    const size_t zb = ((size_t)desc->nx + 64) * 8 + 256;
    if (hipMalloc(&z, zb) == hipSuccess && hipMemsetAsync(z, 0, zb, pl->stream) == hipSuccess) {
      pl->owned.push_back(z);
      pl->zero_row = z;
      pl->ring_nfb = reinterpret_cast<unsigned *>((char *)z + zb - 8);  // beyond anything a (padded) row read touches
    } else if (z) {
      (void)hipFree(z);
    }
  }
  (void)hipStreamSynchronize(pl->stream);
  for (void *p : staged) (void)hipFree(p);
  if (rc != GCMF_OK) return fail(rc);
  *out = pl;
  return GCMF_OK;
#undef PLAN_HIP
}

int gcmf_plan_rows(const gcmf_plan *pl, int64_t *rows_alloc, int64_t *first_owned, int64_t *rows_owned) {
  if (!pl) return GCMF_ERR_INVALID_ARG;
  if (rows_alloc) *rows_alloc = pl->rows_alloc;
  if (first_owned) *first_owned = pl->first_owned;
  if (rows_owned) *rows_owned = pl->rows_owned;
  return GCMF_OK;
}

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
  else if (n == "ringc9") pl->ringc9 = value;
  else if (n == "clenshaw_f32") pl->clenshaw_f32 = value;
  else if (n == "ring_flux_f32") pl->ring_flux_f32 = value;
  else {
    set_error("gcmf_set_option: unknown option '%s'", name);
    return GCMF_ERR_INVALID_ARG;
  }
  return GCMF_OK;
}

int gcmf_cheb_step(gcmf_plan *pl, const void *const *t1, const void *const *t2, const void *const *fbar_in,
                   void *const *t0, void *const *fbar_out, double coef0, double coef1, double c, uint32_t mode,
                   uint32_t flags, int64_t nbatch, int64_t row_lo, int64_t row_hi, void *stream) {
  if (!pl || !t1 || !fbar_out) {
    set_error("gcmf_cheb_step: null argument");
    return GCMF_ERR_INVALID_ARG;
  }
  if (row_lo < 0 || row_hi > pl->rows_alloc || row_lo > row_hi) {
    set_error("gcmf_cheb_step: rows [%lld, %lld) outside the slab allocation of %lld rows", (long long)row_lo,
              (long long)row_hi, (long long)pl->rows_alloc);
    return GCMF_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  StepArgs a{};
  for (int k = 0; k < pl->ncomp; ++k) {
    a.t1[k] = t1[k];
    a.t2[k] = t2 ? t2[k] : nullptr;
    a.fb_in[k] = fbar_in ? fbar_in[k] : nullptr;
    a.t0[k] = t0 ? t0[k] : nullptr;
    a.fb_out[k] = fbar_out[k];
  }
  a.coef0 = coef0;
  a.coef1 = coef1;
  a.c = c;
  a.mode = mode & (GCMF_STEP_FIRST | GCMF_STEP_LAST);
  a.fb_is_f32 = (pl->d.dtype == GCMF_F32) && (flags & GCMF_OUT_F32);
  a.nbatch = nbatch;
  a.row_lo = (int)row_lo;
  a.row_hi = (int)row_hi;
  return step_dispatch(pl, a, (hipStream_t)stream);
}

int gcmf_multi_supported(const gcmf_plan *pl, int S) { return (pl && multi_supported(pl, S)) ? 1 : 0; }

// Backward (Clenshaw) evaluation (gcmf_ringc_impl.hpp): whether gcmf_apply uses it for this plan and polynomial, and how the
// n_steps levels are cut into launches of 5..8 (never leaving 1..4 or 9 behind).  plan->clenshaw = 1: the flux kinds, whose
// launches run at memcpy rate and gain the plane they no longer move (config 3: +10 %); 2: every scalar kind (the land-mask
// kernel is bound by its instruction stream and gains nothing: 93 -> 92-95 us per launch).  Needs the isolated cells fixed up
// by k_land_fix when there is land (land_ok).
static bool land_ok(const gcmf_plan *pl, int n_steps);
// Nine levels per launch (k_ringc<double, K_FLUX, 9>): whole f64 flux-form grids without a tripole seam (the seam's k_fold_band and the
// slabs' early-exit form stop at eight), tall enough for the deeper ghost zone.
static bool ringc9_ok(const gcmf_plan *pl) {
  return pl && pl->ringc9 && pl->kind == K_FLUX && pl->d.dtype == GCMF_F64 && pl->full && !pl->tripolar && !pl->g.fold && pl->g.rows >= 64;
}
static int clenshaw_cut(const gcmf_plan *pl, int n_steps, int *depths, int max_depths, bool f32_asked = false) {
  if (!pl || pl->ncomp != 1 || !(pl->clenshaw >= 2 || (pl->clenshaw == 1 && pl->kind == K_FLUX))) return 0;
  // f32 state (round 5): only when asked for (plan option clenshaw_f32 / GCMF_BACKWARD_F32 per call).  Summed backwards in f32 the
  // polynomial is 15-45 x further from f64 arithmetic than the reference's own f32 path (f32 T_k, f64 running sum; measured:
  // tools/measure_scalar_f32_error.py) -- the coefficients b_k grow like n - k where the T_k stay bounded -- and
  // Reinsch's form only halves that.  The forward kernels are that path itself (bit for bit on the REGULAR / land-mask kinds).
  if (pl->d.dtype != GCMF_F64 && !(pl->clenshaw_f32 || f32_asked)) return 0;
  // (tripolar: of the GRID, not of this slab -- every rank of a slab run must take the same decision)
  // (tripolar plans: the seam rows run k_fold_band's backward form beside every launch)
  // f32 state: the flux kinds only (four cells per lane; the whole polynomial is then carried in f32 -- Filter(evaluation="reference") /
  // GCMF_FORWARD_RECURRENCE keep the reference's f64 running sum)
  // (f32 state: the flux kinds since round 3, the REGULAR / land-mask kinds since round 4)
  if (!pl->ring || !pl->zero_row || pl->multi_s < 8 || !multi_supported(pl, 8)) return 0;
  if (pl->n_land > 0 && !land_ok(pl, n_steps)) return 0;
  if (!(n_steps >= 10 || (n_steps >= 5 && n_steps <= 8) || (n_steps == 9 && ringc9_ok(pl)))) return 0;
  if (ringc9_ok(pl) && (n_steps + 8) / 9 < (n_steps + 7) / 8) {
    // one launch fewer with up to nine levels each: as even as possible (63 = 7 x 9, 65 = 9 + 7 x 8), the nines first
    const int L = (n_steps + 8) / 9, q = n_steps / L, r = n_steps % L;
    if (L > max_depths) return 0;
    for (int k = 0; k < L; ++k) depths[k] = q + (k < r ? 1 : 0);
    return L;
  }
  int n = 0, left = n_steps;
  while (left > 0) {
    int S = 0;
    // (f32 state: the first launch -- it also carries the land bits of the rows that become b_n -- spills at eight levels)
    for (int cand = (n == 0 && pl->d.dtype != GCMF_F64) ? 7 : 8; cand >= 5 && !S; --cand) {
      const int rest = left - cand;
      if (rest == 0 || (rest >= 5 && rest != 9)) S = cand;
    }
    if (!S || n >= max_depths) return 0;
    depths[n++] = S;
    left -= S;
  }
  return n;
}

int gcmf_clenshaw_cut(const gcmf_plan *pl, int n_steps, int *depths, int max_depths) {
  if (!pl || !depths || max_depths < 1) return 0;
  return clenshaw_cut(pl, n_steps, depths, max_depths);
}

int gcmf_cheb_multi(gcmf_plan *pl, const void *u, const void *v, void *uo, void *vo, const void *fbar_in,
                    void *fbar_out, const double *pk, int S, double p0, double c, uint32_t mode, uint32_t flags,
                    int64_t nbatch, int64_t row_lo, int64_t row_hi, void *stream) {
  if (pl && pk && (mode & GCMF_STEP_CLENSHAW)) {
    // S levels of the backward evaluation on rows [row_lo, row_hi): (u, v) = (b_{k+1}, b_{k+2}) (FIRST: unused, the launch forms
    // b_n = p0 * f itself), fbar_in = the constant input f, pk[t] = coefficient of level t + 1, LAST: fbar_out = the result
    const bool first = mode & GCMF_STEP_FIRST, last = mode & GCMF_STEP_LAST;
    int probe[2];
    // (is the backward evaluation on offer for this plan at all: a 10-level polynomial can always be cut, [5, 5]; an f32 filter never
    // starts with eight levels, see clenshaw_cut)
    if (pl->ncomp != 1 || S < 5 || S > (ringc9_ok(pl) ? 9 : 8) || !pl->ring || !pl->zero_row || clenshaw_cut(pl, 10, probe, 2) != 2 ||
        (first && S == 8 && pl->d.dtype != GCMF_F64)) {
      set_error("gcmf_cheb_multi: the backward evaluation is not available for this plan / depth %d", S);
      return GCMF_ERR_UNSUPPORTED;
    }
    if (!fbar_in || (!first && (!u || !v)) || (!last && (!uo || !vo)) || (last && !fbar_out) || (uo && (uo == u || uo == v)) ||
        (vo && (vo == u || vo == v)) || row_lo < 0 || row_hi > pl->rows_alloc || row_lo > row_hi) {
      set_error("gcmf_cheb_multi: missing or aliased buffers / bad row range for the backward evaluation");
      return GCMF_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(pl->mu);
    GCMF_HIP(hipSetDevice(pl->d.device));
    MultiArgs m{};
    m.u0 = u; m.v0 = v; m.uo = uo; m.vo = vo; m.fb_in = fbar_in; m.fb_out = fbar_out;
    for (int t = 0; t < S; ++t) m.pk[t] = pk[t];
    m.p0 = p0; m.c = c; m.S = S; m.first = first; m.last = last; m.nbatch = nbatch; m.row_lo = (int)row_lo; m.row_hi = (int)row_hi;
    m.fb_is_f32 = (pl->d.dtype == GCMF_F32 && (flags & GCMF_OUT_F32)) ? 1 : 0;   // f32 state: the result is f64 unless asked otherwise
    return advance_multi(pl, m, (hipStream_t)stream, nullptr, true);
  }
  if (!pl || !u || !fbar_out || !pk) {
    set_error("gcmf_cheb_multi: null argument");
    return GCMF_ERR_INVALID_ARG;
  }
  if (!multi_supported(pl, S)) {
    set_error("gcmf_cheb_multi: S=%d is not available for this plan", S);
    return GCMF_ERR_UNSUPPORTED;
  }
  if (row_lo < 0 || row_hi > pl->rows_alloc || row_lo > row_hi) {
    set_error("gcmf_cheb_multi: rows [%lld, %lld) outside the slab allocation of %lld rows", (long long)row_lo,
              (long long)row_hi, (long long)pl->rows_alloc);
    return GCMF_ERR_INVALID_ARG;
  }
  const bool first = mode & GCMF_STEP_FIRST, last = mode & GCMF_STEP_LAST;
  if ((!first && (!v || !fbar_in)) || (!last && (!uo || !vo)) || uo == u || uo == v || vo == u || (vo && vo == v)) {
    set_error("gcmf_cheb_multi: missing or aliased state buffers");
    return GCMF_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  MultiArgs m{};
  m.u0 = u; m.v0 = v; m.uo = uo; m.vo = vo; m.fb_in = fbar_in; m.fb_out = fbar_out;
  for (int t = 0; t < S; ++t) m.pk[t] = pk[t];
  m.p0 = p0; m.c = c; m.S = S; m.first = first; m.last = last;
  m.fb_is_f32 = (pl->d.dtype == GCMF_F32) && (flags & GCMF_OUT_F32);
  m.nbatch = nbatch; m.row_lo = (int)row_lo; m.row_hi = (int)row_hi;
  m.land_zero = (mode & GCMF_STEP_LAND_ZERO) ? 1 : 0;
  m.ring_first = (first && !last && (mode & GCMF_STEP_LAND_FIXED)) ? 1 : 0;
  return advance_multi(pl, m, (hipStream_t)stream, nullptr);
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

int gcmf_resident_supported(const gcmf_plan *pl, int64_t row_lo, int64_t row_hi, int L) {
  if (!pl) return 0;
  (void)hipSetDevice(pl->d.device);
  return resident_fits(pl, (int)row_lo, (int)row_hi, L) ? 1 : 0;
}

// L levels of the backward evaluation in one launch on rows [row_lo, row_hi) (their dependency cone [row_lo - L, row_hi + L) must hold
// valid data): (u, v) = (b_{k+1}, b_{k+2}) (GCMF_STEP_FIRST: unused, b_n = p0 * f is formed on load), f = the constant input, pk[l] =
// the coefficient of level l + 1; GCMF_STEP_LAST: `out` receives the result, otherwise (uo, vo) the new states.  Same bits as the
// same levels run through gcmf_cheb_multi(GCMF_STEP_CLENSHAW) in launches of 5..8.
int gcmf_resident_levels(gcmf_plan *pl, const void *u, const void *v, void *uo, void *vo, const void *f, void *out, const double *pk, int L,
                         double p0, double c, uint32_t mode, int64_t row_lo, int64_t row_hi, void *stream) {
  if (!pl || !pk || !f || L < 1) {
    set_error("gcmf_resident_levels: null argument");
    return GCMF_ERR_INVALID_ARG;
  }
  const bool first = mode & GCMF_STEP_FIRST, last = mode & GCMF_STEP_LAST;
  if ((!first && (!u || !v)) || (!last && (!uo || !vo)) || (last && !out) || (uo && (uo == u || uo == v)) || (vo && (vo == u || vo == v))) {
    set_error("gcmf_resident_levels: missing or aliased buffers");
    return GCMF_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  MultiArgs m{};
  m.u0 = u; m.v0 = v; m.uo = uo; m.vo = vo; m.fb_in = f; m.fb_out = out;
  m.p0 = p0; m.c = c; m.S = L; m.first = first; m.last = last; m.nbatch = 1; m.row_lo = (int)row_lo; m.row_hi = (int)row_hi;
  return launch_resident(pl, m, pk, L, (hipStream_t)stream);
}

int gcmf_multi_supported_vec(const gcmf_plan *pl, int S, int64_t nbatch) {
  if (!pl || nbatch < 1) return 0;
  if (pl->ncomp == 1) return multi_supported(pl, S) ? 1 : 0;
  return vec_multi_supported(pl, nbatch, S) ? 1 : 0;
}

int gcmf_cheb_multi_vec(gcmf_plan *pl, const void *const *u, const void *const *v, void *const *uo, void *const *vo,
                        const void *const *fbar_in, void *const *fbar_out, const double *pk, int S, double p0, double c,
                        uint32_t mode, uint32_t flags, int64_t nbatch, int64_t row_lo, int64_t row_hi, void *stream) {
  if (!pl || !u || !fbar_out || !pk) {
    set_error("gcmf_cheb_multi_vec: null argument");
    return GCMF_ERR_INVALID_ARG;
  }
  if (pl->ncomp == 1)
    return gcmf_cheb_multi(pl, u[0], v ? v[0] : nullptr, uo ? uo[0] : nullptr, vo ? vo[0] : nullptr,
                           fbar_in ? fbar_in[0] : nullptr, fbar_out[0], pk, S, p0, c, mode, flags, nbatch, row_lo,
                           row_hi, stream);
  if (!vec_multi_supported(pl, nbatch, S, (mode & GCMF_STEP_CLENSHAW) != 0)) {
    set_error("gcmf_cheb_multi_vec: S=%d with %lld levels is not available for this plan", S, (long long)nbatch);
    return GCMF_ERR_UNSUPPORTED;
  }
  if (row_lo < 0 || row_hi > pl->rows_alloc || row_lo > row_hi) {
    set_error("gcmf_cheb_multi_vec: rows [%lld, %lld) outside the slab allocation of %lld rows", (long long)row_lo,
              (long long)row_hi, (long long)pl->rows_alloc);
    return GCMF_ERR_INVALID_ARG;
  }
  const bool first = mode & GCMF_STEP_FIRST, last = mode & GCMF_STEP_LAST;
  VecMultiArgs m{};
  for (int q = 0; q < 2; ++q) {
    const void *uq = u[q], *vq = v ? v[q] : nullptr, *fi = fbar_in ? fbar_in[q] : nullptr;
    void *uoq = uo ? uo[q] : nullptr, *voq = vo ? vo[q] : nullptr;
    if (!uq || !fbar_out[q] || (!first && (!vq || !fi)) || (!last && (!uoq || !voq)) || uoq == uq || (uoq && uoq == vq) ||
        voq == uq || (voq && voq == vq) || (uoq && uoq == voq)) {
      set_error("gcmf_cheb_multi_vec: missing or aliased state buffers");
      return GCMF_ERR_INVALID_ARG;
    }
    m.u0[q] = uq; m.uprev[q] = vq; m.u2o[q] = uoq; m.u1o[q] = voq; m.fb_in[q] = fi; m.fb_out[q] = fbar_out[q];
  }
  for (int t = 0; t < S; ++t) m.pk[t] = pk[t];
  m.p0 = p0; m.c = c; m.S = S; m.first = first; m.last = last;
  m.fb_is_f32 = (pl->d.dtype == GCMF_F32) && (flags & GCMF_OUT_F32);
  m.nbatch = nbatch; m.row_lo = (int)row_lo; m.row_hi = (int)row_hi;
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  return launch_vec_multi(pl, m, (hipStream_t)stream);
}

static bool land_ok(const gcmf_plan *pl, int n_steps) {
  return pl && (pl->kind == K_FLUX || pl->kind == K_MASK) && pl->zero_land && pl->lbits && pl->n_land > 0 && (pl->d.nx % 4) == 0 && n_steps < 4096;
}

int gcmf_has_land(const gcmf_plan *pl) { return land_ok(pl, 0) ? 1 : 0; }

int gcmf_zero_land(gcmf_plan *pl, void *const *a, void *const *b, int64_t nbatch, void *stream) {
  if (!pl || !a || !b || !a[0] || !b[0] || nbatch < 1) return GCMF_ERR_INVALID_ARG;
  if (!land_ok(pl, 0)) return GCMF_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  return launch_zero_land(pl, a[0], b[0], nbatch, (hipStream_t)stream);
}

int gcmf_land_fix(gcmf_plan *pl, const double *p, int n_steps, double c, const void *const *in, void *const *out,
                  int64_t nbatch, uint32_t flags, void *stream) {
  if (!pl || !p || !in || !out || !in[0] || !out[0] || n_steps < 1 || nbatch < 1) return GCMF_ERR_INVALID_ARG;
  if (!land_ok(pl, n_steps)) return GCMF_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  int rc = ensure_dev_p(pl, p, n_steps, (hipStream_t)stream);
  if (rc) return rc;
  const int fb32 = (pl->d.dtype == GCMF_F32 && (flags & GCMF_OUT_F32)) ? 1 : 0;
  return launch_land_fix(pl, in[0], out[0], pl->dev_p, n_steps, c, fb32, nbatch, (hipStream_t)stream);
}

// One whole filter application on this rank's slab, backward (Clenshaw) evaluation, scalar kinds: the choreography of
// gcm_filters_amd/distributed.py (SlabFilter._apply_backward) in C++ -- launches, ghost-zone bookkeeping, the overlapped edge / interior
// split and the halo exchanges through a gcmf_comm (RCCL) or a gcmf_p2p (mailboxes) -- enqueued on `stream` in ONE call.  The Python
// driver costs ~15 us of host time per launch and 13-33 us per exchange: 0.25-0.35 ms per application, more than the 0.25 ms an
// 8-way slab of a 2400x3600 grid computes, so a multi-GPU run was bound by its host.
//   X     the input with its own rows filled in (ghost rows are exchanged here), (nbatch, rows_alloc, nx)
//   pool  four state planes, out  the result (f64, or the state dtype with GCMF_OUT_F32); all (nbatch, rows_alloc, nx)
//   cut   the launch depths (gcmf_clenshaw_cut), halo  ghost rows per side (>= the deepest launch), south / north  peer ranks or -1
//   comm / p2p: at most one non-NULL (both NULL: a single slab without neighbours)
int gcmf_slab_apply_backward(gcmf_plan *pl, gcmf_comm *comm, gcmf_p2p *p2p, int south, int north, const double *p, int n_steps, double c,
                             const int *cut, int ncut, void *X, void *const *pool, void *out, int64_t nbatch, int halo, int overlap,
                             uint32_t flags, void *stream) {
  if (!pl || !p || !cut || ncut < 1 || !X || !pool || !out || nbatch < 1 || pl->ncomp != 1) {
    set_error("gcmf_slab_apply_backward: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  int total = 0, deepest = 0;
  for (int q = 0; q < ncut; ++q) { total += cut[q]; deepest = std::max(deepest, cut[q]); }
  const bool multi = (south >= 0 || north >= 0);
  if (total != n_steps || (multi && (halo < deepest || !(comm || p2p))) || (comm && p2p)) {
    set_error("gcmf_slab_apply_backward: the cut does not add up to n_steps, the halo is shallower than a launch, or no exchange was given");
    return GCMF_ERR_INVALID_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t fo = pl->first_owned, ro = pl->rows_owned, ra = pl->rows_alloc;
  const bool gs = fo > 0, gn = ra - fo - ro > 0;
  const int hs = multi ? halo : 0;
  const int dtype = pl->d.dtype;
  const int nx = (int)pl->d.nx;
  auto exchange_start = [&](void *const *st, int nst) -> int {
    if (!multi) return GCMF_OK;
    if (p2p) return gcmf_p2p_start(p2p, st, nst, nbatch, ra, nx, fo, ro, hs, dtype, stream);
    return gcmf_halo_start(comm, st, nst, nbatch, ra, nx, fo, ro, hs, dtype, south, north, stream);
  };
  auto exchange_finish = [&]() -> int {
    if (!multi) return GCMF_OK;
    return p2p ? gcmf_p2p_finish(p2p, stream) : gcmf_halo_finish(comm, stream);
  };
  int rc;
  {  // f's ghost rows: the first launch forms b_n = p_n f on them, the later ones read f on the rows they compute
    void *st[1] = {X};
    if ((rc = exchange_start(st, 1)) || (rc = exchange_finish())) return rc;
  }
  const bool fb32 = (dtype == GCMF_F32) && (flags & GCMF_OUT_F32);
  void *u = nullptr, *v = nullptr;
  int valid = hs, lvl = 1;
  // ---- the slab fits on the chip: every stretch between two exchanges is ONE resident launch (gcmf_resident.hip) -- as many levels as
  // there are ghost rows (no exchange: all of them, 64 at a time).  Same bits as the launches of 5..8 below.
  {
    const int per = multi ? std::min(hs, 64) : 64;
    bool fits = nbatch == 1 && !(flags & GCMF_NO_RESIDENT) && per >= 1;
    for (int done = 0; fits && done < n_steps;) {
      const int L = std::min(per, n_steps - done);
      const int vo_ = multi ? hs - L : 0;
      std::lock_guard<std::mutex> lk(pl->mu);
      GCMF_HIP(hipSetDevice(pl->d.device));
      fits = resident_supported(pl, (int)(fo - (gs ? vo_ : 0)), (int)(fo + ro + (gn ? vo_ : 0)), L, n_steps);
      done += L;
    }
    if (fits) {
      std::vector<double> pk(64);
      for (int done = 0; done < n_steps;) {
        const int L = std::min(per, n_steps - done);
        if (multi && done > 0) {   // the ghost zone is used up: refresh it
          void *st[2] = {u, v};
          if ((rc = exchange_start(st, 2)) || (rc = exchange_finish())) return rc;
        }
        void *fr[2] = {nullptr, nullptr};
        int nf = 0;
        for (int k = 0; k < 4 && nf < 2; ++k)
          if (pool[k] != u && pool[k] != v) fr[nf++] = pool[k];
        const int vo_ = multi ? hs - L : 0;
        MultiArgs m{};
        m.u0 = u; m.v0 = v; m.uo = fr[0]; m.vo = fr[1]; m.fb_in = X; m.fb_out = out;
        for (int t = 0; t < L; ++t) pk[t] = p[n_steps - (done + 1 + t)];
        m.p0 = p[n_steps]; m.c = c; m.S = L; m.first = (done == 0); m.last = (done + L == n_steps); m.nbatch = 1;
        m.row_lo = (int)(fo - (gs ? vo_ : 0)); m.row_hi = (int)(fo + ro + (gn ? vo_ : 0));
        {
          std::lock_guard<std::mutex> lk(pl->mu);
          GCMF_HIP(hipSetDevice(pl->d.device));
          if ((rc = launch_resident(pl, m, pk.data(), L, s))) return rc;
        }
        u = fr[0]; v = fr[1];
        done += L;
      }
      goto land_and_guard;
    }
  }
  for (int q = 0; q < ncut; ++q) {
    const int S = cut[q];
    if (multi && valid < S) {
      void *st[2] = {u, v};
      if ((rc = exchange_start(st, 2)) || (rc = exchange_finish())) return rc;
      valid = hs;
    }
    void *fr[2] = {nullptr, nullptr};
    int nf = 0;
    for (int k = 0; k < 4 && nf < 2; ++k)
      if (pool[k] != u && pool[k] != v) fr[nf++] = pool[k];
    int v_out = multi ? valid - S : 0;
    const int lo = (int)(fo - (gs ? v_out : 0)), hi = (int)(fo + ro + (gn ? v_out : 0));
    const bool last = (q == ncut - 1);
    MultiArgs m{};
    m.u0 = u; m.v0 = v; m.uo = fr[0]; m.vo = fr[1]; m.fb_in = X; m.fb_out = out;
    for (int t = 0; t < S; ++t) m.pk[t] = p[n_steps - (lvl + t)];
    m.p0 = p[n_steps]; m.c = c; m.S = S; m.first = (q == 0); m.last = last; m.nbatch = nbatch; m.fb_is_f32 = fb32 ? 1 : 0;
    const int nxt = last ? 0 : cut[q + 1];
    const bool ovl = overlap && multi && !last && v_out < nxt && ro >= 4 * (int64_t)hs;
    auto launch = [&](int r0, int r1) -> int {
      if (r1 <= r0) return GCMF_OK;
      std::lock_guard<std::mutex> lk(pl->mu);
      GCMF_HIP(hipSetDevice(pl->d.device));
      MultiArgs mm = m;
      mm.row_lo = r0; mm.row_hi = r1;
      return advance_multi(pl, mm, s, nullptr, true);
    };
    if (ovl) {
      // the next launch needs fresh ghost rows: advance the rows the neighbours need first, post the exchange of the NEW state, and
      // let the interior rows run while the messages are in flight (the edge launches reach to the inner end of what is sent)
      const int ilo = gs ? (int)(fo + hs) : lo, ihi = gn ? (int)(fo + ro - hs) : hi;
      if (gs && (rc = launch(lo, ilo))) return rc;
      if (gn && (rc = launch(ihi, hi))) return rc;
      void *st[2] = {fr[0], fr[1]};
      if ((rc = exchange_start(st, 2))) return rc;
      if ((rc = launch(ilo, ihi))) return rc;
      if ((rc = exchange_finish())) return rc;
      v_out = hs;
    } else if ((rc = launch(lo, hi))) {
      return rc;
    }
    u = fr[0]; v = fr[1];
    valid = v_out;
    lvl += S;
  }
land_and_guard:
  if (land_ok(pl, n_steps)) {
    std::lock_guard<std::mutex> lk(pl->mu);
    GCMF_HIP(hipSetDevice(pl->d.device));
    if ((rc = ensure_dev_p(pl, p, n_steps, s))) return rc;
    if ((rc = launch_land_fix(pl, X, out, pl->dev_p, n_steps, c, fb32 ? 1 : 0, nbatch, s))) return rc;
  }
  if (p2p && multi) {   // a failed exchange must not leave a plausible result (the stencils take NaN ghost rows as zero)
    const size_t obytes = (size_t)nbatch * ra * nx * ((dtype == GCMF_F32 && fb32) ? 4 : 8);
    if ((rc = gcmf_p2p_guard(p2p, out, (int64_t)(obytes / 16 * 16), stream))) return rc;
  }
  return GCMF_OK;
}

// The vector kinds' counterpart (a C-grid / B-grid field sharded over y used to run the Python choreography of
// distributed.py, forward, ~15 us of host time per launch): one whole backward (Clenshaw) application of a VECTOR plan on this rank's slab
// in one call.  X / out: the two components; pool: four state plane PAIRS, pool[2 q + comp]; all (nbatch, rows_alloc, nx).  The levels are
// cut exactly as gcmf_apply cuts them for this plan (at most four per launch), so a level filtered on a slab and in one piece see the same
// arithmetic; a launch of S levels uses up S ghost rows, the ghost zone is refreshed (both states, both components: four planes in one
// message per neighbour) when fewer are left than the next launch needs.  No edge / interior split.
// Levels of the next launch of a backward VECTOR application with `left` levels to go and at most smax per launch: the fewest launches,
// their depths evened out (44 levels at up to six per launch: 6 6 6 6 5 5 5 5 -- greedy sixes would leave a two-level launch of the
// general kernel at the end), never a lone single level left behind; gcmf_apply and the slab driver cut alike.
static bool ptr_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int vec_backward_next_depth(const gcmf_plan *pl, int64_t nbatch, int left, int smax) {
  if (smax >= 5) {
    const int nl = (left + smax - 1) / smax, S = (left + nl - 1) / nl;
    if (S >= 2 && S <= left && left - S != 1 && vec_multi_supported(pl, nbatch, S, true)) return S;
  }
  for (int cand = smax; cand >= 2; --cand)
    if (cand <= left && left - cand != 1 && vec_multi_supported(pl, nbatch, cand, true)) return cand;
  return left;
}

int gcmf_slab_backward_vec_supported(const gcmf_plan *pl, int64_t nbatch, int halo) {
  if (!pl || pl->ncomp != 2 || nbatch < 1) return 0;
  if (!((pl->kind == K_CGRID && pl->clenshaw >= 1) || (pl->kind == K_BGRID && pl->clenshaw >= 2 && (pl->d.dtype == GCMF_F64 || pl->clenshaw_f32))))
    return 0;
  return (pl->multi_s >= 2 && vec_multi_supported(pl, nbatch, 2) && (halo == 0 || halo >= 4)) ? 1 : 0;
}

int gcmf_slab_apply_backward_vec(gcmf_plan *pl, gcmf_comm *comm, gcmf_p2p *p2p, int south, int north, const double *p, int n_steps, double c,
                                 void *const *X, void *const *pool, void *const *out, int64_t nbatch, int halo, uint32_t flags, void *stream) {
  if (!pl || !p || !X || !pool || !out || !X[0] || !X[1] || !out[0] || !out[1] || nbatch < 1 || n_steps < 2) {
    set_error("gcmf_slab_apply_backward_vec: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  const bool multi = (south >= 0 || north >= 0);
  if (!gcmf_slab_backward_vec_supported(pl, nbatch, multi ? halo : 0) || (multi && !(comm || p2p)) || (comm && p2p)) {
    set_error("gcmf_slab_apply_backward_vec: no backward vector kernel for this plan / batch, a ghost zone shallower than a launch (4), or no exchange given");
    return GCMF_ERR_UNSUPPORTED;
  }
  for (int q = 0; q < 8; ++q)
    if (!pool[q]) return GCMF_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t fo = pl->first_owned, ro = pl->rows_owned, ra = pl->rows_alloc;
  const bool gs = fo > 0, gn = ra - fo - ro > 0;
  const int hs = multi ? halo : 0;
  const int dtype = pl->d.dtype;
  const int nx = (int)pl->d.nx;
  const bool fb32 = (dtype == GCMF_F32) && (flags & GCMF_OUT_F32);
  auto exchange = [&](void *const *st, int nst) -> int {
    if (!multi) return GCMF_OK;
    int rc = p2p ? gcmf_p2p_start(p2p, st, nst, nbatch, ra, nx, fo, ro, hs, dtype, stream)
                 : gcmf_halo_start(comm, st, nst, nbatch, ra, nx, fo, ro, hs, dtype, south, north, stream);
    if (rc) return rc;
    return p2p ? gcmf_p2p_finish(p2p, stream) : gcmf_halo_finish(comm, stream);
  };
  int rc;
  {  // f's ghost rows (both components)
    void *st[2] = {X[0], X[1]};
    if ((rc = exchange(st, 2))) return rc;
  }
  const void *u[2] = {X[0], X[1]}, *v[2] = {nullptr, nullptr};
  // (launches deeper than five levels exist only in k_cgrid_ring, whose 16-byte accesses need the caller's planes aligned)
  bool al16 = true;
  for (int q = 0; q < 2; ++q) al16 = al16 && ptr_al16(X[q]) && ptr_al16(out[q]);
  for (int q = 0; q < 8; ++q) al16 = al16 && ptr_al16(pool[q]);
  const int ring_smax = al16 ? cgrid_ring_smax(pl, nbatch) : std::min(5, cgrid_ring_smax(pl, nbatch));
  const int smax = std::min(std::min(pl->multi_s, std::max(4, ring_smax)), multi ? std::max(4, halo) : 8);   // (as gcmf_apply cuts them; never deeper than the ghost zone)
  int valid = hs, lvl = 1;
  while (lvl <= n_steps) {
    const int left = n_steps - lvl + 1;
    const int S = vec_backward_next_depth(pl, nbatch, left, smax);
    if (multi && valid < S) {   // (never before the first launch: valid = halo >= 4 there)
      void *st[4] = {const_cast<void *>(u[0]), const_cast<void *>(u[1]), const_cast<void *>(v[0]), const_cast<void *>(v[1])};
      if ((rc = exchange(st, 4))) return rc;
      valid = hs;
    }
    void *fr[2][2];
    int nf = 0;
    for (int q = 0; q < 4 && nf < 2; ++q)
      if (pool[2 * q] != u[0] && pool[2 * q] != v[0]) { fr[nf][0] = pool[2 * q]; fr[nf][1] = pool[2 * q + 1]; ++nf; }
    const int v_out = multi ? valid - S : 0;
    const bool is_last = (lvl + S - 1 == n_steps);
    VecMultiArgs m{};
    for (int q = 0; q < 2; ++q) {
      m.u0[q] = u[q]; m.uprev[q] = v[q]; m.u1o[q] = fr[0][q]; m.u2o[q] = fr[1][q];
      m.fb_in[q] = X[q]; m.fb_out[q] = out[q];
    }
    for (int t = 0; t < S; ++t) m.pk[t] = p[n_steps - (lvl + t)];
    m.p0 = p[n_steps]; m.c = c; m.S = S; m.clen = 1;
    m.first = (lvl == 1); m.last = is_last; m.fb_is_f32 = fb32; m.nbatch = nbatch;
    m.row_lo = (int)(fo - (gs ? v_out : 0)); m.row_hi = (int)(fo + ro + (gn ? v_out : 0));
    {
      std::lock_guard<std::mutex> lk(pl->mu);
      GCMF_HIP(hipSetDevice(pl->d.device));
      if ((rc = launch_vec_multi(pl, m, s))) return rc;
    }
    for (int q = 0; q < 2; ++q) { u[q] = fr[1][q]; v[q] = fr[0][q]; }
    valid = v_out;
    lvl += S;
  }
  if (p2p && multi) {
    const size_t obytes = (size_t)nbatch * ra * nx * ((dtype == GCMF_F32 && fb32) ? 4 : 8);
    for (int q = 0; q < 2; ++q)
      if ((rc = gcmf_p2p_guard(p2p, out[q], (int64_t)(obytes / 16 * 16), stream))) return rc;
  }
  return GCMF_OK;
}

int gcmf_prepare(gcmf_plan *pl, const void *const *in, void *const *out, int64_t nbatch, int64_t row_lo,
                 int64_t row_hi, void *stream) {
  if (!pl || !in || !out) return GCMF_ERR_INVALID_ARG;
  if (row_lo < 0 || row_hi > pl->rows_alloc || row_lo > row_hi) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  return launch_prepare(pl, in, out, nbatch, (int)row_lo, (int)row_hi, (hipStream_t)stream);
}

static const int64_t MAX_LAUNCH_BATCH = 32768;  // batch entries per launch (gridDim.y of the scalar kernels)

// Shared driver of gcmf_apply and gcmf_laplacian; the plan's mutex is held and the device is current.
// `timed` = false: the caller (the pipelined host path) brackets the launches with the timing events itself.
static int run_whole_locked(gcmf_plan *pl, const double *p, int n_steps, double c, const void *const *in,
                            void *const *out, int64_t nbatch, uint32_t flags, void *stream, bool lapl_only,
                            bool timed) {
  const bool on_dev = flags & GCMF_DEVICE_PTRS;
  // device pointers: run on exactly the caller's stream (NULL = the HIP default stream) so the work is
  // ordered with the caller's own kernels; host pointers: the plan's private stream unless one is given
  hipStream_t s = (on_dev || stream) ? (hipStream_t)stream : pl->stream;
  const bool f32 = pl->d.dtype == GCMF_F32;
  const bool fb32 = f32 && (flags & GCMF_OUT_F32);
  const size_t ts = dtype_size(pl->d.dtype);
  const size_t fbs = lapl_only ? ts : ((f32 && !fb32) ? 8 : ts);  // element size of fbar == element size of `out`
  const size_t ncell = (size_t)nbatch * pl->d.ny * pl->d.nx;
  const int nc = pl->ncomp;
  const bool prep = pl->area_weighted && !lapl_only;

  // work layout per component: [A][B][fbar] (+ [prepared T0]) (+ host staging: [in][out])
  const size_t szT = align_up(ncell * ts, 256), szF = align_up(ncell * fbs, 256);
  const bool use_multi = !lapl_only && pl->multi_s >= 2 && n_steps >= 2 && multi_supported(pl, 2);
  const bool use_vmulti = !lapl_only && pl->multi_s >= 2 && n_steps >= 2 && vec_multi_supported(pl, nbatch, 2);
  size_t per = 0;
  const size_t oA = per; per += szT;
  const size_t oB = per; per += szT;
  const size_t oC = per; if (use_multi || use_vmulti) per += szT;
  const size_t oD = per; if (use_multi || use_vmulti) per += szT;
  const size_t oF = per; per += szF;
  // second fbar plane (scalar blocked schedule): k_ring re-does a strip from its inputs when it meets a NaN / inf,
  // so a launch must not accumulate fbar in place
  const size_t oF2 = per; if (use_multi) per += szF;
  // flux kinds only: the land-mask kernels have a NaN-only mode that already makes NaN on land free, there the two
  // extra passes would only cost (measured -6 %)
  const bool zero_land = use_multi && land_ok(pl, n_steps);  // n_steps < 4096: k_land_fix keeps p in LDS
  const size_t oP = per; if (prep) per += szT;
  const size_t oIn = per; if (!on_dev) per += szT;
  const size_t oOut = per; if (!on_dev) per += szF;
  int rc = ensure_work(pl, per * nc);
  if (rc) return rc;
  // the plan's work buffers are shared by all calls: a call enqueued on another stream (dask worker threads with
  // their own streams) must not start before the previous one has finished with them
  // (the wait is needed only when this call is on ANOTHER stream than the last one: a stream orders its own work.  The event is
  // recorded lazily, here, on the previous call's stream -- behind everything that stream was given since, which is later
  // than necessary but correct -- so back-to-back calls on one stream pay for no event at all: two queue packets less per
  // application, ~8 us of a 512x512 filter)
  if (pl->busy_valid && pl->busy_stream != s) {
    if (pl->busy_recorded || hipEventRecord(pl->ev_busy, pl->busy_stream) == hipSuccess) {
      GCMF_HIP(hipStreamWaitEvent(s, pl->ev_busy, 0));
    } else {   // the caller destroyed that stream meanwhile: whatever ran on it is waited for the blunt way
      (void)hipGetLastError();
      GCMF_HIP(hipDeviceSynchronize());
    }
  }
  char *w = (char *)pl->work;
  const void *din[2];
  void *dout[2], *A[2], *B[2], *Cb[2], *Db[2], *F[2], *Pp[2];
  for (int k = 0; k < nc; ++k) {
    char *base = w + per * k;
    A[k] = base + oA;
    B[k] = base + oB;
    Cb[k] = base + oC;
    Db[k] = base + oD;
    F[k] = base + oF;
    Pp[k] = base + oP;
    if (on_dev) {
      din[k] = in[k];
      dout[k] = out[k];
    } else {
      din[k] = base + oIn;
      dout[k] = base + oOut;
      GCMF_HIP(hipMemcpyAsync(base + oIn, in[k], ncell * ts, hipMemcpyHostToDevice, s));
    }
  }
  const int rows = (int)pl->rows_alloc;
  int launches = 0;
  const bool timing = pl->timing && timed;
  if (timing) GCMF_HIP(hipEventRecord(pl->ev0, s));
  if (lapl_only) {
    StepArgs a{};
    for (int k = 0; k < nc; ++k) { a.t1[k] = din[k]; a.t0[k] = dout[k]; a.fb_out[k] = nullptr; }
    a.mode = STEP_LAPL;
    a.nbatch = nbatch;
    a.row_lo = 0;
    a.row_hi = rows;
    if ((rc = step_dispatch(pl, a, s))) return rc;
    ++launches;
  } else {
    const void *x0[2] = {din[0], din[1]};
    int depths[1024];
    const bool fwd_only = flags & GCMF_FORWARD_RECURRENCE;   // the caller wants the reference's forward recurrence / accumulation
    const bool back_f32 = pl->clenshaw_f32 || (flags & GCMF_BACKWARD_F32);   // f32 B-grid / scalar state backwards: only when asked for
    const int n_clen = (use_multi && !fwd_only) ? clenshaw_cut(pl, n_steps, depths, 1024, back_f32) : 0;
    // Small fields: the whole polynomial on the chip in ONE launch (64 levels at a time; gcmf_resident.hip) -- the field, both states
    // and the coefficients live in registers / LDS, nothing but the result goes back to memory.  Same bits as the launches below.
    bool resident = false;
    int path = GCMF_PATH_STRIPS;
    if (pl->res_lo) {   // the LAST call of this plan ran on the chip: did one of ITS launches time out?  (told once, to the plan whose
      //                   output was poisoned -- never to an unrelated plan; the process runs the strip-marching launches from now on)
      const unsigned lo = pl->res_lo, hi = pl->res_hi;
      pl->res_lo = pl->res_hi = 0;
      if (resident_take_failure(pl->d.device, lo, hi)) {
        set_error("k_resident: the previous on-chip application of this plan timed out waiting for a neighbour tile and its result is NaN "
                  "(another process running resident kernels on this GPU outside the lock file's reach?); the strip-marching launches are "
                  "used from now on");
        return GCMF_ERR_HIP;
      }
    }
    if (n_clen > 0 && nbatch == 1 && !(flags & GCMF_NO_RESIDENT)) {
      int why = GCMF_RESIDENT_OFF;
      resident = resident_supported(pl, 0, rows, std::min(n_steps, 64), n_steps, &why);   // (small whole grids; GCMF_RESIDENT=1: whatever fits)
      if (!resident && why == GCMF_RESIDENT_LOCK_BUSY) path = GCMF_PATH_STRIPS_LOCK_BUSY;
      if (!resident && why == GCMF_RESIDENT_DISABLED) path = GCMF_PATH_STRIPS_DISABLED;
    }
    if (resident) path = GCMF_PATH_RESIDENT;
    pl->last_path = path;
    ++pl->path_count[path];
    if (resident) {
      pl->res_lo = resident_next_serial(pl->d.device);
      pl->res_hi = pl->res_lo;
      void *pool[4] = {A[0], B[0], Cb[0], Db[0]};
      const void *u = nullptr, *v = nullptr;
      double pkk[64];
      for (int done = 0; done < n_steps;) {
        const int L = std::min(64, n_steps - done);
        void *fr[2] = {nullptr, nullptr};
        int nf = 0;
        for (int q = 0; q < 4 && nf < 2; ++q)
          if (pool[q] != u && pool[q] != v) fr[nf++] = pool[q];
        MultiArgs m{};
        m.u0 = u; m.v0 = v; m.uo = fr[0]; m.vo = fr[1];
        m.fb_in = din[0]; m.fb_out = dout[0];
        m.first = (done == 0); m.last = (done + L == n_steps); m.S = L; m.fb_is_f32 = fb32;
        for (int t = 0; t < L; ++t) pkk[t] = p[n_steps - (done + 1 + t)];
        m.p0 = p[n_steps]; m.c = c; m.nbatch = 1; m.row_lo = 0; m.row_hi = rows;
        if ((rc = dom_begin(pl, s))) return rc;
        pl->res_hi = resident_next_serial(pl->d.device);   // (other plans of the process may have launched in between: a range, not a count)
        if ((rc = launch_resident(pl, m, pkk, L, s))) return rc;
        if ((rc = dom_end(pl, s))) return rc;
        ++launches;
        u = fr[0]; v = fr[1];
        done += L;
      }
      if (pl->n_land > 0) {
        if ((rc = ensure_dev_p(pl, p, n_steps, s))) return rc;
        if ((rc = launch_land_fix(pl, din[0], dout[0], pl->dev_p, n_steps, c, fb32 ? 1 : 0, nbatch, s))) return rc;
      }
    } else if (n_clen > 0) {
      // Backward (Clenshaw) evaluation, gcmf_ringc_impl.hpp: state (b_{k+1}, b_{k+2}) in a pool of four planes, the constant
      // input read by every launch, no fbar planes.  The first launch forms b_n = p[n] f as it loads f; level l = 1..n uses
      // p[n - l]; the last launch writes the result.
      void *pool[4] = {A[0], B[0], Cb[0], Db[0]};
      const void *u = nullptr, *v = nullptr;
      int lvl = 1;
      for (int q = 0; q < n_clen; ++q) {
        const int S = depths[q];
        void *fr[2] = {nullptr, nullptr};
        int nf = 0;
        for (int q = 0; q < 4 && nf < 2; ++q)
          if (pool[q] != u && pool[q] != v) fr[nf++] = pool[q];
        MultiArgs m{};
        m.u0 = u; m.v0 = v; m.uo = fr[0]; m.vo = fr[1];
        m.fb_in = din[0]; m.fb_out = dout[0];
        m.first = (q == 0); m.last = (q == n_clen - 1); m.S = S; m.fb_is_f32 = fb32;
        for (int t = 0; t < S; ++t) m.pk[t] = p[n_steps - (lvl + t)];
        m.p0 = p[n_steps]; m.c = c; m.nbatch = nbatch; m.row_lo = 0; m.row_hi = rows;
        if ((rc = advance_multi(pl, m, s, &launches, true))) return rc;   // (+ the tripole band on tripolar plans)
        u = fr[0]; v = fr[1];
        lvl += S;
      }
      if (pl->n_land > 0) {  // the isolated cells' own polynomial (forward recurrence, as the reference computes it)
        if ((rc = ensure_dev_p(pl, p, n_steps, s))) return rc;
        if ((rc = launch_land_fix(pl, din[0], dout[0], pl->dev_p, n_steps, c, fb32 ? 1 : 0, nbatch, s))) return rc;
      }
    } else if (use_multi) {
      // Temporally blocked schedule (scalar kinds): each launch advances S steps and reads/writes every plane
      // once.  prepare/finalize are fused into the first / last launch.  State buffers rotate through a pool
      // of four because a launch may not overwrite the planes its neighbours' halos are still reading.
      void *pool[4] = {A[0], B[0], Cb[0], Db[0]};
      void *Fcur = F[0], *Fnext = w + oF2;   // fbar ping-pongs between two planes (see oF2)
      const void *u = x0[0], *v = nullptr;
      int k = 1;
      bool land_zeroed = false;  // the first blocked launch kept the isolated cells out of the state
      while (k <= n_steps) {
        const int left = n_steps - k + 1;
        int S = 1;
        const int cand[7] = {8, 7, 6, 5, 4, 3, 2};
        for (int q = 0; q < 7; ++q)  // largest depth that does not strand a lone single step at the end
          if (cand[q] <= left && left - cand[q] != 1 && cand[q] <= pl->multi_s && multi_supported(pl, cand[q])) {
            S = cand[q];
            break;
          }
        void *fr[2] = {nullptr, nullptr};
        int nf = 0;
        for (int q = 0; q < 4 && nf < 2; ++q)
          if (pool[q] != u && pool[q] != v) fr[nf++] = pool[q];
        const bool is_last = (k + S - 1 == n_steps);
        if (S >= 2) {
          MultiArgs m{};
          m.u0 = u; m.v0 = v; m.uo = fr[0]; m.vo = fr[1];
          m.fb_in = Fcur; m.fb_out = is_last ? dout[0] : Fnext;
          std::swap(Fcur, Fnext);
          m.first = (k == 1); m.last = is_last; m.S = S; m.fb_is_f32 = fb32;
          m.land_zero = land_zeroed ? 1 : 0;
          // the first launch may drop land on load if k_land_fix restores it at the end (not for a one-launch filter)
          m.ring_first = (k == 1 && !is_last && (zero_land || pl->n_land == 0)) ? 1 : 0;

          for (int t = 0; t < S; ++t) m.pk[t] = p[k + t];
          m.p0 = p[0]; m.c = c; m.nbatch = nbatch; m.row_lo = 0; m.row_hi = rows;
          if ((rc = advance_multi(pl, m, s, &launches))) return rc;
          --launches;  // counted once more below
          u = fr[0]; v = fr[1];
          if (k == 1 && zero_land && !is_last) {  // keep the isolated cells out of the state from here on
            // (a first launch by k_ring already took them as zero while it loaded the field)
            if (!ring_supported(pl, m)) {
              if ((rc = launch_zero_land(pl, fr[0], fr[1], nbatch, s))) return rc;
            }   // (k_ring's first launch and k_fold_band took the land as zero while they loaded the field)
            land_zeroed = true;
          }
        } else {
          StepArgs a1{};
          a1.mode = (k == 1 ? GCMF_STEP_FIRST : 0u) | (is_last ? GCMF_STEP_LAST : 0u);
          a1.coef0 = (k == 1) ? p[0] : p[k]; a1.coef1 = p[1]; a1.c = c; a1.fb_is_f32 = fb32; a1.nbatch = nbatch;
          a1.row_lo = 0; a1.row_hi = rows;
          const void *src = u;
          if (k == 1 && prep) {  // the single-step kernel wants T_0 = field*area materialised
            if ((rc = launch_prepare(pl, din, Pp, nbatch, 0, rows, s))) return rc;
            ++launches;
            src = Pp[0];
          }
          a1.t1[0] = src; a1.t2[0] = v; a1.t0[0] = fr[0]; a1.fb_in[0] = Fcur; a1.fb_out[0] = is_last ? dout[0] : Fcur;
          if ((rc = step_dispatch(pl, a1, s))) return rc;
          v = src; u = fr[0];
        }
        ++launches;
        k += S;
      }
      if (land_zeroed) {  // the isolated cells' own polynomial, from the caller's untouched input
        if ((rc = ensure_dev_p(pl, p, n_steps, s))) return rc;
        if ((rc = launch_land_fix(pl, din[0], dout[0], pl->dev_p, n_steps, c, fb32 ? 1 : 0, nbatch, s))) return rc;
      }
    } else if (use_vmulti && ((pl->kind == K_CGRID && pl->clenshaw >= 1) ||
                              (pl->kind == K_BGRID && pl->clenshaw >= 2 && (pl->d.dtype == GCMF_F64 || back_f32))) && !fwd_only &&
               n_steps >= 2 && vec_multi_supported(pl, nbatch, 2)) {
      // C-grid (B-grid with GCMF_CLENSHAW=2: it is bit-exact with numpy forward, so backward is an option there like for the land-mask
      // kinds): the polynomial evaluated backwards (k_cgrid_stream2c / k_bgrid_stream2c): state (b_{k+1}, b_{k+2}) in a pool of four
      // plane pairs, the constant input (u, v) read by every launch, no fbar planes.  Level l = 1..n uses p[n - l]; the first
      // launch forms b_n = p[n] f as it loads f, the last one writes the result.
      const void *u[2] = {x0[0], x0[1]}, *v[2] = {nullptr, nullptr};
      void *pool[4][2] = {{A[0], A[1]}, {B[0], B[1]}, {Cb[0], Cb[1]}, {Db[0], Db[1]}};
      // four levels per launch with two operand rows in flight: 353-357 G on config 5; five levels leave one row in flight and
      // spill (305-310 G); the forward kernel at its best (five levels) 280 G
      // (round 5: k_cgrid_ring, gcmf_cgrid_ring.hip, takes batched f32 levels four or five at a time)
      // (launches deeper than five levels exist only in k_cgrid_ring, whose 16-byte accesses need the caller's planes aligned)
      const bool al16 = ptr_al16(x0[0]) && ptr_al16(x0[1]) && ptr_al16(dout[0]) && ptr_al16(dout[1]);
      const int smax = std::min(pl->multi_s, std::max(4, al16 ? cgrid_ring_smax(pl, nbatch) : std::min(5, cgrid_ring_smax(pl, nbatch))));
      int lvl = 1;
      while (lvl <= n_steps) {
        const int left = n_steps - lvl + 1;
        const int S = vec_backward_next_depth(pl, nbatch, left, smax);
        void *fr[2][2];
        int nf = 0;
        for (int q = 0; q < 4 && nf < 2; ++q)
          if (pool[q][0] != u[0] && pool[q][0] != v[0]) { fr[nf][0] = pool[q][0]; fr[nf][1] = pool[q][1]; ++nf; }
        const bool is_last = (lvl + S - 1 == n_steps);
        VecMultiArgs m{};
        for (int q = 0; q < 2; ++q) {
          m.u0[q] = u[q]; m.uprev[q] = v[q]; m.u1o[q] = fr[0][q]; m.u2o[q] = fr[1][q];
          m.fb_in[q] = x0[q]; m.fb_out[q] = dout[q];
        }
        for (int t = 0; t < S; ++t) m.pk[t] = p[n_steps - (lvl + t)];
        m.p0 = p[n_steps]; m.c = c; m.S = S; m.clen = 1;
        m.first = (lvl == 1); m.last = is_last; m.fb_is_f32 = fb32; m.nbatch = nbatch; m.row_lo = 0; m.row_hi = rows;
        if ((rc = dom_begin(pl, s))) return rc;
        if ((rc = launch_vec_multi(pl, m, s))) return rc;
        if ((rc = dom_end(pl, s))) return rc;
        for (int q = 0; q < 2; ++q) { u[q] = fr[1][q]; v[q] = fr[0][q]; }
        lvl += S;
        ++launches;
      }
    } else if (use_vmulti) {
      // vector kinds: S = 2..4 steps per pass, (T_{k-1}, T_{k-2}) -> (T_{k+S-2}, T_{k+S-1}).  Neither output may overwrite
      // T_{k-2}: the halo rows / columns a strip recomputes need its neighbours' T_{k-2}.  The state rotates through
      // four buffers.  A lone last step runs the single-step kernel.
      const void *u[2] = {x0[0], x0[1]}, *v[2] = {nullptr, nullptr};
      int k = 1;
      while (k <= n_steps) {
        const int left = n_steps - k + 1;
        void *pool[4][2] = {{A[0], A[1]}, {B[0], B[1]}, {Cb[0], Cb[1]}, {Db[0], Db[1]}};
        void *fr[2][2];
        int nf = 0;
        for (int q = 0; q < 4 && nf < 2; ++q)
          if (pool[q][0] != u[0] && pool[q][0] != v[0]) { fr[nf][0] = pool[q][0]; fr[nf][1] = pool[q][1]; ++nf; }
        int S = 1;
        const int cand[5] = {6, 5, 4, 3, 2};
        for (int q = 0; q < 5; ++q)  // largest depth that does not strand a lone single step at the end
          if (cand[q] <= left && left - cand[q] != 1 && cand[q] <= pl->multi_s && vec_multi_supported(pl, nbatch, cand[q])) {
            S = cand[q];
            break;
          }
        if (S == 1 && left >= 2) S = 2;
        if (S >= 2) {
          const bool is_last = (k + S - 1 == n_steps);
          VecMultiArgs m{};
          for (int q = 0; q < 2; ++q) {
            m.u0[q] = u[q]; m.uprev[q] = v[q]; m.u1o[q] = fr[0][q]; m.u2o[q] = fr[1][q];
            m.fb_in[q] = F[q]; m.fb_out[q] = is_last ? dout[q] : F[q];
          }
          for (int t = 0; t < S; ++t) m.pk[t] = p[k + t];
          m.p0 = p[0]; m.c = c; m.S = S;
          m.first = (k == 1); m.last = is_last; m.fb_is_f32 = fb32; m.nbatch = nbatch; m.row_lo = 0; m.row_hi = rows;
          if ((rc = dom_begin(pl, s))) return rc;
          if ((rc = launch_vec_multi(pl, m, s))) return rc;
          if ((rc = dom_end(pl, s))) return rc;
          for (int q = 0; q < 2; ++q) { u[q] = fr[1][q]; v[q] = fr[0][q]; }
          k += S;
        } else {
          StepArgs a1{};
          a1.mode = (k == 1 ? GCMF_STEP_FIRST : 0u) | GCMF_STEP_LAST;
          a1.coef0 = (k == 1) ? p[0] : p[k]; a1.coef1 = p[1]; a1.c = c; a1.fb_is_f32 = fb32; a1.nbatch = nbatch;
          a1.row_lo = 0; a1.row_hi = rows;
          for (int q = 0; q < 2; ++q) {
            a1.t1[q] = u[q]; a1.t2[q] = v[q]; a1.t0[q] = fr[0][q]; a1.fb_in[q] = F[q]; a1.fb_out[q] = dout[q];
          }
          if ((rc = step_dispatch(pl, a1, s))) return rc;
          k += 1;
        }
        ++launches;
      }
    } else {
    if (prep) {  // T_0 = field * area
      if ((rc = launch_prepare(pl, din, Pp, nbatch, 0, rows, s))) return rc;
      ++launches;
      for (int k = 0; k < nc; ++k) x0[k] = Pp[k];
    }
    // step k reads T_{k-1} (stencil) and T_{k-2} (centre) and overwrites T_{k-2}'s buffer with T_k:
    //   k=1: X0 -> A      k=2: (A, X0) -> B      k=3: (B, A) -> A      k=4: (A, B) -> B ...
    for (int k = 1; k <= n_steps; ++k) {
      StepArgs a{};
      a.mode = (k == 1 ? GCMF_STEP_FIRST : 0u) | (k == n_steps ? GCMF_STEP_LAST : 0u);
      a.coef0 = (k == 1) ? p[0] : p[k];
      a.coef1 = p[1];
      a.c = c;
      a.fb_is_f32 = fb32;
      a.nbatch = nbatch;
      a.row_lo = 0;
      a.row_hi = rows;
      for (int q = 0; q < nc; ++q) {
        if (k == 1) { a.t1[q] = x0[q]; a.t2[q] = nullptr; a.t0[q] = A[q]; }
        else if (k == 2) { a.t1[q] = A[q]; a.t2[q] = x0[q]; a.t0[q] = B[q]; }
        else if (k % 2) { a.t1[q] = B[q]; a.t2[q] = A[q]; a.t0[q] = A[q]; }
        else { a.t1[q] = A[q]; a.t2[q] = B[q]; a.t0[q] = B[q]; }
        a.fb_in[q] = F[q];
        a.fb_out[q] = (k == n_steps) ? dout[q] : F[q];
      }
      if ((rc = step_dispatch(pl, a, s))) return rc;
      ++launches;
    }
    }
  }
  if (timing) GCMF_HIP(hipEventRecord(pl->ev1, s));
  // "work buffers busy": recorded lazily by the NEXT call when it arrives on another stream (back-to-back calls on one stream pay for no
  // event).  That needs this stream to be alive then: a stream handed in by the caller (not the plan's own, not the null stream) may be
  // destroyed before the next call, so for such a stream the event is recorded now, while the handle is known to be good, whenever
  // the stream differs from the previous call's (a caller cycling through streams) -- the common case, one long-lived stream, stays free.
  if (s != pl->stream && s != nullptr && pl->busy_valid && pl->busy_stream != s) {
    GCMF_HIP(hipEventRecord(pl->ev_busy, s));
    pl->busy_recorded = true;
  } else {
    pl->busy_recorded = false;
  }
  pl->busy_stream = s;
  pl->busy_valid = true;
  pl->last_launches = timed ? launches : pl->last_launches + launches;
  if (!on_dev) {
    for (int k = 0; k < nc; ++k)
      GCMF_HIP(hipMemcpyAsync(out[k], dout[k], ncell * fbs, hipMemcpyDeviceToHost, s));
    GCMF_HIP(hipStreamSynchronize(s));
    const unsigned rlo = pl->res_lo, rhi = pl->res_hi;
    pl->res_lo = pl->res_hi = 0;   // (the stream is drained: nothing of this call is pending any more)
    if (rlo && resident_take_failure(pl->d.device, rlo, rhi)) {   // (the synchronising host path can tell for THIS call)
      set_error("k_resident: the on-chip launch timed out waiting for a neighbour tile: the result is NaN (another process running "
                "resident kernels on this GPU?); the strip-marching launches are used from now on");
      return GCMF_ERR_HIP;
    }
  }
  if (timing) {
    GCMF_HIP(hipEventSynchronize(pl->ev1));
    GCMF_HIP(hipEventElapsedTime(&pl->last_ms, pl->ev0, pl->ev1));
  }
  if (pl->timing_detail && timed && (rc = dom_collect(pl))) return rc;
  return GCMF_OK;
}

// Host pointers and a batch of fields: the batch is cut into chunks of ~32 MB per component that stream through two
// staging slots in HBM -- upload of chunk k+1 and download of chunk k-1 run on their own streams while chunk k is
// filtered ("overlap H2D of chunk k+1 with compute of chunk k").  The host issues upload(k+1) and the
// launches of chunk k+1 BEFORE download(k), so that a blocking download into pageable memory still overlaps with compute.
static int run_host_pipelined(gcmf_plan *pl, const double *p, int n_steps, double c, const void *const *in,
                              void *const *out, int64_t nbatch, uint32_t flags, void *stream, bool lapl_only,
                              int64_t chunk_nb) {
  const int nc = pl->ncomp;
  const bool f32 = pl->d.dtype == GCMF_F32;
  const size_t ts = dtype_size(pl->d.dtype);
  const size_t fbs = lapl_only ? ts : ((f32 && !(flags & GCMF_OUT_F32)) ? 8 : ts);
  const size_t cell = (size_t)pl->d.ny * pl->d.nx;
  const size_t szI = align_up((size_t)chunk_nb * cell * ts, 256), szO = align_up((size_t)chunk_nb * cell * fbs, 256);
  const size_t need = (size_t)nc * 2 * (szI + szO);
  if (need > pl->stage_bytes) {
    if (pl->stage) GCMF_HIP(hipFree(pl->stage));
    pl->stage = nullptr;
    pl->stage_bytes = 0;
    GCMF_HIP(hipMalloc(&pl->stage, need));
    pl->stage_bytes = need;
  }
  if (!pl->s_in) {
    GCMF_HIP(hipStreamCreateWithFlags(&pl->s_in, hipStreamNonBlocking));
    GCMF_HIP(hipStreamCreateWithFlags(&pl->s_out, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) {
      GCMF_HIP(hipEventCreateWithFlags(&pl->ev_in[q], hipEventDisableTiming));
      GCMF_HIP(hipEventCreateWithFlags(&pl->ev_cmp[q], hipEventDisableTiming));
      GCMF_HIP(hipEventCreateWithFlags(&pl->ev_out[q], hipEventDisableTiming));
    }
  }
  hipStream_t s_cmp = stream ? (hipStream_t)stream : pl->stream;
  char *base = (char *)pl->stage;
  auto In = [&](int slot, int k) { return base + ((size_t)(slot * nc + k)) * szI; };
  auto Out = [&](int slot, int k) { return base + (size_t)2 * nc * szI + ((size_t)(slot * nc + k)) * szO; };
  const int64_t nchunks = (nbatch + chunk_nb - 1) / chunk_nb;
  auto nb_of = [&](int64_t ch) { return ch == nchunks - 1 ? nbatch - ch * chunk_nb : chunk_nb; };
  const uint32_t dflags = flags | GCMF_DEVICE_PTRS;
  int rc = GCMF_OK;
  pl->last_launches = 0;

  auto upload_and_launch = [&](int64_t ch) -> int {
    const int slot = (int)(ch & 1);
    const int64_t nb = nb_of(ch);
    if (ch >= 2) GCMF_HIP(hipStreamWaitEvent(pl->s_in, pl->ev_cmp[slot], 0));  // In[slot] was read by chunk ch-2
    for (int k = 0; k < nc; ++k)
      GCMF_HIP(hipMemcpyAsync(In(slot, k), (const char *)in[k] + (size_t)ch * chunk_nb * cell * ts, (size_t)nb * cell * ts,
                              hipMemcpyHostToDevice, pl->s_in));
    GCMF_HIP(hipEventRecord(pl->ev_in[slot], pl->s_in));
    GCMF_HIP(hipStreamWaitEvent(s_cmp, pl->ev_in[slot], 0));
    if (ch >= 2) GCMF_HIP(hipStreamWaitEvent(s_cmp, pl->ev_out[slot], 0));  // Out[slot] was drained by chunk ch-2
    if (ch == 0 && pl->timing) GCMF_HIP(hipEventRecord(pl->ev0, s_cmp));
    const void *din[2] = {In(slot, 0), nc > 1 ? In(slot, 1) : nullptr};
    void *dout[2] = {Out(slot, 0), nc > 1 ? Out(slot, 1) : nullptr};
    int r = run_whole_locked(pl, p, n_steps, c, din, dout, nb, dflags, (void *)s_cmp, lapl_only, false);
    if (r) return r;
    if (ch == nchunks - 1 && pl->timing) GCMF_HIP(hipEventRecord(pl->ev1, s_cmp));
    GCMF_HIP(hipEventRecord(pl->ev_cmp[slot], s_cmp));
    return GCMF_OK;
  };
  auto download = [&](int64_t ch) -> int {
    const int slot = (int)(ch & 1);
    const int64_t nb = nb_of(ch);
    GCMF_HIP(hipStreamWaitEvent(pl->s_out, pl->ev_cmp[slot], 0));
    for (int k = 0; k < nc; ++k)
      GCMF_HIP(hipMemcpyAsync((char *)out[k] + (size_t)ch * chunk_nb * cell * fbs, Out(slot, k), (size_t)nb * cell * fbs,
                              hipMemcpyDeviceToHost, pl->s_out));
    GCMF_HIP(hipEventRecord(pl->ev_out[slot], pl->s_out));
    return GCMF_OK;
  };

  // Page-lock the caller's input for the duration of the call: uploads from pageable memory block the host and do
  // not overlap with the downloads (2.5 ms per 69 MB field); from registered memory they are plain asynchronous DMA and
  // the pipeline runs at the filter's own rate (1.65 ms).  Registration is best effort (already page-locked or
  // read-only mappings simply stay as they are).
  bool registered[2] = {false, false};
  if (pl->host_register)
    for (int k = 0; k < nc; ++k) {
      registered[k] = host_register(in[k], (size_t)nbatch * cell * ts);
    }
  auto finish = [&](int r) {
    (void)hipStreamSynchronize(pl->s_in);
    (void)hipStreamSynchronize(pl->s_out);
    (void)hipStreamSynchronize(s_cmp);
    for (int k = 0; k < nc; ++k)
      if (registered[k]) host_unregister(in[k]);
    return r;
  };
  // Order of the enqueues (round 5): the download of chunk ch goes out BEFORE the upload of chunk ch + 2, so that an upload that blocks the
  // calling thread (memory that could not be page-locked) never holds a download back.  Measured (tools/measure_host_batch.py, 12 fields of
  // 2400 x 3600 f64): 2.7 ms per field either way with a freshly registered input, 1.6 ms with an input that is ALREADY page-locked (a torch
  // pinned tensor): what separates the two is the per-call hipHostRegister / unregister of the caller's array (~1 ms per 69 MB), not the
  // order of the copies -- and a registration cache would only help a caller that passes the same buffer again.
  if ((rc = upload_and_launch(0))) return finish(rc);
  if (nchunks > 1 && (rc = upload_and_launch(1))) return finish(rc);
  for (int64_t ch = 0; ch < nchunks; ++ch) {
    if ((rc = download(ch))) return finish(rc);
    if (ch + 2 < nchunks && (rc = upload_and_launch(ch + 2))) return finish(rc);
  }
  if ((rc = finish(GCMF_OK))) return rc;
  if (pl->timing) GCMF_HIP(hipEventElapsedTime(&pl->last_ms, pl->ev0, pl->ev1));
  return GCMF_OK;
}

static int run_whole(gcmf_plan *pl, const double *p, int n_steps, double c, const void *const *in, void *const *out,
                     int64_t nbatch, uint32_t flags, void *stream, bool lapl_only) {
  if (!pl || !in || !out || nbatch < 0 || (!lapl_only && (!p || n_steps < 1))) {
    set_error("gcmf_apply: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  for (int k = 0; k < pl->ncomp; ++k)
    if (!in[k] || !out[k]) {
      set_error("gcmf_apply: null component pointer");
      return GCMF_ERR_INVALID_ARG;
    }
  for (int k = 0; k < pl->ncomp; ++k)
    for (int q = 0; q < pl->ncomp; ++q)
      if (in[k] == out[q]) {  // the input is read by the first launch, by neighbouring strips and by k_land_fix at the end
        set_error("gcmf_apply: `out` must not alias `in` (filtering in place is not supported)");
        return GCMF_ERR_INVALID_ARG;
      }
  if (!pl->full) {
    set_error("gcmf_apply / gcmf_laplacian need a plan covering the whole grid; use gcmf_cheb_step on row slabs");
    return GCMF_ERR_INVALID_ARG;
  }
  if (nbatch == 0) return GCMF_OK;
  std::lock_guard<std::mutex> lk(pl->mu);
  GCMF_HIP(hipSetDevice(pl->d.device));
  if (!(flags & GCMF_DEVICE_PTRS) && nbatch > 1 && pl->host_chunk_bytes > 0) {
    const size_t entry = (size_t)pl->d.ny * pl->d.nx * dtype_size(pl->d.dtype);
    int64_t chunk_nb = (int64_t)(pl->host_chunk_bytes / entry);
    if (chunk_nb < 1) chunk_nb = 1;
    if (chunk_nb > MAX_LAUNCH_BATCH) chunk_nb = MAX_LAUNCH_BATCH;
    if (chunk_nb < nbatch) return run_host_pipelined(pl, p, n_steps, c, in, out, nbatch, flags, stream, lapl_only, chunk_nb);
  }
  if (nbatch <= MAX_LAUNCH_BATCH)
    return run_whole_locked(pl, p, n_steps, c, in, out, nbatch, flags, stream, lapl_only, true);
  // very long batches of small fields: the scalar kernels index the batch with gridDim.y (<= 65535)
  const size_t cell = (size_t)pl->d.ny * pl->d.nx, ts = dtype_size(pl->d.dtype);
  const size_t fbs = lapl_only ? ts : ((pl->d.dtype == GCMF_F32 && !(flags & GCMF_OUT_F32)) ? 8 : ts);
  float ms_total = 0.f;
  int launches = 0;
  for (int64_t b0 = 0; b0 < nbatch; b0 += MAX_LAUNCH_BATCH) {
    const int64_t nb = nbatch - b0 < MAX_LAUNCH_BATCH ? nbatch - b0 : MAX_LAUNCH_BATCH;
    const void *in2[2] = {nullptr, nullptr};
    void *out2[2] = {nullptr, nullptr};
    for (int k = 0; k < pl->ncomp; ++k) {
      in2[k] = (const char *)in[k] + (size_t)b0 * cell * ts;
      out2[k] = (char *)out[k] + (size_t)b0 * cell * fbs;
    }
    int rc = run_whole_locked(pl, p, n_steps, c, in2, out2, nb, flags, stream, lapl_only, true);
    if (rc) return rc;
    ms_total += pl->last_ms;
    launches += pl->last_launches;
  }
  pl->last_ms = ms_total;
  pl->last_launches = launches;
  return GCMF_OK;
}

int gcmf_apply(gcmf_plan *pl, const double *p, int n_steps, double c, const void *const *in, void *const *out,
               int64_t nbatch, uint32_t flags, void *stream) {
  return run_whole(pl, p, n_steps, c, in, out, nbatch, flags, stream, false);
}

int gcmf_laplacian(gcmf_plan *pl, const void *const *in, void *const *out, int64_t nbatch, uint32_t flags,
                   void *stream) {
  return run_whole(pl, nullptr, 0, 0.0, in, out, nbatch, flags, stream, true);
}

}  // extern "C"
