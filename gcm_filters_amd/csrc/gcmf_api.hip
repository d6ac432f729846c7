// libgcmf C ABI: plan lifetime, the whole-polynomial apply loop and the per-step building blocks.
// See include/gcmf.h for the contract and the reference interfaces each entry point replaces.
#include "gcmf_api_internal.hpp"

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
      if (!pl->g.fold || pl->alone_now) {   // (nothing has to fit beside these waves: no seam in this launch, or its band runs afterwards)
        const bool f64 = pl->d.dtype == GCMF_F64;
        const int wi = f64 ? (m.S == 9 ? 108 : 112) : 240;   // useful columns of a window (f32: four cells per lane)
        const long long nrows = m.row_hi - m.row_lo, per = ((pl->g.nx + wi - 1) / wi) * std::max<long long>(1, m.nbatch);
        const long long want = strips_per_column(per, nrows, m.S, 12);
        const long long H0 = std::min(nrows, std::max(4LL, pl->strip_rows > 0 ? (long long)pl->strip_rows : (nrows + want - 1) / want));
        const long long need = H0 + 2 * m.S;
        const long long rows_xe = std::max(12LL, (need + 3) / 4 * 4), rows_pad = (need + 11) / 12 * 12;
        const bool xe = m.S <= 8 && H0 < pl->ringc_xe_rows && rows_xe * 100 <= rows_pad * (f64 ? 95 : 90);
        // Round 6: strips zipped in pairs at a shared seam (k_ringcz) march H + S + 1 rows instead of H + 2 S: where strips are as short as
        // their ghost zones (1/4-degree grids: 15 rows behind 2 x 9; the 300-row slab of one of eight ranks: 11 behind 2 x 8) -- and, for
        // less, at BASELINE size (2400 x 3600: 30 strips of 80 rows marching 92 instead of 27 of 90 marching 108: 890.5 against 906 us,
        // same box, alternating; the launch is bound by HBM there)
        const long long rounds = (per * ((nrows + H0 - 1) / H0) + 1023) / 1024;
        const int mz = ringc_zip_march(pl, m, nullptr);
        if (mz > 0 && m.nbatch <= 1 && (long long)mz * 100 <= rounds * (xe ? rows_xe : rows_pad) * 90) return launch_ringc_zip(pl, m, s);
        // batches: against the better of whole strips per field and the packed column (1/4-degree grids, 2 .. 8 fields: + 3 .. 20 %)
        if (mz > 0 && m.nbatch > 1 && f64 &&
            mz * 100.0 <= ringc_batch_cost((pl->g.nx + wi - 1) / wi, m.nbatch, nrows, m.S, xe ? 4 : 12, pl->pack_batch != 0) * 90.0)
          return launch_ringc_zip(pl, m, s);
        if (xe) return launch_ringc_flux_slab(pl, m, s);
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
  // Short launches on the seam's plan (round 6): the band AFTER the blocked launch, in its stream, 1024 threads per tile.  Beside a launch
  // that lasts no longer than itself the band is the slower of the two (its waves share the SIMDs with the marching waves) and the fork /
  // join costs ~5 us on top: a 1080 x 1440 tripolar grid took 294 us against 215 us for the same grid without a seam.
  const bool seq = band && pl->band_seq_cells > 0 && (long long)m.nbatch * (m.row_hi - m.row_lo) * g.nx <= pl->band_seq_cells;
  pl->alone_now = !band || seq;
  if (!band) {
    if ((rc = dom_begin(pl, s))) return rc;
    if ((rc = blocked(m))) return rc;
    if ((rc = dom_end(pl, s))) return rc;
    if (launches) ++*launches;
    return GCMF_OK;
  }
  if (backward && ringc_zip_fold_ok(pl, m)) {   // (round 6) the seam's rows inside the launch: strips that start at the seam, zipped with their mirror windows
    MultiArgs mz = m;
    mz.zip_fold = 1;
    pl->alone_now = true;
    if ((rc = dom_begin(pl, s))) return rc;
    if ((rc = launch_ringc_zip(pl, mz, s))) return rc;
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
  if (seq) {
    if (mm.row_hi > mm.row_lo) {
      if ((rc = dom_begin(pl, s))) return rc;
      if ((rc = blocked(mm))) return rc;
      if ((rc = dom_end(pl, s))) return rc;
      if (launches) ++*launches;
    }
    if ((rc = launch_fold_band(pl, m, backward, s, true))) return rc;
    if (launches) ++*launches;
    return GCMF_OK;
  }
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

int step_dispatch(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  return pl->ncomp == 1 ? launch_scalar_step(pl, a, s) : launch_vector_step(pl, a, s);
}

// p[0..n_steps] on the device for k_land_fix; uploaded only when it changed (a pageable upload stalls the host behind
// the stream)
int ensure_dev_p(gcmf_plan *pl, const double *p, int n_steps, hipStream_t s) {
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
  if (const char *e = getenv("GCMF_RINGC_ZIP")) pl->ringc_zip = atoi(e);
  if (const char *e = getenv("GCMF_BAND_SEQ_CELLS")) pl->band_seq_cells = atoll(e);
  if (const char *e = getenv("GCMF_ZIP_FOLD")) pl->zip_fold = atoi(e);
  if (const char *e = getenv("GCMF_PACK_BATCH")) pl->pack_batch = atoi(e);
  if (const char *e = getenv("GCMF_SINGLE_LAUNCH")) pl->single_launch = atoi(e);
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
    const int n_clen = (use_multi && !fwd_only) ? clenshaw_cut(pl, n_steps, depths, 1024, back_f32, nbatch) : 0;
    // Small fields: the whole polynomial on the chip in ONE launch (64 levels at a time; gcmf_resident.hip) -- the field, both states
    // and the coefficients live in registers / LDS, nothing but the result goes back to memory.  Same bits as the launches below.
    bool resident = false;
    int path = GCMF_PATH_STRIPS;
    if (pl->res_lo) {   // the LAST call of this plan ran on the chip: did one of ITS launches time out?  (told once, to the plan whose
      //                   output was poisoned -- never to an unrelated plan; the process runs the strip-marching launches from now on)
      const unsigned lo = pl->res_lo, hi = pl->res_hi;
      pl->res_lo = pl->res_hi = 0;
      if (resident_take_failure(pl->d.device, lo, hi)) {
        set_error("the previous on-chip / single-launch application of this plan (k_resident, k_ringc_one) timed out waiting for another "
                  "workgroup and its result is NaN (another process running persistent kernels on this GPU outside the lock file's reach?); "
                  "the back-to-back strip-marching launches are used from now on");
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
      pl->res_lo = pl->res_hi = 0;   // (launch_resident notes the serial numbers of this application's launches)
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
    } else if (n_clen > 0 && !(flags & GCMF_NO_RESIDENT) && ringc_one_depth(pl, n_steps, nbatch) > 0 &&
               [&]() -> bool {   // OPT-IN ("single_launch"): the whole polynomial in ONE persistent launch (gcmf_ringc_one.hip); taken when
                 //               the process may run persistent kernels now, else the back-to-back launches below (same bits)
                 void *pool[4] = {A[0], B[0], Cb[0], Db[0]};
                 if (dom_begin(pl, s)) return false;
                 const int r1 = launch_ringc_one(pl, ringc_one_depth(pl, n_steps, nbatch), p, n_steps, c, din[0], dout[0], pool, s);
                 if (dom_end(pl, s)) return false;
                 if (r1 == GCMF_OK) ++launches;
                 return r1 == GCMF_OK;
               }()) {
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
