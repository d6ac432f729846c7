// Internal declarations shared by the libgcmf translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "gcmf.h"

namespace gcmf {

// Stencil families.  Every reference Laplacian maps onto one of them after plan-time folding.
enum Kind : int {
  K_REG = 0,   // REGULAR, REGULAR_AREA_WEIGHTED: 5-point, no coefficients              (40 B/cell.step f64)
  K_MASK = 1,  // *_WITH_LAND regular grids + tripolar regular: 1 byte of neighbour bits (41 B)
  K_FLUX = 2,  // IRREGULAR / POP / MOM5U / MOM5T: east-face, north-face, 1/area planes  (64 B)
  K_CGRID = 3, // VECTOR_C_GRID: 14 folded planes
  K_BGRID = 4  // VECTOR_B_GRID: 8 folded planes
};

// internal mode bit on top of GCMF_STEP_FIRST / GCMF_STEP_LAST
constexpr unsigned STEP_LAPL = 0x100u;  // t0 = L(t1) only (gcmf_laplacian)

constexpr int MAX_COEF = 14;

void set_error(const char *fmt, ...);
const char *hip_err_name(hipError_t e);

#define GCMF_HIP(call)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      ::gcmf::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? GCMF_ERR_NO_DEVICE : GCMF_ERR_HIP; \
    }                                                                                     \
  } while (0)

// Device-side view of the slab geometry + coefficient planes, passed by value to kernels.
struct Geom {
  int nx;          // columns (never sharded; x is periodic)
  int rows;        // rows in the slab allocation (owned + ghost)
  int south_wrap;  // row 0's southern neighbour is row rows-1 (single slab, periodic in y)
  int north_wrap;  // row rows-1's northern neighbour is row 0
  int fold;        // row rows-1's northern neighbour is itself mirrored in x (tripole seam)
  int area_weighted;
  const void *coef[MAX_COEF];  // K_FLUX: cE, cN, ra;  K_CGRID: 14;  K_BGRID: 8
  const uint8_t *mbits;        // K_MASK: bit0 wet, bit1 E wet, bit2 W wet, bit3 N wet, bit4 S wet, bits 5-7 their count
  const void *area;            // prepare / finalize plane (area-weighted types)
};

// Arguments of one recurrence step (device pointers, per component).
struct StepArgs {
  const void *t1[2];
  const void *t2[2];
  const void *fb_in[2];
  void *t0[2];
  void *fb_out[2];
  double coef0, coef1, c;
  unsigned mode;
  int fb_is_f32;  // f32 plans only: fbar arrays are f32 (GCMF_OUT_F32)
  int64_t nbatch;
  int row_lo, row_hi;
  int fb_lo;  // scalar kinds: rows < fb_lo leave fbar untouched
  int rpw;    // scalar kinds: rows marched per wave (0 = the plan's / default 2); the tripole band steps use 1
};

// Arguments of one temporally blocked launch: S recurrence steps in one pass (scalar kinds, one component).
struct MultiArgs {
  const void *u0, *v0;      // T_{k-1}, T_{k-2}
  void *uo, *vo;            // T_{k-1+S}, T_{k-2+S}  (must not alias u0 / v0)
  const void *fb_in;
  void *fb_out;
  double pk[9];             // coefficients of the S steps (p[k] .. p[k+S-1]); with `first`: p[1] .. p[S]
  double p0, c;
  int S, first, last, fb_is_f32;
  int64_t nbatch;
  int row_lo, row_hi;
  int land_zero;  // the caller guarantees that isolated (land) cells of u0 and v0 are zero: GCMF_STEP_LAND_ZERO
  int ring_first; // first launch: the caller will overwrite the result of the isolated (land) cells (k_land_fix) or there are
                  // none, so the launch may take them as zero while it loads the field (k_ring<..., FIRST>)
  int zip_fold = 0;  // k_ringcz: [row_lo, row_hi) ends at the tripole seam and the launch advances the seam rows itself (no k_fold_band)
};

// Arguments of one S-step vector launch (gcmf_cgrid_stream2.hip / gcmf_bgrid_stream2.hip): T_{k-1}, T_{k-2} -> T_{k+S-2}, T_{k+S-1}.
struct VecMultiArgs {
  const void *u0[2];     // T_{k-1} (u, v)
  const void *uprev[2];  // T_{k-2}             (ignored with `first`)
  void *u1o[2];          // T_{k+S-2} out (must not alias u0 / uprev)
  void *u2o[2];          // T_{k+S-1} out (must not alias u0 / uprev / u1o)
  const void *fb_in[2];
  void *fb_out[2];
  double pk[8];          // p[k] .. p[k+S-1]; with `first`: p[1] .. p[S]
  double p0, c;
  int S, first, last, fb_is_f32;
  int64_t nbatch;
  int row_lo, row_hi;
  int clen;              // backward (Clenshaw) evaluation: u0 / uprev = (b_{k+1}, b_{k+2}) (first: u0 = the input f, uprev unused), fb_in =
                         // the input f, p0 = p_n, pk[t] = coefficient of level t + 1, fb_out = the result (last launch)
};

}  // namespace gcmf

struct gcmf_plan {
  gcmf_plan_desc d{};
  int kind = 0;
  int ncomp = 1;
  bool tripolar = false;
  bool area_weighted = false;
  bool dimensional = false;
  bool full = true;  // single slab covering the whole grid
  int64_t rows_alloc = 0, first_owned = 0, rows_owned = 0;
  gcmf::Geom g{};
  std::vector<void *> owned;  // device allocations freed at destroy
  // work buffers of gcmf_apply (grow-only)
  void *work = nullptr;
  size_t work_bytes = 0;
  hipStream_t side = nullptr;           // k_fold_band (the tripole seam rows) runs here, concurrently with the blocked launch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t stream = nullptr;
  // pipelined host path (batched host arrays): staging slots, upload / download streams, per-slot events
  void *stage = nullptr;
  size_t stage_bytes = 0;
  size_t host_chunk_bytes = 32u << 20;  // per component and chunk; 0 = no pipelining (env GCMF_HOST_CHUNK_MB)
  int host_register = 1;                // page-lock the caller's input during a pipelined call (env GCMF_HOST_REGISTER)
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_busy = nullptr;  // end of the last gcmf_apply that used the plan's work buffers
  bool busy_valid = false;
  bool busy_recorded = false;   // ev_busy already stands behind the last call's work (recorded eagerly: see run_whole_locked)
  hipStream_t busy_stream = nullptr;  // the stream that call ran on
  bool timing = false;
  // gcmf_set_timing(plan, 2): an event pair around every temporally blocked launch of gcmf_apply (the dominant kernel)
  bool timing_detail = false;
  std::vector<hipEvent_t> dom_ev;  // pairs
  std::vector<std::string> dom_name;  // kernel each pair bracketed
  std::string last_launched;          // name of the most recent recurrence kernel launch
  int dom_used = 0;
  float dom_ms = 0.f, dom_min = 0.f, dom_max = 0.f;
  int dom_n = 0;
  // the recurrence kernel with the most steps per launch since gcmf_last_kernel was last read (instrumentation:
  // bench.py ties its HBM-traffic figures to the kernel that actually ran)
  std::string last_kernel;
  int last_kernel_weight = 0;
  std::string last_geom;   // launch geometry of last_kernel (note_kernel)
  float last_ms = 0.f;
  int last_launches = 0;
  int rows_per_wave = 0;
  int xcd_remap = 1;
  int ringc_xe_rows = 64;   // flux plans without a seam: strips shorter than this MAY run the early-exit form k_ringcs (launch_ringc decides; GCMF_RINGC_XE_ROWS=0: never)
  int zigzag = 1;        // k_ringc (flux kinds): neighbouring strips march in opposite directions (GCMF_ZIGZAG=0: all downwards)
  int multi_s = 8;     // steps fused per pass by the temporally blocked kernel (1 = off); 8 measured best
  int strip_rows = 0;  // rows per wave strip of that kernel (0 = auto)
  int prefetch_rows = 0;  // rows of operands in flight per wave (0 = default per S)
  int cgrid_tile = 0;     // 1: force the LDS-tile C-grid kernel instead of the streaming one (A/B testing)
  int cgrid_ring = 1;     // k_cgrid_ring (gcmf_cgrid_ring.hip) for batched f32 levels (0: k_cgrid_stream2c everywhere); env GCMF_CGRID_RING, gcmf_set_option
  int cgrid_ring_smax = 6;  // levels per launch of that kernel (4 / 5 / 6; six since round 6); env GCMF_CGRID_RING_SMAX
  int ring_flux_f32 = 1;    // forward ring kernel for f32 flux-kind state (0: k_flux_multi2 as until round 4); gcmf_set_option
  int cgrid_ring_hmax = 0;  // tallest strip its launcher picks (0 = 96); gcmf_set_option
  int cgrid_ring_ncarry = 0;  // 0: the levels carry their previous row's scaled copies in registers (round 6; six levels: the top two), 1: only the last one (round 5); gcmf_set_option
  // Land kept out of the state (scalar plans; slab-row layout): bit 0 of lbits[cell] = the cell exchanges with a neighbour.
  // A cell that does not (land under a wet mask; a flux-form cell whose four faces are closed) has L = 0 and evolves
  // on its own: gcmf_apply zeroes such cells in the two states the first blocked launch wrote -- NaN on land then never
  // reaches the NaN / inf bookkeeping of the blocked kernels -- and writes their polynomial into the result with
  // k_land_fix at the end.
  const uint8_t *lbits = nullptr;
  int64_t n_land = 0;
  int zero_land = 1;      // env GCMF_ZERO_LAND=0 turns it off
  int ring = 1;           // env GCMF_RING=0: deep launches stay with k_flux_multi2 / k_scalar_multi
  unsigned *ring_nfb = nullptr;    // device counter behind gcmf_ring_fallbacks (lives behind zero_row)
  const void *zero_row = nullptr;  // nx zeros: what rows beyond a closed boundary read as coefficients / mask bits (k_ring)
  int single_launch = 0;  // whole f64 flux grids: the whole polynomial in ONE persistent launch (k_ringc_one; measured slower); gcmf_set_option / GCMF_SINGLE_LAUNCH
  int pack_batch = 1;     // k_ringc / k_ringcs: the fields of a batch as one column per window (ringc_walk, round 6); gcmf_set_option "pack_batch"
  int ringc_zip = 1;      // f64 flux plans without a tripole seam: k_ringcz where it marches fewer rows (env GCMF_RINGC_ZIP, gcmf_set_option "ringc_zip")
  long long band_seq_cells = 3000000;   // tripolar plans: blocked launches over at most this many cells run k_fold_band AFTER themselves (its 1024-thread form), not beside (env GCMF_BAND_SEQ_CELLS; 0 = never)
  bool alone_now = true;  // (set by advance_multi for the blocked launch it issues: no k_fold_band waves will share its SIMDs)
  int zip_fold = 1;       // tripolar f64 flux plans, backward evaluation: k_ringcz advances the seam's rows itself (no k_fold_band); gcmf_set_option "zip_fold", env GCMF_ZIP_FOLD
  int slab_nines = 0;     // row slabs of f64 flux grids without a tripole seam: nine levels per launch where that saves one (gcmf_set_option "slab_nines"; every rank or none)
  int ringc_smax = 0;     // backward scalar launches: at most this many levels each (5..8; 0 = the default cut: nine where offered, else eight); gcmf_set_option "ringc_smax"
  int ringc9 = 1;         // whole f64 flux-form grids without a tripole seam: up to NINE levels per k_ringc launch (env GCMF_RINGC9, gcmf_set_option "ringc9")
  int clenshaw = 2;       // backward (Clenshaw) evaluation: 0 off, 1 the flux kinds + C-grid, 2 (default since round 4) every kind that has a
                          // backward kernel (f64 REGULAR / land-mask kinds, B-grid too); env GCMF_CLENSHAW.  Per call: GCMF_FORWARD_RECURRENCE
  int clenshaw_f32 = 0;   // backward evaluation also for f32 SCALAR state and the f32 B-grid (round 5: off by default -- an all-f32 Clenshaw
                          // sum is 15-45 x less accurate than the reference's own f32 path (f32 T_k, f64 running sum) on the scalar kinds,
                          // 2-3 x on the B-grid, and the forward kernels reproduce that path bit for bit on the REGULAR / land-mask /
                          // B-grid kinds; the f32 C-grid stays backward: in Reinsch's form it is MORE accurate than the reference's).
                          // GCMF_CLENSHAW_F32=1, gcmf_set_option("clenshaw_f32"), or per call GCMF_BACKWARD_F32.
  void *resident = nullptr;  // state of the on-chip (resident) kernel: exchange planes, tile flags (gcmf_resident.hip)
  unsigned res_lo = 0, res_hi = 0;   // serial numbers of the resident launches of the last gcmf_apply that took that path (0: none pending)
  int last_path = 0;                 // GCMF_PATH_* of the last gcmf_apply
  long long path_count[5] = {0, 0, 0, 0, 0};
  double *dev_p = nullptr;   // p[0..n_steps] of the last filter, for k_land_fix
  size_t dev_p_n = 0;
  std::vector<double> host_p;
  std::mutex mu;
};

namespace gcmf {
template <typename T> inline const char *tyname() { return sizeof(T) == 8 ? "double" : "float"; }
// name as rocprofv3 prints it (without "void " and the argument list); weight = recurrence steps per launch
// geom: the launch geometry of blocked kernels ("H=.. nstrips=.. nwx=.. xcd=.. grid=..x.. rows=.."), what a PMC traffic
// record of the kernel is only valid for (gcmf_last_kernel_geometry; bench.py load_traffic)
inline void note_kernel(gcmf_plan *pl, const std::string &name, int weight, const std::string &geom = std::string()) {
  pl->last_launched = name;
  if (weight >= pl->last_kernel_weight) {
    pl->last_kernel = name;
    pl->last_kernel_weight = weight;
    pl->last_geom = geom;
  }
}
inline std::string launch_geom(int H, int nstrips, int nwx, int xcd, unsigned gx, unsigned gy, int nrows) {
  char b[160];
  snprintf(b, sizeof b, "H=%d nstrips=%d nwx=%d xcd=%d grid=%ux%u rows=%d", H, nstrips, nwx, xcd, gx, gy, nrows);
  return b;
}
size_t dtype_size(int dtype);
// How many strips a (window, batch entry) column of the one-wave-per-SIMD strip-marching kernels is cut into: ideally as many as fill ONE
// resident round of 1024 waves (all strips march in lock-step) -- a single field at BASELINE size: 33 windows x 31 strips.  Batches do
// not divide that well (33 windows x 16 fields = 528 columns: one strip each left half the SIMDs without a wave -- a batch of 16 ran at
// 515 G against 811 G for a batch of 4 before round 5): the number of rounds k <= 8 is chosen that minimises k x (rows a wave marches:
// H + 2 S, rounded up to the exit period), one round being preferred by 4 % per extra round.
inline long long strips_per_column(long long per_strip, long long nrows, int S, int period) {
  if (per_strip < 1) per_strip = 1;
  long long best = 1;
  double best_cost = -1.0;
  for (int k = 1; k <= 8; ++k) {
    long long w = (1024LL * k) / per_strip;
    if (w < 1) w = 1;
    if (w > nrows) w = nrows;
    const long long H = (nrows + w - 1) / w;
    long long march = H + 2 * S;
    if (period > 1) march = (march + period - 1) / period * period;
    const long long rounds = (w * per_strip + 1023) / 1024;
    const double cost = (double)(rounds * march) * (1.0 + 0.04 * (rounds - 1));
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = w; }
    if (w >= nrows) break;
  }
  return best;
}
// k_ringcz (gcmf_ringc_impl.hpp): strips zipped in pairs
// rows a ZIP march of `need` rows runs -- with early exits: one every second row up to eight levels, every fourth at nine; without: whole
// ring periods (taken when it is no longer: 80 fewer registers, the same time per row -- 1080 x 1440 at eight levels, 24 rows either way:
// 215.2 against 214.7 us)
inline long long ringc_zip_rows(long long need, int S, bool *xe) {
  const long long ex = S <= 8 ? 2 : 4;
  const long long mx = std::max(12LL, (need + ex - 1) / ex * ex), mp = (need + 11) / 12 * 12;
  if (xe) *xe = mx < mp;
  return std::min(mx, mp);
}
// pairs per window: whole rounds of the 1024 wave slots, strips of at least two rows; *march = rows the launch marches (all rounds)
inline int ringc_zip_pairs(long long nwx, long long nbatch, long long nrows, int S, int *march) {
  long long best = 0, best_cost = 0;
  for (int k = 1; k <= 8; ++k) {
    long long np = (512LL * k) / std::max(1LL, nwx * nbatch);
    np = std::min(np, nrows / 4);
    if (np < 1) continue;
    const long long H = (nrows + 2 * np - 1) / (2 * np);                 // the taller strips
    const long long m = ringc_zip_rows(H + S + 1, S, nullptr);
    const long long rounds = (2 * np * nwx * nbatch + 1023) / 1024;
    const long long cost = rounds * m * (100 + 4 * (rounds - 1));
    if (!best || cost < best_cost) { best = np; best_cost = cost; if (march) *march = (int)(rounds * m); }
    if (np >= nrows / 4) break;
  }
  return (int)best;
}

// What a batch costs WITHOUT k_ringcz, in rows marched (x 1.04 per extra round of the wave slots): the better of whole strips per field and
// the packed column of launch_ringc_sf (gcmf_ringc_impl.hpp: the same formulas) -- for the policies that weigh the zipped strips against it
inline double ringc_batch_cost(long long nwx, long long nbatch, long long nrows, int S, int exitp, bool pack) {
  auto padded = [&](long long m) { return (m + exitp - 1) / exitp * exitp; };
  const long long want = strips_per_column(nwx * nbatch, nrows, S, exitp);
  long long H = std::max(4LL, (nrows + want - 1) / want);
  if (exitp == 12) H += (12 - (H + 2 * S) % 12) % 12;
  H = std::min(H, nrows);
  const long long nstrips = (nrows + H - 1) / H, rounds_u = (nwx * nbatch * nstrips + 1023) / 1024;
  double best = (double)(rounds_u * padded(H + 2 * S)) * (1.0 + 0.04 * (rounds_u - 1));
  if (!pack || nbatch <= 1) return best;
  const long long total = nbatch * nrows, slots = std::max(1LL, 1024LL / nwx);
  for (long long k = 1; k <= 16; ++k) {
    const long long w = std::min(total, slots * k), q = (total + w - 1) / w;
    if (q > nrows || q > 320) continue;
    const long long rounds = (w * nwx + 1023) / 1024;
    const bool crosses = (nrows % q) != 0;
    const double cost = (double)(rounds * (padded(q + 2 * S) + (crosses ? padded(2 * S + exitp / 2) : 0))) * (1.0 + 0.04 * (rounds - 1));
    if (cost < 0.97 * best) best = cost;
    if (w >= total) break;
  }
  return best;
}
// kernel launchers (defined in gcmf_scalar.hip / gcmf_vector.hip)
int launch_scalar_step(gcmf_plan *pl, const StepArgs &a, hipStream_t s);
int launch_vector_step(gcmf_plan *pl, const StepArgs &a, hipStream_t s);
int launch_scalar_multi(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
bool flux_multi2_supported(const gcmf_plan *pl, int S);
bool ring_supported(const gcmf_plan *pl, const MultiArgs &a);
int launch_ring_flux_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);   // gcmf_ring_flux_f32.hip
// backward (Clenshaw) evaluation, gcmf_ringc_impl.hpp: one launch of S = 5..8 levels; a.fb_in = the constant input f, a.fb_out =
// the result (last launch), a.pk[t] = the coefficient of level t + 1 of this launch
int launch_ringc_reg(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_maskz(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_flux(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_flux9(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);   // nine levels: whole f64 flux grids without a seam (gcmf_ringc_flux9.hip)
int launch_ringc_flux_slab(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
int launch_ringc_zip(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);   // f64 flux plans without a tripole seam, short strips: pairs of strips zipped at a shared seam (k_ringcz, gcmf_ringc_zip.hip)
int ringc_zip_march(const gcmf_plan *pl, const MultiArgs &a, int *pairs);
bool ringc_zip_fold_ok(const gcmf_plan *pl, const MultiArgs &a);   // can k_ringcz advance the tripole seam's rows of this launch itself?  // rows a k_ringcz launch would march (all rounds), 0 = not offered
int launch_ringc_flux_slab_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);   // flux plans without a tripole seam, short strips: early exits (k_ringcs)
int launch_flux_multi2(gcmf_plan *pl, const MultiArgs &a, hipStream_t s);
// the on-chip kernel (gcmf_resident.hip): L <= 64 levels of the backward evaluation in ONE launch on a field that fits the register
// files + LDS of the chip (short slabs, small grids); pk = the L coefficients (a.S / a.pk are ignored)
bool resident_supported(const gcmf_plan *pl, int row_lo, int row_hi, int L, int n_total, int *why = nullptr);   // fits AND the policy says so (n_total: levels of the whole filter; why: GCMF_RESIDENT_* when not)
bool resident_fits(const gcmf_plan *pl, int row_lo, int row_hi, int L);
int launch_resident(gcmf_plan *pl, const MultiArgs &a, const double *pk, int L, hipStream_t s);
void resident_free(gcmf_plan *pl);
bool resident_take_failure(int dev, unsigned lo, unsigned hi);   // did one of the resident launches with serials [lo, hi] time out?  (reported once)
void resident_status(int dev, int *state, unsigned long long *failures);
// another persistent kernel of this process (k_ringc_one): under the on-chip lock, chained behind the process's other persistent launches;
// launch(flags, fail word (device address of mapped host memory), serial, first value of the arrival counter flags[1001]); nbar = arrivals it adds
int resident_persistent_launch(int dev, hipStream_t s, unsigned nbar, const std::function<int(unsigned *, unsigned *, unsigned, unsigned)> &launch);
// the whole polynomial of a whole f64 flux grid in ONE launch (gcmf_ringc_one.hip; opt-in): levels per pass (0: not this application)
int ringc_one_depth(const gcmf_plan *pl, int n_steps, int64_t nbatch);
int launch_ringc_one(gcmf_plan *pl, int S, const double *p, int n_steps, double c, const void *f, void *out, void *const *pool, hipStream_t s);
int launch_cgrid_stream(gcmf_plan *pl, const StepArgs &a, hipStream_t s);
bool cgrid_stream_supported(const gcmf_plan *pl, const StepArgs &a);
int launch_bgrid_stream(gcmf_plan *pl, const StepArgs &a, hipStream_t s);
bool bgrid_stream_supported(const gcmf_plan *pl, const StepArgs &a);
bool multi_supported(const gcmf_plan *pl, int S);
bool cgrid_multi_supported(const gcmf_plan *pl, int64_t nbatch, int S, bool backward = false);   // (backward: k_cgrid_ring's deeper launches count)
int launch_cgrid_multi(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s);
// the static-ring form of the backward C-grid kernel (gcmf_cgrid_ring.hip): batched f32 levels, S = 4 .. 8
bool cgrid_ring_supported(const gcmf_plan *pl, int64_t nbatch, int S);
bool cgrid_ring_args_aligned(const VecMultiArgs &a);   // the caller's state / input / result planes on 16-byte boundaries
int cgrid_ring_smax(const gcmf_plan *pl, int64_t nbatch);   // deepest launch it offers this plan / batch (0: none)
int launch_cgrid_ring(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s);
// ... and of the forward recurrence with an f64 running sum (gcmf_cgrid_ringf.hip: Filter(evaluation="reference") on batched f32 levels)
bool cgrid_ringf_supported(const gcmf_plan *pl, const VecMultiArgs &a);
int launch_cgrid_ringf(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s);
bool bgrid_multi_supported(const gcmf_plan *pl, int64_t nbatch, int S);
int launch_bgrid_multi(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s);
inline bool vec_multi_supported(const gcmf_plan *pl, int64_t nbatch, int S, bool backward = false) {
  return pl->kind == K_CGRID ? cgrid_multi_supported(pl, nbatch, S, backward) : bgrid_multi_supported(pl, nbatch, S);
}
inline int launch_vec_multi(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  return pl->kind == K_CGRID ? launch_cgrid_multi(pl, a, s) : launch_bgrid_multi(pl, a, s);
}
// S fused steps on rows [row_lo,row_hi) incl. the tripole band when the range ends at the fold row (gcmf_api.hip)
// backward: m = the arguments of a k_ringc launch (the polynomial evaluated backwards)
int advance_multi(gcmf_plan *pl, const MultiArgs &m, hipStream_t s, int *launches, bool backward = false);
// the tripole seam rows of an S-step launch in one launch (gcmf_foldband.hip)
bool fold_band_supported(const gcmf_plan *pl, const MultiArgs &a);
int launch_fold_band(gcmf_plan *pl, const MultiArgs &a, bool backward, hipStream_t s, bool wide = false);
int launch_prepare(gcmf_plan *pl, const void *const *in, void *const *out, int64_t nbatch, int row_lo,
                   int row_hi, hipStream_t s);
// the isolated cells' own polynomial, written over out (gcmf_landfix.hip); dp = p[0..n_steps] on the device
int launch_zero_land(gcmf_plan *pl, void *a, void *b, int64_t nbatch, hipStream_t s, int row_lo = 0, int row_hi = 0);
int launch_land_fix(gcmf_plan *pl, const void *in, void *out, const double *dp, int n_steps, double c, int fb_is_f32,
                    int64_t nbatch, hipStream_t s);
// plan-time precompute (gcmf_precompute.hip): fills pl->g from the raw global planes (device pointers)
int precompute(gcmf_plan *pl, const void *const *dplanes, const void *const *hplanes_or_null);
}  // namespace gcmf
