// libgcmf C ABI, part 2: the row-slab building blocks (single steps, blocked launches, on-chip levels, land helpers) and the drivers that
// run a whole backward application on one rank's slab (gcmf_slab_apply_backward[_vec]); the cuts of a polynomial into launches that
// gcmf_apply (gcmf_api.hip) and these drivers share.  See include/gcmf.h for the contract.
#include "gcmf_api_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gcmf {

bool land_ok(const gcmf_plan *pl, int n_steps) {
  return pl && (pl->kind == K_FLUX || pl->kind == K_MASK) && pl->zero_land && pl->lbits && pl->n_land > 0 && (pl->d.nx % 4) == 0 && n_steps < 4096;
}

// Nine levels per launch (k_ringc<double, K_FLUX, 9>): whole f64 flux-form grids without a tripole seam (the seam's k_fold_band and the
// slabs' early-exit form stop at eight), tall enough for the deeper ghost zone.
bool ringc9_ok(const gcmf_plan *pl) {
  return pl && pl->ringc9 && pl->kind == K_FLUX && pl->d.dtype == GCMF_F64 && pl->full && !pl->tripolar && !pl->g.fold && pl->g.rows >= 64;
}

// ... and whole tripolar f64 flux grids whose launches advance the seam themselves (k_ringcz's fold strips, round 6): no k_fold_band (which
// stops at eight levels) is involved then.  Depends on the batch: a packed batch keeps the band.
static bool ringc9_fold_ok(const gcmf_plan *pl, int64_t nbatch) {
  if (!(pl && pl->ringc9 && pl->kind == K_FLUX && pl->d.dtype == GCMF_F64 && pl->full && pl->g.fold && pl->g.rows >= 64)) return false;
  MultiArgs a{};
  a.S = 9;
  a.nbatch = nbatch;
  a.row_lo = 0;
  a.row_hi = pl->g.rows;
  return ringc_zip_fold_ok(pl, a);
}

// ... and ROW SLABS of f64 flux grids without a tripole seam, when the slab's owner says so (option "slab_nines": SlabFilter sets it on every
// rank or on none -- the ranks of a run must cut alike, and only they know whether all of them qualify and the ghost zone is nine rows deep).
// Nothing here may depend on the slab's size beyond the 64-row floor: edge and interior ranks own different numbers of rows.
static bool ringc9_slab_ok(const gcmf_plan *pl) {
  return pl && pl->slab_nines && pl->ringc9 && pl->kind == K_FLUX && pl->d.dtype == GCMF_F64 && !pl->full && !pl->tripolar && !pl->g.fold && pl->g.rows >= 64;
}

// Backward (Clenshaw) evaluation (gcmf_ringc_impl.hpp): whether gcmf_apply uses it for this plan and polynomial, and how the
// n_steps levels are cut into launches of 5..8 (never leaving 1..4 or 9 behind).  plan->clenshaw = 1: the flux kinds, whose
// launches run at memcpy rate and gain the plane they no longer move (config 3: +10 %); 2: every scalar kind (the land-mask
// kernel is bound by its instruction stream and gains nothing: 93 -> 92-95 us per launch).  Needs the isolated cells fixed up
// by k_land_fix when there is land (land_ok).
int clenshaw_cut(const gcmf_plan *pl, int n_steps, int *depths, int max_depths, bool f32_asked, int64_t nbatch) {
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
  const bool nines_full = ringc9_ok(pl) || ringc9_fold_ok(pl, nbatch), nines = nines_full || ringc9_slab_ok(pl);
  if (!(n_steps >= 10 || (n_steps >= 5 && n_steps <= 8) || (n_steps == 9 && nines))) return 0;
  int smax = pl->ringc_smax;
  if (!smax && nbatch == 1 && nines_full && pl->ringc_zip && (long long)pl->g.rows * pl->g.nx <= 2500000LL && n_steps >= 10) {
    // Whole grids that live in the caches and run k_ringcz (1/4-degree grids): a launch is paced by the rows its strips march, not by the
    // bytes it moves, so fewer launches are not always faster -- a strip marches H + S + 1 rows whose cost grows with S, and fewer
    // levels mean narrower ghost columns (sometimes a window less).  Measured (experiments/scripts/zip_ab.py, us per launch):
    // ~4 + rows x (0.6 + 0.045 S); 1080 x 1440 n 63: 7 x 9 levels 225, 8 launches of <= 8 215, 9 x 7 217; 720 x 1440: 158 / 170 / 169.
    double best = 0.0;
    for (int S = 9; S >= 7; --S) {
      const int M = (S + 1) / 2 * 2, WI = 128 - 2 * M, L = (n_steps + S - 1) / S;
      int march = 0;
      if (ringc_zip_pairs((pl->g.nx + WI - 1) / WI, 1, pl->g.rows, S, &march) < 1) continue;
      const double t = L * (4.0 + march * (0.6 + 0.045 * S));
      if (best == 0.0 || t < 0.98 * best) { best = t; smax = S; }
    }
    if (smax == 9) smax = 0;
  }
  if (smax >= 5 && smax <= 7) {   // at most smax levels per launch, as even as possible
    const int L = (n_steps + smax - 1) / smax, q = n_steps / L, r = n_steps % L;
    if (q >= 5 && L <= max_depths) {
      for (int k = 0; k < L; ++k) depths[k] = q + (k < r ? 1 : 0);
      return L;
    }
  }
  if (nines && smax != 8 && (n_steps + 8) / 9 < (n_steps + 7) / 8) {
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

// The vector kinds' counterpart (a C-grid / B-grid field sharded over y used to run the Python choreography of
// distributed.py, forward, ~15 us of host time per launch): one whole backward (Clenshaw) application of a VECTOR plan on this rank's slab
// in one call.  X / out: the two components; pool: four state plane PAIRS, pool[2 q + comp]; all (nbatch, rows_alloc, nx).  The levels are
// cut exactly as gcmf_apply cuts them for this plan (at most four per launch), so a level filtered on a slab and in one piece see the same
// arithmetic; a launch of S levels uses up S ghost rows, the ghost zone is refreshed (both states, both components: four planes in one
// message per neighbour) when fewer are left than the next launch needs.  No edge / interior split.
// Levels of the next launch of a backward VECTOR application with `left` levels to go and at most smax per launch: the fewest launches,
// their depths evened out (44 levels at up to six per launch: 6 6 6 6 5 5 5 5 -- greedy sixes would leave a two-level launch of the
// general kernel at the end), never a lone single level left behind; gcmf_apply and the slab driver cut alike.
bool ptr_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int vec_backward_next_depth(const gcmf_plan *pl, int64_t nbatch, int left, int smax) {
  if (smax >= 5) {
    const int nl = (left + smax - 1) / smax, S = (left + nl - 1) / nl;
    if (S >= 2 && S <= left && left - S != 1 && vec_multi_supported(pl, nbatch, S, true)) return S;
  }
  for (int cand = smax; cand >= 2; --cand)
    if (cand <= left && left - cand != 1 && vec_multi_supported(pl, nbatch, cand, true)) return cand;
  return left;
}

}  // namespace gcmf

using namespace gcmf;

extern "C" {

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
    if (pl->ncomp != 1 || S < 5 || S > ((ringc9_ok(pl) || ringc9_slab_ok(pl)) ? 9 : 8) || !pl->ring || !pl->zero_row || clenshaw_cut(pl, 10, probe, 2, false, 2) != 2 ||
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
  {   // an on-chip launch of this plan's PREVIOUS application timed out (its result is NaN): told once, here (as gcmf_apply does)
    std::lock_guard<std::mutex> lk(pl->mu);
    if (pl->res_lo) {
      const unsigned rlo = pl->res_lo, rhi = pl->res_hi;
      pl->res_lo = pl->res_hi = 0;
      if (resident_take_failure(pl->d.device, rlo, rhi)) {
        set_error("k_resident: the previous on-chip application of this slab plan timed out waiting for a neighbour tile and its result is NaN; "
                  "the strip-marching launches are used from now on");
        return GCMF_ERR_HIP;
      }
    }
  }
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
  // f's ghost rows: the first launch forms b_n = p_n f on them, the later ones read f on the rows they compute.  Batches (round 6): the
  // rows of the first launch whose S-level cone stays inside the owned rows need none of them, so that exchange -- the only one of an
  // application when the ghost zone is as deep as the filter -- runs BESIDE them, and the two edge pieces follow (one launch cut in
  // three).  MEASURED, round 6 (tools/measure_batched_scaling.py, 300-row slab of 2400 x 3600, same box, off / on): 16 fields RCCL 1.922 /
  // 1.974 ms, mailboxes 1.969 / 2.076; 8 fields RCCL 1.185 / 1.134, mailboxes 1.158 / 1.214; config 4, 16 fields 1.672 / 1.742 -- the two
  // edge pieces cost what the hidden exchange saves.  OFF unless GCMF_SLAB_OVERLAP_FIRST=1.
  static const bool first_beside = getenv("GCMF_SLAB_OVERLAP_FIRST") && atoi(getenv("GCMF_SLAB_OVERLAP_FIRST")) != 0;
  bool first_pending = false;
  {
    void *st[1] = {X};
    if ((rc = exchange_start(st, 1))) return rc;
    first_pending = multi && first_beside && nbatch >= 2 && nbatch * ro >= 2000 && ro >= 4 * (int64_t)cut[0];
    if (!first_pending && (rc = exchange_finish())) return rc;
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
    if (first_pending && ovl) {   // (both at once is not worth a fourth piece: the input's ghost rows first, then as below)
      if ((rc = exchange_finish())) return rc;
      first_pending = false;
    }
    if (first_pending) {
      const int ilo = gs ? (int)(fo + S) : lo, ihi = gn ? (int)(fo + ro - S) : hi;
      if ((rc = launch(ilo, ihi))) return rc;
      if ((rc = exchange_finish())) return rc;
      first_pending = false;
      if (gs && (rc = launch(lo, ilo))) return rc;
      if (gn && (rc = launch(ihi, hi))) return rc;
    } else if (ovl) {
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

}  // extern "C"
