// k_fold_band: the tripole seam of a temporally blocked launch in ONE launch.
//
// On tripolar grids (reference gcm_filters/kernels.py:33-40, 469-487, 517-585) the northern neighbour of cell (ny-1, i) is
// (ny-1, nx-1-i): the top row couples every column with its mirror image, i.e. with a DIFFERENT wave of the strip-marching
// kernels (k_ring / k_ringc / k_scalar_multi), which therefore stop S rows below the seam.  The top S rows ("band") of an
// S-step launch are advanced here: level t = 1..S is computed on rows [rows-2S+t, rows) -- a ghost zone below the band that
// shrinks by one row per level, the same trick the multi-GPU slabs use -- and only the band rows [rows-S, rows) are stored.
//
//   * a workgroup owns a PAIR of mirrored column windows (wx, its mirror image): 32 columns each incl. 8 halo columns per side
//     that go stale by one per level (the windows of the strip-marching kernels do the same), 2S rows; both state levels of
//     the tile live in LDS, so the fold partner of a top-row cell is a plain LDS read from the sister window (a wave owns a
//     tile row of both windows).  Narrow windows = many small workgroups (113 on a 3600-column grid): the band is a chain of S
//     dependent levels, so its time is the time of ONE workgroup;
//   * T_t overwrites T_{t-2} in place (only the centre value of T_{t-2} is needed): two state arrays, one barrier per level;
//   * coefficients / mask bytes / the running sum (forward) / the constant input (backward) of the tile are LDS arrays too, so
//     the kernel needs few registers and its waves fit on the SIMDs next to the 400-450-register waves of the blocked launch
//     that runs beside it on the main stream (gcmf_api.hip advance_multi): the seam costs no CUs and no extra time;
//   * arithmetic: FORWARD = the single-step kernel's (gcmf_scalar.hip: nan_to_num on the stencil operands, raw centre value in
//     "-x", gcmf_recurrence.hpp helpers), so the band is bit-identical to S single steps like every blocked kernel;
//     BACKWARD = k_ringc's march (gcmf_ringc_impl.hpp): b_k = p_k f + 2 A(b_{k+1}) - b_{k+2};
//   * nan_to_num is the identity on finite data: a workgroup only runs its stencil operands through it from the level on at
//     which one of its values is not finite (a workgroup-wide OR rides on the level barrier) -- same results, 40 % fewer
//     instructions on clean data.
//
// Replaces the S dependent k_scalar_step launches (~10 us each) rounds 1-2 ran on the side stream.
#include "gcmf_multi_common.hpp"

namespace gcmf {

constexpr int FB_WW = 32;            // columns of a window (a wave owns one tile row of both windows)
constexpr int FB_M = 8;              // halo columns per side (>= the deepest launch)
constexpr int FB_WI = FB_WW - 2 * FB_M;
constexpr int FB_TR = 2 * MAX_S;     // tile rows (2 S are used)
constexpr int FB_CELLS = FB_TR * 2 * FB_WW;

template <typename T, typename FB> struct FoldBandP {
  const T *u0, *v0;        // level 0 / level -1 (forward: T_{k-1}, T_{k-2}; backward: b_{k+1}, b_{k+2}); first: u0 = the field (forward)
  T *uo, *vo;              // level S / level S-1 on the band rows
  const FB *fb_in;         // forward: running sum in
  FB *fb_out;              // forward: running sum out / finalised result (last); backward: the result (last)
  const T *f;              // backward: the constant input
  double *d_out;           // backward, last launch, f32 state: f64 result (NumPy >= 2 promotion) or NULL
  const T *cE, *cN, *ra;
  const uint8_t *mbits, *lbits;
  const T *area;
  int nx, rows, S, npairs;
  int first, last, area_weighted, zero_land;
  long long bstride;
  double pk[MAX_S];
  double p0, c;
};

// NT = 256: the form that runs BESIDE the blocked launch (four tile rows per pass, few registers).  NT = 1024 (round 6): one cell per thread,
// run AFTER the blocked launch in its stream where that launch is short (1/4-degree grids, the top rank's slab of an 8-way run): alone on
// the chip the band is a few microseconds, beside a launch that lasts no longer than itself it is the slower of the two and costs a
// fork / join on top.
template <typename T, typename FB, int KIND, bool BACK, int NT>
__global__ __launch_bounds__(NT) void k_fold_band(const FoldBandP<T, FB> P) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr bool FLUX = (KIND == K_FLUX);
  constexpr bool FUSED = FLUX || BACK;   // gcmf_recurrence.hpp; the backward evaluation fuses every multiply-add pair
  T *sA = reinterpret_cast<T *>(smem);
  T *sB = sA + FB_CELLS;
  T *sC0 = sB + FB_CELLS;                       // FLUX: cE | BACK: (after the coefficient arrays) f
  T *scE = sC0, *scN = scE + (FLUX ? FB_CELLS : 0), *sra = scN + (FLUX ? FB_CELLS : 0);
  T *sF = sra + (FLUX ? FB_CELLS : 0);           // BACK: the constant input (prepared, land out)
  FB *sFB = reinterpret_cast<FB *>(sF + (BACK ? FB_CELLS : 0));   // forward: running sum (band rows)
  uint8_t *smb = reinterpret_cast<uint8_t *>(sFB + (BACK ? 0 : FB_CELLS));   // K_MASK: mask bytes

  const int tid = threadIdx.x;
  constexpr int RSTEP = NT / (2 * FB_WW);   // tile rows a pass of the workgroup covers
  constexpr int NPASS = FB_TR / RSTEP;
  const int q = tid & (FB_WW - 1), w = (tid / FB_WW) & 1, tr0 = tid / (2 * FB_WW);
  const int S = P.S, nx = P.nx, rows = P.rows;
  const int ntr = 2 * S;
  const int x0 = (int)blockIdx.x * FB_WI;
  // global column of (w, q): window 0 runs east from x0 - M, window 1 is its mirror image
  int col;
  {
    const int c0 = x0 - FB_M + (w == 0 ? q : FB_WW - 1 - q);
    col = (w == 0) ? c0 : nx - 1 - c0;
    col %= nx;
    if (col < 0) col += nx;
  }
  const long long boff = (long long)blockIdx.y * P.bstride;
  const T c = (T)P.c;
  const bool weigh = !FLUX && P.area_weighted;
  const int row_base = rows - ntr;

  // ---- load the tile: level 0, level -1, constants -------------------------------------------------------------------------
  bool bad = false;   // a value of this thread is not finite
#pragma unroll 2   // two rows of loads in flight; more would push the kernel beyond the 64 registers a k_ring wave leaves on its SIMD
  for (int m = 0; m < NPASS; ++m) {
    const int tr = tr0 + RSTEP * m;
    if (tr >= ntr) continue;
    const int cell = (tr * 2 + w) * FB_WW + q;
    const long long g = (long long)(row_base + tr) * nx + col;
    const bool keep = !P.zero_land || (P.lbits[g] & 1u);
    const T ar = weigh ? P.area[g] : T(1);
    if constexpr (FLUX) {
      scE[cell] = P.cE[g];
      scN[cell] = P.cN[g];
      sra[cell] = P.ra[g];
    } else {
      smb[cell] = P.mbits[g];
    }
    if constexpr (BACK) {
      T fv = P.f[boff + g];
      if (weigh) fv = fv * ar;                    // prepare(): f * area (kernels.py:100-101)
      fv = keep ? fv : T(0);                      // isolated cells stay out of the state (k_land_fix writes their polynomial)
      sF[cell] = fv;
      T x;
      if (P.first) {                              // b_n = p_n f, b_{n+1} = 0
        x = keep ? (T)P.p0 * fv : T(0);
        sB[cell] = T(0);
      } else {
        x = P.u0[boff + g];
        sB[cell] = P.v0[boff + g];
      }
      sA[cell] = x;
      bad = bad || !(mabs(x) <= MLim<T>::big());
    } else {
      T x = P.u0[boff + g];
      if (P.first) {
        if (weigh) x = x * ar;
        x = keep ? x : T(0);
        sB[cell] = T(0);
      } else {
        sB[cell] = P.v0[boff + g];
        if (tr >= S) sFB[cell] = P.fb_in[boff + g];
      }
      sA[cell] = x;
      bad = bad || !(mabs(x) <= MLim<T>::big());
    }
  }
  int sani = __syncthreads_or(bad);   // workgroup-uniform: some stencil operand of this tile needs nan_to_num

  // ---- S levels ---------------------------------------------------------------------------------------------------------------
  T *cur = sA, *prv = sB;
  const int qe = q < FB_WW - 1 ? q + 1 : q, qw = q > 0 ? q - 1 : q;
  auto level = [&](auto sani_c, const int t, const T *cur, T *prv) -> bool {
    constexpr bool SANI = decltype(sani_c)::value;
    auto san = [](T v) { return SANI ? msan(v) : v; };   // kernels.py:175, 300, 472, 566: the stencil sees nan_to_num(field)
    const T pk = (T)P.pk[t - 1];
    const T two = (BACK && P.last && t == S) ? T(1) : T(2);   // the result  p_0 f + A(b_1) - b_2: A, not 2 A
    bool nf = false;
#pragma unroll 1
    for (int m = 0; m < NPASS; ++m) {
      const int tr = tr0 + RSTEP * m;
      if (tr >= ntr) break;
      if (tr < t) continue;                       // below the shrinking ghost zone
      const int line = (tr * 2 + w) * FB_WW;
      const int cell = line + q;
      const T xraw = cur[cell];
      const T gC = san(xraw);
      const T gE = san(cur[line + qe]), gW = san(cur[line + qw]);
      const T gS = san(cur[cell - 2 * FB_WW]);
      const T gN = san(tr == ntr - 1 ? cur[(tr * 2 + (1 - w)) * FB_WW + (FB_WW - 1 - q)]   // the fold: [rows-1, nx-1-i]
                                    : cur[cell + 2 * FB_WW]);
      T L;
      if constexpr (FLUX) {   // kernels.py:302-314, 571-584 with plan-time folded face coefficients (gcmf_scalar.hip)
        const T fe = (gE - gC) * scE[cell];
        const T fw = (gC - gW) * scE[line + qw];
        const T fn = (gN - gC) * scN[cell];
        const T fs = (gC - gS) * scN[cell - 2 * FB_WW];
        L = ((fe - fw) + (fn - fs)) * sra[cell];
      } else {
        const unsigned b = smb[cell];
        if constexpr (BACK) {   // k_ringc's land-mask form (land is zero in the state)
          const T wf = (T)(b >> 5);
          L = rfma(-wf, gC, gE);
          L = L + gW;
          L = L + gN;
          L = L + gS;
          L = (b & 1u) ? L : T(0);
        } else {                // kernels.py:175-186, numpy's evaluation order (gcmf_scalar.hip K_MASK)
          const T mC = (b & 1u) ? gC : T(0);
          const T wf = (T)(b >> 5);
          L = -wf * mC + ((b & 2u) ? gE : T(0));
          L = L + ((b & 4u) ? gW : T(0));
          L = L + ((b & 8u) ? gN : T(0));
          L = L + ((b & 16u) ? gS : T(0));
          L = (b & 1u) ? L : T(0);
        }
      }
      const T a = cheb_a<FUSED>(xraw, c, L);      // "-x" takes the raw value: a NaN stays in its cell (filter.py:166-175)
      T tk;
      if constexpr (BACK) {
        tk = rfma(two, a, -prv[cell]);
        tk = rfma(pk, sF[cell], tk);
      } else {
        if (P.first && t == 1) {
          tk = a;
          if (tr >= S) sFB[cell] = cheb_acc_first<FUSED, T, FB>(P.p0, P.pk[0], xraw, a);
        } else {
          tk = cheb_t<FUSED>(a, prv[cell]);
          if (tr >= S) sFB[cell] = cheb_acc<FUSED, T, FB>(sFB[cell], P.pk[t - 1], tk);
        }
      }
      prv[cell] = tk;                              // T_t over T_{t-2}: only this thread reads that cell of `prv`
      nf = nf || !(mabs(tk) <= MLim<T>::big());
    }
    return nf;
  };
#pragma unroll 1
  for (int t = 1; t <= S; ++t) {
    const bool nf = sani ? level(std::true_type{}, t, cur, prv) : level(std::false_type{}, t, cur, prv);
    sani = __syncthreads_or(nf || sani);          // (the level barrier) a value that is not finite: nan_to_num from here on
    T *sw = cur;
    cur = prv;
    prv = sw;
  }

  // ---- store the band rows of the window interiors --------------------------------------------------------------------------
  if (q < FB_M || q >= FB_WW - FB_M) return;
  if (x0 + (w == 0 ? q : FB_WW - 1 - q) - FB_M >= nx) return;   // (duplicates of a tiny grid)
#pragma unroll 1
  for (int m = 0; m < NPASS; ++m) {
    const int tr = tr0 + RSTEP * m;
    if (tr >= ntr) break;
    if (tr < S) continue;
    const int cell = (tr * 2 + w) * FB_WW + q;
    const long long g = (long long)(row_base + tr) * nx + col;
    if (!P.last) {
      P.uo[boff + g] = cur[cell];
      P.vo[boff + g] = prv[cell];
      if constexpr (!BACK) P.fb_out[boff + g] = sFB[cell];
    } else if constexpr (BACK) {
      T r = cur[cell];
      if (weigh) r = r / P.area[g];               // finalize(): / area (kernels.py:103-104)
      if (P.d_out) P.d_out[boff + g] = (double)r;
      else P.fb_out[boff + g] = (FB)r;
    } else {
      FB r = sFB[cell];
      if (weigh) r = r / (FB)P.area[g];
      P.fb_out[boff + g] = r;
    }
  }
}

template <typename T, typename FB, int KIND, bool BACK, int NT>
static int launch_fb_nt(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  const Geom &g = pl->g;
  FoldBandP<T, FB> P{};
  P.u0 = (const T *)a.u0;
  P.v0 = (const T *)a.v0;
  P.uo = (T *)a.uo;
  P.vo = (T *)a.vo;
  if (BACK) {
    P.f = (const T *)a.fb_in;
    P.fb_out = (FB *)a.fb_out;
    if (sizeof(T) == 4 && !a.fb_is_f32) {   // f32 state: the result is f64 unless the caller asked for f32 (GCMF_OUT_F32)
      P.d_out = (double *)a.fb_out;
      P.fb_out = nullptr;
    }
  } else {
    P.fb_in = (const FB *)a.fb_in;
    P.fb_out = (FB *)a.fb_out;
  }
  P.cE = (const T *)g.coef[0];
  P.cN = (const T *)g.coef[1];
  P.ra = (const T *)g.coef[2];
  P.mbits = g.mbits;
  P.lbits = pl->lbits;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.S = a.S;
  P.npairs = ((g.nx + 1) / 2 + FB_WI - 1) / FB_WI;
  P.first = a.first;
  P.last = a.last;
  P.area_weighted = (KIND == K_FLUX) ? 0 : g.area_weighted;
  // backward: f is masked in every launch (k_ringc does the same); forward: only a first launch whose land k_land_fix restores
  P.zero_land = (pl->lbits && pl->n_land > 0 && (BACK || (a.first && a.ring_first))) ? 1 : 0;
  P.bstride = (long long)g.rows * g.nx;
  for (int t = 0; t < MAX_S; ++t) P.pk[t] = t < a.S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  if (a.nbatch <= 0) return GCMF_OK;
  size_t lds = (size_t)FB_CELLS * sizeof(T) * (2 + (KIND == K_FLUX ? 3 : 0) + (BACK ? 1 : 0)) + (BACK ? 0 : (size_t)FB_CELLS * sizeof(FB)) +
               (KIND == K_FLUX ? 0 : (size_t)FB_CELLS);
  static bool attr_set = false;  // per instantiation
  if (!attr_set && lds > 48 * 1024) {
    GCMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fold_band<T, FB, KIND, BACK, NT>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_fold_band<T, FB, KIND, BACK, NT>), dim3((unsigned)P.npairs, (unsigned)a.nbatch), dim3(NT), lds, s, P);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}
template <typename T, typename FB, int KIND, bool BACK>
static int launch_fb(gcmf_plan *pl, const MultiArgs &a, hipStream_t s, bool wide) {
  return wide ? launch_fb_nt<T, FB, KIND, BACK, 1024>(pl, a, s) : launch_fb_nt<T, FB, KIND, BACK, 256>(pl, a, s);
}

bool fold_band_supported(const gcmf_plan *pl, const MultiArgs &a) {
  return pl && pl->g.fold && (pl->kind == K_MASK || pl->kind == K_FLUX) && a.S >= 1 && a.S <= MAX_S && pl->g.rows >= 2 * a.S &&
         a.nbatch <= 65535;
}

// The top S rows of an S-step launch on a plan whose last row is the tripole seam.  `backward`: a = the arguments of a k_ringc
// launch (a.fb_in = the constant input f), otherwise of a forward launch.
// (wide: 1024 threads per tile, for a band that runs alone)
int launch_fold_band(gcmf_plan *pl, const MultiArgs &a, bool backward, hipStream_t s, bool wide) {
  if (!fold_band_supported(pl, a)) {
    set_error("k_fold_band: not a tripolar scalar plan / depth %d", a.S);
    return GCMF_ERR_UNSUPPORTED;
  }
  const bool f64 = pl->d.dtype == GCMF_F64, flux = pl->kind == K_FLUX;
  if (backward) {
    if (f64) return flux ? launch_fb<double, double, K_FLUX, true>(pl, a, s, wide) : launch_fb<double, double, K_MASK, true>(pl, a, s, wide);
    return flux ? launch_fb<float, float, K_FLUX, true>(pl, a, s, wide) : launch_fb<float, float, K_MASK, true>(pl, a, s, wide);
  }
  if (f64) return flux ? launch_fb<double, double, K_FLUX, false>(pl, a, s, wide) : launch_fb<double, double, K_MASK, false>(pl, a, s, wide);
  if (a.fb_is_f32) return flux ? launch_fb<float, float, K_FLUX, false>(pl, a, s, wide) : launch_fb<float, float, K_MASK, false>(pl, a, s, wide);
  return flux ? launch_fb<float, double, K_FLUX, false>(pl, a, s, wide) : launch_fb<float, double, K_MASK, false>(pl, a, s, wide);
}

}  // namespace gcmf
