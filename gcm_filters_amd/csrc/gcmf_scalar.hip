// Scalar Chebyshev-step kernels for gfx950 (MI355X): REGULAR / land-mask / flux-form 5-point stencils.
//
// One launch = one step of the recurrence  T_k = 2 A(T_{k-1}) - T_{k-2},  fbar += p_k T_k,
// A(x) = -x - c L(x)   (reference gcm_filters/filter.py:162-175, 192-206) fused with the Laplacian
// L of the grid type (reference gcm_filters/kernels.py:113-121, 172-187, 297-315, 351-372, 408-429,
// 469-487, 564-585), i.e. what the reference does in ~25-40 full-array numpy passes per step is ONE
// streaming pass here: 5 state words + (0 | 1 byte | 3 words) of coefficients per cell.
//
// Mapping to the hardware (bandwidth-bound, no MFMA):
//   * a wave (64 lanes) owns a 64*VEC-cell x-chunk of a row and MARCHES north over `rpw` rows, keeping
//     the previous/current/next rows of T_{k-1} (and the north-face coefficient of the previous row) in
//     registers: every T_{k-1} row is fetched once per strip (+2 halo rows), 16 B per lane per load
//     => 1 KiB fully coalesced per wave instruction;
//   * east/west neighbours come from the adjacent lane (ds_bpermute / DPP shuffles), only the two
//     edge lanes of a wave issue an extra (L1/L2-resident) scalar load; x is periodic so the first
//     and last chunk wrap;
//   * 4 waves per workgroup work on 4 different row strips of the same x-chunk.
#include "gcmf_internal.hpp"
#include "gcmf_recurrence.hpp"

#include <cfloat>
#include <type_traits>

namespace gcmf {

template <typename T> struct Lim;
template <> struct Lim<float> { static __device__ __forceinline__ float big() { return FLT_MAX; } };
template <> struct Lim<double> { static __device__ __forceinline__ double big() { return DBL_MAX; } };

// numpy.nan_to_num defaults: nan -> 0, +inf -> max, -inf -> -max   (kernels.py:175, 300, 353, 472, 566)
template <typename T> __device__ __forceinline__ T sanitize(T x) {
  if (x != x) return T(0);
  if (x > Lim<T>::big()) return Lim<T>::big();
  if (x < -Lim<T>::big()) return -Lim<T>::big();
  return x;
}

// VEC consecutive cells of one lane; 16-byte aligned chunks so the compiler emits dwordx4 / dwordx2 accesses
template <typename T, int VEC> struct alignas((sizeof(T) * VEC) > 16 ? 16 : (sizeof(T) * VEC)) Pack { T s[VEC]; };

template <typename T, int VEC> __device__ __forceinline__ void load_vec(T (&d)[VEC], const T *p) {
  const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(p);
#pragma unroll
  for (int k = 0; k < VEC; ++k) d[k] = v.s[k];
}
template <typename T, int VEC> __device__ __forceinline__ void store_vec(T *p, const T (&d)[VEC]) {
  Pack<T, VEC> v;
#pragma unroll
  for (int k = 0; k < VEC; ++k) v.s[k] = d[k];
  *reinterpret_cast<Pack<T, VEC> *>(p) = v;
}
// mirrored load: d[k] = row[nx-1-(i0+k)]
template <typename T, int VEC> __device__ __forceinline__ void load_vec_rev(T (&d)[VEC], const T *row, int nx, int i0) {
  T t[VEC];
  load_vec<T, VEC>(t, row + (nx - VEC - i0));
#pragma unroll
  for (int k = 0; k < VEC; ++k) d[k] = t[VEC - 1 - k];
}

template <typename T> __device__ __forceinline__ T shfl_up1(T v) { return __shfl_up(v, 1, 64); }
template <typename T> __device__ __forceinline__ T shfl_down1(T v) { return __shfl_down(v, 1, 64); }

template <typename T, typename FB> struct ScalarP {
  const T *t1;
  const T *t2;
  const FB *fb_in;
  T *t0;
  FB *fb_out;
  const T *cE, *cN, *ra;
  const uint8_t *mbits;
  const T *area;
  int nx, rows, row_lo, row_hi, rpw;
  int fb_lo;  // rows below this index do not touch fbar (ghost rows of the tripole band, see gcmf_api.hip)
  int ntx, ntiles, per_xcd;  // tile grid: ntx x-chunks per row group, ntiles total, tiles per XCD (0 = no remap)
  long long bstride;
  int south_wrap, north_wrap, fold, area_weighted;
  unsigned mode;
  double coef0, coef1, c;
};

template <typename T, typename FB, int KIND, int VEC>
__global__ __launch_bounds__(256) void k_scalar_step(const ScalarP<T, FB> P) {
  const int lane = threadIdx.x;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), so give XCD k
  // the k-th contiguous band of row groups; the j+-1 rows a workgroup re-reads were then fetched into the SAME
  // XCD's L2 by its neighbours instead of being pulled from HBM/MALL again by another XCD.
  int tile = blockIdx.x;
  if (P.per_xcd > 0) tile = (tile & 7) * P.per_xcd + (tile >> 3);
  if (tile >= P.ntiles) return;
  const int ty = tile / P.ntx, tx = tile - ty * P.ntx;
  const int strip = ty * blockDim.y + threadIdx.y;
  const int jb = P.row_lo + strip * P.rpw;
  if (jb >= P.row_hi) return;  // wave-uniform
  const int je = min(jb + P.rpw, P.row_hi);
  const int nx = P.nx;
  const int i0r = (tx * 64 + lane) * VEC;
  const bool active = i0r < nx;
  const int i0 = active ? i0r : 0;
  const long long boff = (long long)blockIdx.y * P.bstride;
  const T *t1 = P.t1 + boff;
  const int iw = (i0 == 0) ? nx - 1 : i0 - 1;
  const int ie = (i0 + VEC >= nx) ? 0 : i0 + VEC;
  const bool edge_w = (lane == 0);
  const bool edge_e = (lane == 63) || (i0r + VEC >= nx);
  const bool first = P.mode & GCMF_STEP_FIRST, last = P.mode & GCMF_STEP_LAST, lapl = P.mode & STEP_LAPL;
  const T c = (T)P.c;

  auto south_of = [&](int j) { return j > 0 ? j - 1 : (P.south_wrap ? P.rows - 1 : 0); };
  auto row_ptr = [&](const T *base, int j) { return base + (long long)j * nx; };

  T prev[VEC], cur[VEC], nxt[VEC];  // T_{k-1} rows j-1, j, j+1; `cur` raw, `prev`/`nxt` as the stencil sees them
  T cNs[VEC];                        // north-face coefficient of row j-1 (K_FLUX)
  load_vec<T, VEC>(prev, row_ptr(t1, south_of(jb)) + i0);
  load_vec<T, VEC>(cur, row_ptr(t1, jb) + i0);
  if (KIND != K_REG) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) prev[k] = sanitize(prev[k]);
  }
  if (KIND == K_FLUX) {
    if (jb > 0 || P.south_wrap) {
      load_vec<T, VEC>(cNs, row_ptr(P.cN, south_of(jb)) + i0);
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) cNs[k] = T(0);
    }
  }

  for (int j = jb; j < je; ++j) {
    const T *rowj = row_ptr(t1, j);
    // ---- northern row (wrap / tripole fold / plain) ----
    if (j < P.rows - 1) {
      load_vec<T, VEC>(nxt, row_ptr(t1, j + 1) + i0);
    } else if (P.fold) {
      load_vec_rev<T, VEC>(nxt, rowj, nx, i0);
    } else {
      load_vec<T, VEC>(nxt, row_ptr(t1, P.north_wrap ? 0 : j) + i0);
    }
    // ---- other per-row operands, issued before any use so the loads overlap ----
    const long long off = boff + (long long)j * nx + i0;
    T x2[VEC];
    FB fb[VEC];
    if (!first && !lapl) {
      load_vec<T, VEC>(x2, P.t2 + off);
      load_vec<FB, VEC>(fb, P.fb_in + off);
    }
    T cEv[VEC], cNv[VEC], rav[VEC], cEw = T(0);
    uint8_t mb[VEC];
    if (KIND == K_FLUX) {
      const long long coff = (long long)j * nx + i0;
      load_vec<T, VEC>(cEv, P.cE + coff);
      load_vec<T, VEC>(cNv, P.cN + coff);
      load_vec<T, VEC>(rav, P.ra + coff);
      cEw = shfl_up1(cEv[VEC - 1]);
      if (edge_w) cEw = P.cE[(long long)j * nx + iw];
    }
    if (KIND == K_MASK) load_vec<uint8_t, VEC>(mb, P.mbits + (long long)j * nx + i0);
    T ar[VEC];
    if (last && P.area_weighted) load_vec<T, VEC>(ar, P.area + (long long)j * nx + i0);

    // ---- stencil values ----
    T g[VEC], gn[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      g[k] = (KIND == K_REG) ? cur[k] : sanitize(cur[k]);
      gn[k] = (KIND == K_REG) ? nxt[k] : sanitize(nxt[k]);
    }
    T wv = shfl_up1(g[VEC - 1]);
    T ev = shfl_down1(g[0]);
    if (edge_w) { T t = rowj[iw]; wv = (KIND == K_REG) ? t : sanitize(t); }
    if (edge_e) { T t = rowj[ie]; ev = (KIND == K_REG) ? t : sanitize(t); }

    T t0v[VEC];
    FB fbo[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T gC = g[k];
      const T gW = (k == 0) ? wv : g[k > 0 ? k - 1 : 0];
      const T gE = (k == VEC - 1) ? ev : g[k < VEC - 1 ? k + 1 : k];
      const T gN = gn[k], gS = prev[k];
      T L;
      if (KIND == K_REG) {  // -4 f + E + W + N + S, reference evaluation order (kernels.py:115-121)
        L = T(-4) * gC + gE;
        L = L + gW;
        L = L + gN;
        L = L + gS;
      } else if (KIND == K_MASK) {  // m (-wf g + gE + gW + gN + gS), g = m nan_to_num(f) (kernels.py:175-186)
        const unsigned b = mb[k];
        const T mC = (b & 1u) ? gC : T(0);
        const T wf = (T)(b >> 5);  // wet-neighbour count, precomputed in bits 5-7
        L = -wf * mC + ((b & 2u) ? gE : T(0));
        L = L + ((b & 4u) ? gW : T(0));
        L = L + ((b & 8u) ? gN : T(0));
        L = L + ((b & 16u) ? gS : T(0));
        L = (b & 1u) ? L : T(0);
      } else {  // flux form: east/west/north/south face fluxes times 1/area (kernels.py:302-314, 571-584)
        const T cw = (k == 0) ? cEw : cEv[k > 0 ? k - 1 : 0];
        const T fe = (gE - gC) * cEv[k];
        const T fw = (gC - gW) * cw;
        const T fn = (gN - gC) * cNv[k];
        const T fs = (gC - gS) * cNs[k];
        L = ((fe - fw) + (fn - fs)) * rav[k];
      }
      const T x = cur[k];  // raw centre: NaNs survive in "-x" exactly as in the reference (filter.py:171-173)
      if (lapl) {
        t0v[k] = L;
      } else {
        constexpr bool FUSED = (KIND == K_FLUX);  // see gcmf_recurrence.hpp
        const T a = cheb_a<FUSED>(x, c, L);
        if (first) {
          t0v[k] = a;
          fbo[k] = cheb_acc_first<FUSED, T, FB>(P.coef0, P.coef1, x, a);
        } else {
          const T tk = cheb_t<FUSED>(a, x2[k]);
          t0v[k] = tk;
          fbo[k] = cheb_acc<FUSED, T, FB>(fb[k], P.coef0, tk);
        }
        if (last && P.area_weighted) fbo[k] = fbo[k] / (FB)ar[k];  // finalize (kernels.py:103-104)
      }
    }
    if (active) {
      if (!(last && !lapl)) store_vec<T, VEC>(P.t0 + off, t0v);
      if (!lapl && j >= P.fb_lo) store_vec<FB, VEC>(P.fb_out + off, fbo);
    }
    // ---- march north ----
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      prev[k] = g[k];
      cur[k] = nxt[k];
      if (KIND == K_FLUX) cNs[k] = cNv[k];
    }
  }
}

template <typename T, typename FB, int KIND, int VEC>
static int launch_k(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  const Geom &g = pl->g;
  ScalarP<T, FB> P;
  P.t1 = (const T *)a.t1[0];
  P.t2 = (const T *)a.t2[0];
  P.fb_in = (const FB *)a.fb_in[0];
  P.t0 = (T *)a.t0[0];
  P.fb_out = (FB *)a.fb_out[0];
  P.cE = (const T *)g.coef[0];
  P.cN = (const T *)g.coef[1];
  P.ra = (const T *)g.coef[2];
  P.mbits = g.mbits;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.row_lo = a.row_lo;
  P.row_hi = a.row_hi;
  P.fb_lo = a.fb_lo;
  P.rpw = a.rpw > 0 ? a.rpw : (pl->rows_per_wave > 0 ? pl->rows_per_wave : 2);  // measured on MI355X: 1-4 rows per wave within noise, 8+ slower
  P.bstride = (long long)g.rows * g.nx;
  P.south_wrap = g.south_wrap;
  P.north_wrap = g.north_wrap;
  P.fold = g.fold;
  P.area_weighted = g.area_weighted;
  P.mode = a.mode;
  P.coef0 = a.coef0;
  P.coef1 = a.coef1;
  P.c = a.c;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  const int nstrips = (nrows + P.rpw - 1) / P.rpw;
  dim3 block(64, 4, 1);
  P.ntx = (g.nx + 64 * VEC - 1) / (64 * VEC);
  const int nty = (nstrips + 3) / 4;
  P.ntiles = P.ntx * nty;
  P.per_xcd = (pl->xcd_remap && nty >= 16) ? (P.ntiles + 7) / 8 : 0;
  dim3 grid(P.per_xcd ? 8 * P.per_xcd : P.ntiles, (unsigned)a.nbatch, 1);
  hipLaunchKernelGGL((k_scalar_step<T, FB, KIND, VEC>), grid, block, 0, s, P);
  note_kernel(pl, std::string("gcmf::k_scalar_step<") + tyname<T>() + ", " + tyname<FB>() + ", " + std::to_string(KIND) + ", " +
                      std::to_string(VEC) + ">", 1);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T, typename FB, int KIND> static int launch_kv(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  constexpr int V = 16 / sizeof(T);
  const Geom &g = pl->g;
  bool vec_ok = (g.nx % V == 0) && g.nx >= V;
  const void *ptrs[] = {a.t1[0], a.t2[0], a.fb_in[0], a.t0[0], a.fb_out[0], g.coef[0], g.coef[1], g.coef[2], g.area};
  for (const void *p : ptrs) vec_ok = vec_ok && aligned16(p);
  if (KIND == K_MASK) vec_ok = vec_ok && ((reinterpret_cast<uintptr_t>(g.mbits) & (V - 1)) == 0);
  if (vec_ok) return launch_k<T, FB, KIND, V>(pl, a, s);
  return launch_k<T, FB, KIND, 1>(pl, a, s);
}

template <typename T, typename FB> static int launch_kind(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  switch (pl->kind) {
    case K_REG: return launch_kv<T, FB, K_REG>(pl, a, s);
    case K_MASK: return launch_kv<T, FB, K_MASK>(pl, a, s);
    case K_FLUX: return launch_kv<T, FB, K_FLUX>(pl, a, s);
  }
  set_error("launch_scalar_step: plan is not a scalar kind");
  return GCMF_ERR_INVALID_ARG;
}

int launch_scalar_step(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_kind<double, double>(pl, a, s);
  if (a.fb_is_f32) return launch_kind<float, float>(pl, a, s);
  return launch_kind<float, double>(pl, a, s);
}

// ---- prepare: T_0 = field * area (AreaWeightedMixin.prepare, kernels.py:100-101) or a plain copy ----
template <typename T>
__global__ __launch_bounds__(256) void k_prepare(const T *in, T *out, const T *area, int nx, long long bstride,
                                                 int row_lo, int row_hi) {
  const long long n = (long long)(row_hi - row_lo) * nx;
  const long long boff = (long long)blockIdx.y * bstride + (long long)row_lo * nx;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const T a = area ? area[(long long)row_lo * nx + q] : T(1);
    out[boff + q] = area ? in[boff + q] * a : in[boff + q];
  }
}

int launch_prepare(gcmf_plan *pl, const void *const *in, void *const *out, int64_t nbatch, int row_lo, int row_hi,
                   hipStream_t s) {
  const Geom &g = pl->g;
  if (row_hi <= row_lo || nbatch <= 0) return GCMF_OK;
  const long long n = (long long)(row_hi - row_lo) * g.nx;
  const long long bstride = (long long)g.rows * g.nx;
  dim3 block(256), grid((unsigned)std::min<long long>((n + 255) / 256, 4096), (unsigned)nbatch);
  for (int cpt = 0; cpt < pl->ncomp; ++cpt) {
    if (pl->d.dtype == GCMF_F64)
      hipLaunchKernelGGL(k_prepare<double>, grid, block, 0, s, (const double *)in[cpt], (double *)out[cpt],
                         (const double *)(g.area_weighted ? g.area : nullptr), g.nx, bstride, row_lo, row_hi);
    else
      hipLaunchKernelGGL(k_prepare<float>, grid, block, 0, s, (const float *)in[cpt], (float *)out[cpt],
                         (const float *)(g.area_weighted ? g.area : nullptr), g.nx, bstride, row_lo, row_hi);
    GCMF_HIP(hipGetLastError());
  }
  return GCMF_OK;
}

}  // namespace gcmf
