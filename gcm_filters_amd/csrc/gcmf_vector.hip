// Vector (u, v) Chebyshev-step kernels for gfx950: VECTOR_C_GRID and VECTOR_B_GRID.
//
// Same fused recurrence step as gcmf_scalar.hip, on the coupled pair (reference filter.py:225-283), with
//   * C-grid viscous operator of Griffies & Hallberg (reference kernels.py:647-696): the stress-tensor
//     components are evaluated ONCE per cell for a (16+1) x (64+1) tile, kept in LDS (pre-scaled by the
//     two metric factors each consumer wants), and differenced from LDS -- the reference's ~58 full-array
//     passes become one pass over 4 state + 14 coefficient planes;
//   * POP B-grid operator (reference kernels.py:740-837) with the ten stencil weights hoisted to plan
//     time (8 planes; the reference rebuilds ~20 temporaries on every call).
#include "gcmf_internal.hpp"

#include <algorithm>
#include <cfloat>
#include <type_traits>

namespace gcmf {

template <typename T> __device__ __forceinline__ T vbig();
template <> __device__ __forceinline__ float vbig<float>() { return FLT_MAX; }
template <> __device__ __forceinline__ double vbig<double>() { return DBL_MAX; }
template <typename T> __device__ __forceinline__ T vsan(T x) {  // numpy.nan_to_num (kernels.py:650-651, 743-744)
  if (x != x) return T(0);
  if (x > vbig<T>()) return vbig<T>();
  if (x < -vbig<T>()) return -vbig<T>();
  return x;
}

template <typename T, typename FB> struct VecP {
  const T *t1[2];
  const T *t2[2];
  const FB *fb_in[2];
  T *t0[2];
  FB *fb_out[2];
  const T *coef[MAX_COEF];
  int nx, rows, row_lo, row_hi;
  long long bstride;
  int south_wrap, north_wrap;
  unsigned mode;
  double coef0, coef1, c;
};

// recurrence update of one component at one cell (filter.py:259-283)
template <typename T, typename FB>
__device__ __forceinline__ void cheb_update(const VecP<T, FB> &P, int comp, long long off, T x, T L) {
  if (P.mode & STEP_LAPL) {
    P.t0[comp][off] = L;
    return;
  }
  const T a = -x - (T)P.c * L;
  FB fb;
  T tk;
  if (P.mode & GCMF_STEP_FIRST) {
    tk = a;
    if (std::is_same<FB, T>::value) fb = (FB)((T)P.coef0 * x + (T)P.coef1 * a);
    else fb = (FB)(P.coef0 * (double)x + P.coef1 * (double)a);
  } else {
    tk = T(2) * a - P.t2[comp][off];
    if (std::is_same<FB, T>::value) fb = P.fb_in[comp][off] + (FB)((T)P.coef0 * tk);
    else fb = P.fb_in[comp][off] + (FB)(P.coef0 * (double)tk);
  }
  if (!(P.mode & GCMF_STEP_LAST)) P.t0[comp][off] = tk;
  P.fb_out[comp][off] = fb;
}

// same update with the centre-only operands (T_{k-2}, fbar) already in registers
template <typename T, typename FB>
__device__ __forceinline__ void cheb_update_pre(const VecP<T, FB> &P, int comp, long long off, T x, T L, T x2, FB fbin) {
  if (P.mode & STEP_LAPL) {
    P.t0[comp][off] = L;
    return;
  }
  const T a = -x - (T)P.c * L;
  FB fb;
  T tk;
  if (P.mode & GCMF_STEP_FIRST) {
    tk = a;
    if (std::is_same<FB, T>::value) fb = (FB)((T)P.coef0 * x + (T)P.coef1 * a);
    else fb = (FB)(P.coef0 * (double)x + P.coef1 * (double)a);
  } else {
    tk = T(2) * a - x2;
    if (std::is_same<FB, T>::value) fb = fbin + (FB)((T)P.coef0 * tk);
    else fb = fbin + (FB)(P.coef0 * (double)tk);
  }
  if (!(P.mode & GCMF_STEP_LAST)) P.t0[comp][off] = tk;
  P.fb_out[comp][off] = fb;
}

// periodic column index.  Tile points far outside a narrow grid (nx smaller than the tile) are never used, but
// their loads must still stay inside the plane: clamp after the single wrap.
__device__ __forceinline__ int wrapx(int i, int nx) {
  i = i < 0 ? i + nx : (i >= nx ? i - nx : i);
  return i < 0 ? 0 : (i >= nx ? nx - 1 : i);
}

// row index inside the slab allocation: periodic wrap for a single slab, clamp otherwise (clamped rows are
// only ever read for tile points whose results are not used)
__device__ __forceinline__ int slab_row(int j, int rows, int south_wrap, int north_wrap) {
  if (j < 0) j = south_wrap ? j + rows : 0;
  else if (j >= rows) j = north_wrap ? j - rows : rows - 1;
  return j < 0 ? 0 : (j >= rows ? rows - 1 : j);  // grids shorter than a tile: keep unused tile points in bounds
}

// ---------------------------------------------------------------------------------------------------
// C-grid
// ---------------------------------------------------------------------------------------------------
constexpr int CT_I = 64, CT_J = 16, CT_LD = CT_I + 1, CT_PTS = (CT_J + 1) * CT_LD;

// Workgroup order for batched fields: workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8).
// Tile T is pinned to XCD T % 8 and its `nlev` batch entries (vertical levels) occupy CONSECUTIVE slots of that
// XCD, so the tile's 14 coefficient planes are fetched from HBM once into that XCD's L2 and hit there for the
// other levels: coefficient traffic per cell.level drops from 14w to ~14w / nlev without any register cost.
template <typename T, typename FB>
__global__ __launch_bounds__(256) void k_cgrid_step(const VecP<T, FB> P, int ntx, int ntiles, int nlev) {
  __shared__ T sP[CT_PTS], sQ[CT_PTS], sR[CT_PTS], sS[CT_PTS];
  const int nx = P.nx;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / nlev) * 8 + xcd, lev = slot % nlev;
  if (tile >= ntiles) return;
  const int tyy = tile / ntx, txx = tile - tyy * ntx;
  const int I0 = txx * CT_I, J0 = P.row_lo + tyy * CT_J;
  const long long boff = (long long)lev * P.bstride;
  const T *u = P.t1[0] + boff, *v = P.t1[1] + boff;
  const T *r_dyCu = P.coef[0], *r_dxCu = P.coef[1], *r_dxCv = P.coef[2], *r_dyCv = P.coef[3];
  const T *a1 = P.coef[4], *a2 = P.coef[5], *rh = P.coef[6], *b1 = P.coef[7], *b2 = P.coef[8], *rq = P.coef[9];
  const int tid = threadIdx.y * 64 + threadIdx.x;

  // operands of phase 2 (this thread's 4 output cells): issued now so that they are in flight during phase 1
  const int bcol = threadIdx.x;
  const bool col_ok = (I0 + bcol) < nx;
  long long cc2[CT_J / 4];
  bool ok2[CT_J / 4];
  T k10[CT_J / 4], k11[CT_J / 4], k12[CT_J / 4], k13[CT_J / 4], xu[CT_J / 4], xv[CT_J / 4], x2u[CT_J / 4], x2v[CT_J / 4];
  FB fbu[CT_J / 4], fbv[CT_J / 4];
  const bool need_prev = !(P.mode & (GCMF_STEP_FIRST | STEP_LAPL));
#pragma unroll
  for (int q = 0; q < CT_J / 4; ++q) {
    const int j = J0 + threadIdx.y + 4 * q;
    ok2[q] = col_ok && j < P.row_hi;
    cc2[q] = ok2[q] ? (long long)j * nx + (I0 + bcol) : 0;
    k10[q] = P.coef[10][cc2[q]]; k11[q] = P.coef[11][cc2[q]]; k12[q] = P.coef[12][cc2[q]]; k13[q] = P.coef[13][cc2[q]];
    xu[q] = u[cc2[q]]; xv[q] = v[cc2[q]];
    x2u[q] = T(0); x2v[q] = T(0); fbu[q] = FB(0); fbv[q] = FB(0);
    if (need_prev) {
      x2u[q] = P.t2[0][boff + cc2[q]]; x2v[q] = P.t2[1][boff + cc2[q]];
      fbu[q] = P.fb_in[0][boff + cc2[q]]; fbv[q] = P.fb_in[1][boff + cc2[q]];
    }
  }

  // phase 1: stresses.  sP[a][b] = dy2h*str_xx at (J0+a, I0+b);  sR[a][b] = dx2q*str_xy at (J0-1+a, I0-1+b)
  for (int idx = tid; idx < CT_PTS; idx += 256) {
    const int a = idx / CT_LD, b = idx - a * CT_LD;
    {
      const int j = slab_row(J0 + a, P.rows, P.south_wrap, P.north_wrap);
      const int js = slab_row(J0 + a - 1, P.rows, P.south_wrap, P.north_wrap);
      const int i = wrapx(I0 + b, nx), iw = wrapx(I0 + b - 1, nx);
      const long long c = (long long)j * nx + i, cw = (long long)j * nx + iw, cs = (long long)js * nx + i;
      const T ut = vsan(u[c]) * r_dyCu[c], utw = vsan(u[cw]) * r_dyCu[cw];
      const T vt = vsan(v[c]) * r_dxCv[c], vts = vsan(v[cs]) * r_dxCv[cs];
      const T p = a1[c] * (ut - utw) - a2[c] * (vt - vts);
      sP[idx] = p;
      sQ[idx] = rh[c] * p;
    }
    {
      const int j = slab_row(J0 - 1 + a, P.rows, P.south_wrap, P.north_wrap);
      const int jn = slab_row(J0 + a, P.rows, P.south_wrap, P.north_wrap);
      const int i = wrapx(I0 - 1 + b, nx), ie = wrapx(I0 + b, nx);
      const long long c = (long long)j * nx + i, ce = (long long)j * nx + ie, cn = (long long)jn * nx + i;
      const T vh = vsan(v[c]) * r_dyCv[c], vhe = vsan(v[ce]) * r_dyCv[ce];
      const T uh = vsan(u[c]) * r_dxCu[c], uhn = vsan(u[cn]) * r_dxCu[cn];
      const T r = b1[c] * (vhe - vh) + b2[c] * (uhn - uh);
      sR[idx] = r;
      sS[idx] = rq[c] * r;
    }
  }
  __syncthreads();

  // phase 2: divergence of the stresses + recurrence update (operands were fetched before phase 1)
  if (!col_ok) return;
#pragma unroll
  for (int q = 0; q < CT_J / 4; ++q) {
    if (!ok2[q]) continue;
    const int a = threadIdx.y + 4 * q;
    const int l00 = a * CT_LD + bcol;  // [a][b]
    const T lu = k10[q] * (sP[l00] - sP[l00 + 1]) + k11[q] * (sR[l00 + 1] - sR[l00 + CT_LD + 1]);
    const T lv = k12[q] * (sS[l00 + CT_LD] - sS[l00 + CT_LD + 1]) - k13[q] * (sQ[l00] - sQ[l00 + CT_LD]);
    cheb_update_pre<T, FB>(P, 0, boff + cc2[q], xu[q], lu, x2u[q], fbu[q]);
    cheb_update_pre<T, FB>(P, 1, boff + cc2[q], xv[q], lv, x2v[q], fbv[q]);
  }
}

// ---------------------------------------------------------------------------------------------------
// B-grid: two coupled 5-point stencils sharing 8 coefficient planes
// ---------------------------------------------------------------------------------------------------
constexpr int BG_ROWS = 4;  // rows per thread

template <typename T, typename FB> __global__ __launch_bounds__(256) void k_bgrid_step(const VecP<T, FB> P) {
  const int nx = P.nx;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nx) return;
  const int jb = P.row_lo + (blockIdx.y * 4 + threadIdx.y) * BG_ROWS;
  const long long boff = (long long)blockIdx.z * P.bstride;
  const T *u = P.t1[0] + boff, *v = P.t1[1] + boff;
  const int ie = wrapx(i + 1, nx), iw = wrapx(i - 1, nx);
  for (int j = jb; j < jb + BG_ROWS && j < P.row_hi; ++j) {
    const int jn = slab_row(j + 1, P.rows, P.south_wrap, P.north_wrap);
    const int js = slab_row(j - 1, P.rows, P.south_wrap, P.north_wrap);
    const long long c = (long long)j * nx + i;
    const long long rn = (long long)jn * nx + i, rs = (long long)js * nx + i;
    const long long re = (long long)j * nx + ie, rw = (long long)j * nx + iw;
    const T uc = vsan(u[c]), un = vsan(u[rn]), us = vsan(u[rs]), ue = vsan(u[re]), uw = vsan(u[rw]);
    const T vc = vsan(v[c]), vn = vsan(v[rn]), vs = vsan(v[rs]), ve = vsan(v[re]), vw = vsan(v[rw]);
    const T cc = P.coef[0][c], dun = P.coef[1][c], dus = P.coef[2][c], due = P.coef[3][c], duw = P.coef[4][c];
    const T dmc = P.coef[5][c], dmn = P.coef[6][c], dme = P.coef[7][c];
    const T dms = -dmn, dmw = -dme;
    // reference summation order (kernels.py:811-835)
    T lu = cc * uc + dun * un;
    lu = lu + dus * us; lu = lu + due * ue; lu = lu + duw * uw; lu = lu + dmc * vc;
    lu = lu + dmn * vn; lu = lu + dms * vs; lu = lu + dme * ve; lu = lu + dmw * vw;
    T lv = cc * vc + dun * vn;
    lv = lv + dus * vs; lv = lv + due * ve; lv = lv + duw * vw; lv = lv + dmc * uc;
    lv = lv + dmn * un; lv = lv + dms * us; lv = lv + dme * ue; lv = lv + dmw * uw;
    cheb_update<T, FB>(P, 0, boff + c, u[c], lu);
    cheb_update<T, FB>(P, 1, boff + c, v[c], lv);
  }
}

template <typename T, typename FB> static int launch_vec(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  const Geom &g = pl->g;
  VecP<T, FB> P;
  for (int k = 0; k < 2; ++k) {
    P.t1[k] = (const T *)a.t1[k];
    P.t2[k] = (const T *)a.t2[k];
    P.fb_in[k] = (const FB *)a.fb_in[k];
    P.t0[k] = (T *)a.t0[k];
    P.fb_out[k] = (FB *)a.fb_out[k];
  }
  for (int k = 0; k < MAX_COEF; ++k) P.coef[k] = (const T *)g.coef[k];
  P.nx = g.nx;
  P.rows = g.rows;
  P.row_lo = a.row_lo;
  P.row_hi = a.row_hi;
  P.bstride = (long long)g.rows * g.nx;
  P.south_wrap = g.south_wrap;
  P.north_wrap = g.north_wrap;
  P.mode = a.mode;
  P.coef0 = a.coef0;
  P.coef1 = a.coef1;
  P.c = a.c;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  dim3 block(64, 4, 1);
  if (pl->kind == K_CGRID && !pl->cgrid_tile && cgrid_stream_supported(pl, a)) return launch_cgrid_stream(pl, a, s);
  if (pl->kind == K_BGRID && !pl->cgrid_tile && bgrid_stream_supported(pl, a)) return launch_bgrid_stream(pl, a, s);
  if (pl->kind == K_CGRID) {
    const int ntx = (g.nx + CT_I - 1) / CT_I, nty = (nrows + CT_J - 1) / CT_J;
    const int ntiles = ntx * nty;
    const long long nblocks = (long long)((ntiles + 7) / 8) * 8 * a.nbatch;
    dim3 grid((unsigned)nblocks, 1, 1);
    hipLaunchKernelGGL((k_cgrid_step<T, FB>), grid, block, 0, s, P, ntx, ntiles, (int)a.nbatch);
  } else {
    dim3 grid((g.nx + 63) / 64, (nrows + 4 * BG_ROWS - 1) / (4 * BG_ROWS), (unsigned)a.nbatch);
    hipLaunchKernelGGL((k_bgrid_step<T, FB>), grid, block, 0, s, P);
  }
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

int launch_vector_step(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  if (pl->kind != K_CGRID && pl->kind != K_BGRID) {
    set_error("launch_vector_step: plan is not a vector kind");
    return GCMF_ERR_INVALID_ARG;
  }
  if (pl->d.dtype == GCMF_F64) return launch_vec<double, double>(pl, a, s);
  if (a.fb_is_f32) return launch_vec<float, float>(pl, a, s);
  return launch_vec<float, double>(pl, a, s);
}

}  // namespace gcmf
