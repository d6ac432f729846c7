// Streaming B-grid vector Chebyshev step (VECTOR_B_GRID, reference gcm_filters/kernels.py:740-837 with the ten
// stencil weights hoisted to plan time, fused with the recurrence of filter.py:225-283).
//
// Wave-march mapping (see gcmf_cgrid_stream.hip): a wave owns 64*VEC contiguous cells, keeps the sanitised rows
// r-2, r-1, r of u and v in registers, gets east/west neighbours by DPP and writes row r-1 in the iteration that takes
// delivery of row r.  The eight coefficient planes, T_{k-2} and fbar are centre-only and travel one row late.
// Summation order is the reference's, term by term, so the result is bit-identical to numpy.
#include "gcmf_multi_common.hpp"

namespace gcmf {

template <typename T, typename FB> struct BStreamP {
  const T *u0, *v0, *u2, *v2;
  const FB *fu_in, *fv_in;
  T *uo, *vo;
  FB *fu_out, *fv_out;
  const T *coef[8];  // cc, DUN, DUS, DUE, DUW, DMC, DMN, DME
  int nx, rows, out_lo, out_hi, H, nwx, ngroups, nlev, wrap;
  unsigned mode;
  long long bstride;
  double coef0, coef1, c;
};

template <typename T> __device__ __forceinline__ T bsan(T x) {
  const bool isn = (x != x);
  const bool big = (mabs(x) > MLim<T>::big());
  const T clamped = big ? (x > T(0) ? MLim<T>::big() : -MLim<T>::big()) : x;
  return isn ? T(0) : clamped;
}

template <typename T, typename FB> __global__ __launch_bounds__(256, 2) void k_bgrid_stream(const BStreamP<T, FB> P) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC, M = VEC, WI = W - 2 * M;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int group = wid / P.nlev, lev = wid % P.nlev;
  if (group >= P.ngroups) return;
  const int wx = group % P.nwx, st = group / P.nwx;
  const int nx = P.nx, rows = P.rows;
  const int a = P.out_lo + st * P.H;
  const int b = min(a + P.H, P.out_hi);
  const long long boff = (long long)lev * P.bstride;
  const int pos = wx * WI - M + lane * VEC;
  int col = pos % nx;
  if (col < 0) col += nx;
  const bool keep = (lane * VEC >= M) && (lane * VEC < W - M) && (pos < nx);
  const T c = (T)P.c;
  const bool first = P.mode & GCMF_STEP_FIRST, last = P.mode & GCMF_STEP_LAST, lapl = P.mode & STEP_LAPL;
  const bool need_prev = !first && !lapl;

  struct Row {
    T u[VEC], v[VEC];      // row r
    T k[8][VEC];           // coefficients of row r-1
    T u2[VEC], v2[VEC];    // row r-1
    FB fu[VEC], fv[VEC];   // row r-1
  };
  auto row_index = [&](int r) {
    if (P.wrap) return r < 0 ? r + rows : (r >= rows ? r - rows : r);
    return r < 0 ? 0 : (r >= rows ? rows - 1 : r);
  };
  auto load_row = [&](Row &x, int r) {
    const long long ro = (long long)row_index(r) * nx + col;
    const long long rc = (long long)row_index(r - 1) * nx + col;
    mload<T, VEC>(x.u, P.u0 + boff + ro);
    mload<T, VEC>(x.v, P.v0 + boff + ro);
#pragma unroll
    for (int q = 0; q < 8; ++q) mload<T, VEC>(x.k[q], P.coef[q] + rc);
    if (need_prev) {
      mload<T, VEC>(x.u2, P.u2 + boff + rc);
      mload<T, VEC>(x.v2, P.v2 + boff + rc);
      mload<FB, VEC>(x.fu, P.fu_in + boff + rc);
      mload<FB, VEC>(x.fv, P.fv_in + boff + rc);
    }
  };

  // sanitised rows r-2 (S), r-1 (C) and raw row r-1 of both components
  T uS[VEC], uC[VEC], vS[VEC], vC[VEC], uR[VEC], vR[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) uS[k] = uC[k] = vS[k] = vC[k] = uR[k] = vR[k] = T(0);

  auto update = [&](T x, T L, T x2, FB fbin, T &tk, FB &fb) {
    const T av = -x - c * L;
    if (first) {
      tk = av;
      if (std::is_same<FB, T>::value) fb = (FB)((T)P.coef0 * x + (T)P.coef1 * av);
      else fb = (FB)(P.coef0 * (double)x + P.coef1 * (double)av);
    } else {
      tk = T(2) * av - x2;
      if (std::is_same<FB, T>::value) fb = fbin + (FB)((T)P.coef0 * tk);
      else fb = fbin + (FB)(P.coef0 * (double)tk);
    }
  };

  auto step = [&](const Row &x, int r) {
    T uN[VEC], vN[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { uN[k] = bsan(x.u[k]); vN[k] = bsan(x.v[k]); }
    const T uw_ = from_lower_lane0(uC[VEC - 1]), ue_ = from_upper_lane0(uC[0]);
    const T vw_ = from_lower_lane0(vC[VEC - 1]), ve_ = from_upper_lane0(vC[0]);
    const int j = r - 1;
    T tu[VEC], tv[VEC];
    FB fu[VEC], fv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T uc = uC[k], vc = vC[k];
      const T uw = (k == 0) ? uw_ : uC[k > 0 ? k - 1 : 0], ue = (k == VEC - 1) ? ue_ : uC[k < VEC - 1 ? k + 1 : k];
      const T vw = (k == 0) ? vw_ : vC[k > 0 ? k - 1 : 0], ve = (k == VEC - 1) ? ve_ : vC[k < VEC - 1 ? k + 1 : k];
      const T cc = x.k[0][k], dun = x.k[1][k], dus = x.k[2][k], due = x.k[3][k], duw = x.k[4][k];
      const T dmc = x.k[5][k], dmn = x.k[6][k], dme = x.k[7][k];
      const T dms = -dmn, dmw = -dme;
      // reference summation order (kernels.py:811-835)
      T lu = cc * uc + dun * uN[k];
      lu = lu + dus * uS[k]; lu = lu + due * ue; lu = lu + duw * uw; lu = lu + dmc * vc;
      lu = lu + dmn * vN[k]; lu = lu + dms * vS[k]; lu = lu + dme * ve; lu = lu + dmw * vw;
      T lv = cc * vc + dun * vN[k];
      lv = lv + dus * vS[k]; lv = lv + due * ve; lv = lv + duw * vw; lv = lv + dmc * uc;
      lv = lv + dmn * uN[k]; lv = lv + dms * uS[k]; lv = lv + dme * ue; lv = lv + dmw * uw;
      if (lapl) {
        tu[k] = lu;
        tv[k] = lv;
      } else {
        update(uR[k], lu, x.u2[k], x.fu[k], tu[k], fu[k]);
        update(vR[k], lv, x.v2[k], x.fv[k], tv[k], fv[k]);
      }
    }
    if (keep && j >= a && j < b) {
      const long long off = boff + (long long)j * nx + col;
      if (!(last && !lapl)) {
        mstore<T, VEC>(P.uo + off, tu);
        mstore<T, VEC>(P.vo + off, tv);
      }
      if (!lapl) {
        mstore<FB, VEC>(P.fu_out + off, fu);
        mstore<FB, VEC>(P.fv_out + off, fv);
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      uS[k] = uC[k]; uC[k] = uN[k]; uR[k] = x.u[k];
      vS[k] = vC[k]; vC[k] = vN[k]; vR[k] = x.v[k];
    }
  };

  const int r_begin = a - 1, r_end = b + 1;  // rows delivered: [a-1, b]
  Row q0, q1;
  load_row(q0, r_begin);
  load_row(q1, min(r_begin + 1, r_end - 1));
#define GCMF_BSLOT(Q, dd)                                  \
  if (r + (dd) < r_end) {                                  \
    step(Q, r + (dd));                                     \
    load_row(Q, min(r + (dd) + 2, r_end - 1));             \
  }
  for (int r = r_begin; r < r_end; r += 2) {
    GCMF_BSLOT(q0, 0)
    GCMF_BSLOT(q1, 1)
  }
#undef GCMF_BSLOT
}

static bool bal16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T, typename FB> static int launch_bs(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC, WI = W - 2 * VEC;
  const Geom &g = pl->g;
  BStreamP<T, FB> P;
  P.u0 = (const T *)a.t1[0];
  P.v0 = (const T *)a.t1[1];
  P.u2 = (const T *)a.t2[0];
  P.v2 = (const T *)a.t2[1];
  P.fu_in = (const FB *)a.fb_in[0];
  P.fv_in = (const FB *)a.fb_in[1];
  P.uo = (T *)a.t0[0];
  P.vo = (T *)a.t0[1];
  P.fu_out = (FB *)a.fb_out[0];
  P.fv_out = (FB *)a.fb_out[1];
  for (int k = 0; k < 8; ++k) P.coef[k] = (const T *)g.coef[k];
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  P.nlev = (int)a.nbatch;
  long long want = 2048 / ((long long)P.nwx * a.nbatch);
  if (want < 1) want = 1;
  int H = (int)((nrows + want - 1) / want);
  if (H > 48) H = 48;
  if (H < 16) H = 16;
  if (H > nrows) H = nrows;
  P.H = H;
  P.ngroups = P.nwx * ((nrows + H - 1) / H);
  P.wrap = g.south_wrap && g.north_wrap;
  P.mode = a.mode;
  P.bstride = (long long)g.rows * g.nx;
  P.coef0 = a.coef0;
  P.coef1 = a.coef1;
  P.c = a.c;
  const long long nwaves = (long long)P.ngroups * P.nlev;
  dim3 block(256), grid((unsigned)((nwaves + 3) / 4));
  hipLaunchKernelGGL((k_bgrid_stream<T, FB>), grid, block, 0, s, P);
  note_kernel(pl, std::string("gcmf::k_bgrid_stream<") + tyname<T>() + ", " + tyname<FB>() + ">", 1);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

bool bgrid_stream_supported(const gcmf_plan *pl, const StepArgs &a) {
  if (pl->kind != K_BGRID) return false;
  const int vec = pl->d.dtype == GCMF_F64 ? 2 : 4;
  if (pl->g.nx % vec || pl->g.nx < vec || pl->g.rows < 3) return false;
  for (int k = 0; k < 2; ++k)
    if (!bal16(a.t1[k]) || !bal16(a.t2[k]) || !bal16(a.fb_in[k]) || !bal16(a.t0[k]) || !bal16(a.fb_out[k])) return false;
  for (int k = 0; k < 8; ++k)
    if (!bal16(pl->g.coef[k])) return false;
  return true;
}

int launch_bgrid_stream(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_bs<double, double>(pl, a, s);
  if (a.fb_is_f32) return launch_bs<float, float>(pl, a, s);
  return launch_bs<float, double>(pl, a, s);
}

}  // namespace gcmf
