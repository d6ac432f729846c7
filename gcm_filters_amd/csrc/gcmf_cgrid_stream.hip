// Streaming C-grid vector Chebyshev step (VECTOR_C_GRID, reference gcm_filters/kernels.py:647-696 fused with the
// recurrence of filter.py:225-283).
//
// Same hardware mapping as the scalar kernels: a wave owns a window of 64*VEC contiguous cells in x and marches north;
// every plane is read exactly once per step with 16-byte accesses, x-neighbours come from the adjacent lane by DPP
// moves, y-neighbours from the previous iteration's registers.  No LDS, no barriers.
//
// In the iteration that takes delivery of row r (u, v and the seven "row r" coefficient planes), together with the
// centre-only operands of row r-1 (T_{k-2}, fbar and the seven coefficient planes that are only needed there):
//     P_r     = dy2h*str_xx(r)   from u~(r), W u~(r), v~(r), v~(r-1)                 Q_r = (dx2h/dy2h) P_r
//     R_{r-1} = dx2q*str_xy(r-1) from v^(r-1), E v^(r-1), u^(r), u^(r-1)             S_{r-1} = (dy2q/dx2q) R_{r-1}
//     L_u(r-1) = cu1 (P_{r-1} - E P_{r-1}) + cu2 (R_{r-2} - R_{r-1})
//     L_v(r-1) = cv1 (W S_{r-1} - S_{r-1}) - cv2 (Q_{r-1} - Q_r)
// and row r-1 of T_k / fbar is written.  The outermost cell of a window goes stale (one DPP hop per stage), so windows
// overlap by VEC cells per side; strips of rows overlap by one row per side.
//
// Batched fields (vertical levels) share the 2-D coefficient planes: the waves of all levels of one (window, strip)
// are placed on the SAME XCD in consecutive slots and march in step, so a coefficient row is fetched from HBM by the
// first of them and served from that XCD's L2 to the others (14w / nlev instead of 14w bytes per cell.level).
#include "gcmf_multi_common.hpp"

#include <cstdlib>

namespace gcmf {

template <typename T, typename FB> struct CStreamP {
  const T *u0, *v0;        // T_{k-1}
  const T *u2, *v2;        // T_{k-2}
  const FB *fu_in, *fv_in;
  T *uo, *vo;
  FB *fu_out, *fv_out;
  const T *coef[MAX_COEF];
  int nx, rows, out_lo, out_hi;
  int H, nwx, ngroups, nlev, nlev4;
  int wrap, lockstep;
  unsigned mode;
  long long bstride;
  double coef0, coef1, c;
};

template <typename T> __device__ __forceinline__ T csan(T x) {  // numpy.nan_to_num as selects
  const bool isn = (x != x);
  const bool big = (mabs(x) > MLim<T>::big());
  const T clamped = big ? (x > T(0) ? MLim<T>::big() : -MLim<T>::big()) : x;
  return isn ? T(0) : clamped;
}

// SHARED: the 4 waves of a workgroup are 4 levels of one (window, strip) group marching in lock-step; each wave
// fetches a quarter of the 14 coefficient rows and they are exchanged through LDS (double-buffered, one barrier
// per row), so a coefficient row crosses the memory system once per workgroup instead of once per level.
template <typename T, typename FB, int D, bool SHARED>
__global__ __launch_bounds__(256, 2) void k_cgrid_stream(const CStreamP<T, FB> P) {
  __shared__ MPack<T, 16 / sizeof(T)> s_coef[SHARED ? 2 : 1][SHARED ? 14 : 1][SHARED ? 64 : 1];
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC;
  constexpr int M = VEC;
  constexpr int WI = W - 2 * M;

  const int lane = threadIdx.x & 63;
  // wave id -> (group = window x strip, level) with all levels of a group on one XCD, consecutive slots
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int blk = blockIdx.x;
  const int xcd = blk & 7, slot = (blk >> 3) * 4 + (threadIdx.x >> 6);  // wave slot inside this XCD's sequence
  // every group owns nlev4 = nlev rounded up to a multiple of 4 consecutive wave slots, so that the 4 waves of a
  // workgroup are always 4 levels of ONE group (they march in lock-step and share coefficient rows through L1/L2);
  // the padding waves shadow the last level without storing
  const int group = (slot / P.nlev4) * 8 + xcd;
  int lev = slot % P.nlev4;
  (void)wid;
  if (group >= P.ngroups) return;  // whole workgroups exit together (nlev4 is a multiple of the 4 waves)
  const bool shadow = lev >= P.nlev;
  if (shadow) lev = P.nlev - 1;
  const int wx = group % P.nwx, st = group / P.nwx;
  const int nx = P.nx, rows = P.rows;
  const int a = P.out_lo + st * P.H;
  const int b = min(a + P.H, P.out_hi);
  const long long boff = (long long)lev * P.bstride;
  const int pos = wx * WI - M + lane * VEC;
  int col = pos % nx;
  if (col < 0) col += nx;
  const bool keep = (lane * VEC >= M) && (lane * VEC < W - M) && (pos < nx) && !shadow;
  const T c = (T)P.c;
  const bool first = P.mode & GCMF_STEP_FIRST, last = P.mode & GCMF_STEP_LAST, lapl = P.mode & STEP_LAPL;
  const bool need_prev = !first && !lapl;

  struct Row {
    T u[VEC], v[VEC];                                             // row r
    T rdyCu[VEC], rdxCu[VEC], rdxCv[VEC], rdyCv[VEC], a1[VEC], a2[VEC], rh[VEC];  // row r
    T b1[VEC], b2[VEC], rq[VEC], cu1[VEC], cu2[VEC], cv1[VEC], cv2[VEC];         // row r-1
    T u2[VEC], v2[VEC];                                           // row r-1
    FB fu[VEC], fv[VEC];                                          // row r-1
    T share[4][VEC];                                              // SHARED: planes wv, wv+4, wv+8, wv+12 of this row
  };
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this wave's coefficient planes, resolved once: indexing the kernel argument inside the row loop costs a
  // dependent memory load plus a full vmcnt(0) drain per plane and row
  const T *cp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cp[q] = P.coef[wv + 4 * q < 14 ? wv + 4 * q : 0];
  auto row_index = [&](int r) {
    if (P.wrap) return r < 0 ? r + rows : (r >= rows ? r - rows : r);
    return r < 0 ? 0 : (r >= rows ? rows - 1 : r);
  };
  auto load_row = [&](Row &x, int r) {
    const long long ro = (long long)row_index(r) * nx + col;
    const long long rc = (long long)row_index(r - 1) * nx + col;
    mload<T, VEC>(x.u, P.u0 + boff + ro);
    mload<T, VEC>(x.v, P.v0 + boff + ro);
    if (need_prev) {
      mload<T, VEC>(x.u2, P.u2 + boff + rc);
      mload<T, VEC>(x.v2, P.v2 + boff + rc);
      mload<FB, VEC>(x.fu, P.fu_in + boff + rc);
      mload<FB, VEC>(x.fv, P.fv_in + boff + rc);
    }
    if (SHARED) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pidx = wv + 4 * q;  // planes 0..6 belong to row r, 7..13 to row r-1
        if (pidx < 14) mload<T, VEC>(x.share[q], cp[q] + (pidx < 7 ? ro : rc));
      }
      return;
    }
    mload<T, VEC>(x.rdyCu, P.coef[0] + ro);
    mload<T, VEC>(x.rdxCu, P.coef[1] + ro);
    mload<T, VEC>(x.rdxCv, P.coef[2] + ro);
    mload<T, VEC>(x.rdyCv, P.coef[3] + ro);
    mload<T, VEC>(x.a1, P.coef[4] + ro);
    mload<T, VEC>(x.a2, P.coef[5] + ro);
    mload<T, VEC>(x.rh, P.coef[6] + ro);
    mload<T, VEC>(x.b1, P.coef[7] + rc);
    mload<T, VEC>(x.b2, P.coef[8] + rc);
    mload<T, VEC>(x.rq, P.coef[9] + rc);
    mload<T, VEC>(x.cu1, P.coef[10] + rc);
    mload<T, VEC>(x.cu2, P.coef[11] + rc);
    mload<T, VEC>(x.cv1, P.coef[12] + rc);
    mload<T, VEC>(x.cv2, P.coef[13] + rc);
  };

  // state carried from row r-1 / r-2
  T vt_p[VEC], vh_p[VEC], uh_p[VEC], P_p[VEC], Q_p[VEC], R_pp[VEC], u_p[VEC], v_p[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) vt_p[k] = vh_p[k] = uh_p[k] = P_p[k] = Q_p[k] = R_pp[k] = u_p[k] = v_p[k] = T(0);

  auto update = [&](T x, T L, T x2, FB fbin, T &tk, FB &fb) {  // recurrence of one component (filter.py:259-283)
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

  auto step = [&](Row &x, int r) {
    if (SHARED) {
      const int buf = (r - (a - 1)) & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pidx = wv + 4 * q;
        if (pidx < 14) {
          MPack<T, VEC> pk;
#pragma unroll
          for (int k = 0; k < VEC; ++k) pk.s[k] = x.share[q][k];
          s_coef[buf][pidx][lane] = pk;
        }
      }
      __syncthreads();
      auto get = [&](T (&dst)[VEC], int pidx) {
        const MPack<T, VEC> pk = s_coef[buf][pidx][lane];
#pragma unroll
        for (int k = 0; k < VEC; ++k) dst[k] = pk.s[k];
      };
      get(x.rdyCu, 0); get(x.rdxCu, 1); get(x.rdxCv, 2); get(x.rdyCv, 3); get(x.a1, 4); get(x.a2, 5); get(x.rh, 6);
      get(x.b1, 7); get(x.b2, 8); get(x.rq, 9); get(x.cu1, 10); get(x.cu2, 11); get(x.cv1, 12); get(x.cv2, 13);
    }
    T ut[VEC], uh[VEC], vt[VEC], vh[VEC], Pr[VEC], Qr[VEC], Rm[VEC], Sm[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T su = csan(x.u[k]), sv = csan(x.v[k]);
      ut[k] = su * x.rdyCu[k];
      uh[k] = su * x.rdxCu[k];
      vt[k] = sv * x.rdxCv[k];
      vh[k] = sv * x.rdyCv[k];
    }
    const T ut_w = from_lower_lane0(ut[VEC - 1]);   // W u~(r)
    const T vh_e = from_upper_lane0(vh_p[0]);       // E v^(r-1)
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T utw = (k == 0) ? ut_w : ut[k > 0 ? k - 1 : 0];
      Pr[k] = x.a1[k] * (ut[k] - utw) - x.a2[k] * (vt[k] - vt_p[k]);
      Qr[k] = x.rh[k] * Pr[k];
      const T vhe = (k == VEC - 1) ? vh_e : vh_p[k < VEC - 1 ? k + 1 : k];
      Rm[k] = x.b1[k] * (vhe - vh_p[k]) + x.b2[k] * (uh[k] - uh_p[k]);
      Sm[k] = x.rq[k] * Rm[k];
    }
    const T P_e = from_upper_lane0(P_p[0]);         // E P(r-1)
    const T S_w = from_lower_lane0(Sm[VEC - 1]);    // W S(r-1)
    const int j = r - 1;
    T tu[VEC], tv[VEC];
    FB fu[VEC], fv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T pe = (k == VEC - 1) ? P_e : P_p[k < VEC - 1 ? k + 1 : k];
      const T sw = (k == 0) ? S_w : Sm[k > 0 ? k - 1 : 0];
      const T lu = x.cu1[k] * (P_p[k] - pe) + x.cu2[k] * (R_pp[k] - Rm[k]);
      const T lv = x.cv1[k] * (sw - Sm[k]) - x.cv2[k] * (Q_p[k] - Qr[k]);
      if (lapl) {
        tu[k] = lu;
        tv[k] = lv;
      } else {
        update(u_p[k], lu, x.u2[k], x.fu[k], tu[k], fu[k]);
        update(v_p[k], lv, x.v2[k], x.fv[k], tv[k], fv[k]);
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
      vt_p[k] = vt[k];
      vh_p[k] = vh[k];
      uh_p[k] = uh[k];
      P_p[k] = Pr[k];
      Q_p[k] = Qr[k];
      R_pp[k] = Rm[k];
      u_p[k] = x.u[k];
      v_p[k] = x.v[k];
    }
  };

  const int r_begin = a - 1, r_end = b + 1;  // rows delivered: [a-1, b]
  Row q0, q1;
  load_row(q0, r_begin);
  if (D >= 2) load_row(q1, min(r_begin + 1, r_end - 1));
  // the slot is consumed in place and re-loaded right after (no register copy): with D = 2 the other slot's
  // loads are in flight during the arithmetic
#define GCMF_CSLOT(Q, dd)                                    \
  if (r + (dd) < r_end) {                                    \
    step(Q, r + (dd));                                       \
    load_row(Q, min(r + (dd) + D, r_end - 1));               \
  }
  for (int r = r_begin; r < r_end; r += D) {
    GCMF_CSLOT(q0, 0)
    if (D >= 2) { GCMF_CSLOT(q1, 1) }
  }
#undef GCMF_CSLOT
}

static bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T, typename FB> static int launch_cs(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC, WI = W - 2 * VEC;
  const Geom &g = pl->g;
  CStreamP<T, FB> P;
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
  for (int k = 0; k < MAX_COEF; ++k) P.coef[k] = (const T *)g.coef[k];
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  P.nlev = (int)a.nbatch;
  P.nlev4 = (P.nlev + 3) / 4 * 4;
  P.lockstep = (P.nlev4 * 10 <= P.nlev * 11) ? 1 : 0;
  if (!P.lockstep) P.nlev4 = P.nlev;
  int H = pl->strip_rows;
  if (H <= 0) {
    // enough waves for a resident round (2 per SIMD), but strips of at most 48 rows: many (window, strip) groups
    // keep the 8 XCDs (which each take whole groups with all their levels) evenly loaded -- measured best 32..64
    long long want = 2048 / ((long long)P.nwx * a.nbatch);
    if (want < 1) want = 1;
    H = (int)((nrows + want - 1) / want);
    if (H > 48) H = 48;
    if (H < 16) H = 16;
  }
  if (H > nrows) H = nrows;
  P.H = H;
  const int nstrips = (nrows + H - 1) / H;
  P.ngroups = P.nwx * nstrips;
  P.wrap = g.south_wrap && g.north_wrap;
  // lock-step needs whole workgroups per group: pad the levels to a multiple of 4 unless that wastes > 10 %
  P.mode = a.mode;
  P.bstride = (long long)g.rows * g.nx;
  P.coef0 = a.coef0;
  P.coef1 = a.coef1;
  P.c = a.c;
  // waves per XCD sequence: groups are dealt to XCDs round-robin, each contributes nlev consecutive wave slots
  const long long groups_per_xcd = (P.ngroups + 7) / 8;
  const long long waves_per_xcd = groups_per_xcd * P.nlev4;
  const long long blocks_per_xcd = (waves_per_xcd + 3) / 4;
  dim3 block(256), grid((unsigned)(blocks_per_xcd * 8));
  note_kernel(pl, std::string("gcmf::k_cgrid_stream<") + tyname<T>() + ", " + tyname<FB>() + ", 2, " + (P.lockstep ? "true" : "false") + ">", 1);
  if (P.lockstep) hipLaunchKernelGGL((k_cgrid_stream<T, FB, 2, true>), grid, block, 0, s, P);
  else hipLaunchKernelGGL((k_cgrid_stream<T, FB, 2, false>), grid, block, 0, s, P);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

bool cgrid_stream_supported(const gcmf_plan *pl, const StepArgs &a) {
  if (pl->kind != K_CGRID) return false;
  const int vec = pl->d.dtype == GCMF_F64 ? 2 : 4;
  if (pl->g.nx % vec || pl->g.nx < vec) return false;
  if (pl->g.rows < 3) return false;
  for (int k = 0; k < 2; ++k)
    if (!al16(a.t1[k]) || !al16(a.t2[k]) || !al16(a.fb_in[k]) || !al16(a.t0[k]) || !al16(a.fb_out[k])) return false;
  for (int k = 0; k < MAX_COEF; ++k)
    if (!al16(pl->g.coef[k])) return false;
  return true;
}

int launch_cgrid_stream(gcmf_plan *pl, const StepArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_cs<double, double>(pl, a, s);
  if (a.fb_is_f32) return launch_cs<float, float>(pl, a, s);
  return launch_cs<float, double>(pl, a, s);
}

}  // namespace gcmf
