// Streaming C-grid kernel, S Chebyshev steps per pass over HBM (temporal blocking for the vector path).
//
// Same wave-march / lock-step / LDS-shared-coefficient structure as k_cgrid_stream (gcmf_cgrid_stream.hip), with S
// levels riding one row behind each other, all in registers:
//   iteration r (row r of T_{k-1} delivered, coefficient rows A_r = planes 0..6 at row r, B_{r-1} = planes 7..13 at
//   row r-1 exchanged through LDS):
//     level j = 1..S:  T_{k-1+j} row r-j  from T_{k-2+j} rows r-j-1..r-j+1, with the coefficient rows A_{r-j+1},
//                      B_{r-j}, i.e. the LDS slot filled j-1 iterations ago
// so the LDS coefficient ring has S+1 slots (S readable, one being refilled) and one barrier per row.  A pass reads
// T_{k-1}, T_{k-2}, fbar and (once per four levels) the 14 coefficient planes and writes T_{k+S-1}, T_{k+S-2}, fbar:
// 4w + 2f + 14w/4 + 4w + 2f bytes per cell.level whatever S is (78 B with f32 state and f64 fbar; 65 B for S = 1).
// Arithmetic per level is the single-step kernel's: bit-identical results.
#include "gcmf_multi_common.hpp"
#include "gcmf_recurrence.hpp"
#include <cstdlib>

// Row-loop unroll factor: the per-level state of one iteration (previous-row stresses, the last two output rows, the fbar
// pipeline) is handed to the next iteration by renaming inside an unrolled body instead of by register moves -- a third
// of the rolled loop's VALU instructions were `v_mov` (792 VALU per row at S = 5, 283 of them moves).
#ifndef GCMF_CG_UNROLL
#define GCMF_CG_UNROLL 1
#endif

namespace gcmf {

template <typename T, typename FB> struct CStream2P {
  const T *u0, *v0;          // T_{k-1}
  const T *up, *vp;          // T_{k-2}      (unused when first)
  const FB *fu_in, *fv_in;   //              (unused when first)
  T *u1o, *v1o;              // T_{k+S-2} out (unused when last)
  T *u2o, *v2o;              // T_{k+S-1} out (unused when last)
  FB *fu_out, *fv_out;
  const T *coef[MAX_COEF];
  int nx, rows, out_lo, out_hi;
  int H, nwx, ngroups, nlev, nlev4, wrap, first, last;
  long long bstride;
  double p0, pk[6], c;
  // backward (Clenshaw) evaluation (k_cgrid_stream2c): fu_in / fv_in = the constant input fields, p0 = p_n (first launch),
  // last launch: the result goes to du_out / dv_out as f64 (f32 state, default output) or to fu_out / fv_out (state dtype)
  double *du_out, *dv_out;
};

template <typename T> __device__ __forceinline__ T c2san(T x) {
  const bool isn = (x != x);
  const bool big = (mabs(x) > MLim<T>::big());
  const T clamped = big ? (x > T(0) ? MLim<T>::big() : -MLim<T>::big()) : x;
  return isn ? T(0) : clamped;
}

// state machine of one level of the C-grid operator: feed it row r, get L_u, L_v of row r-1
template <typename T, int VEC> struct CgLevel {
  T vt_p[VEC], vh_p[VEC], uh_p[VEC], P_p[VEC], Q_p[VEC], R_pp[VEC];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < VEC; ++k) vt_p[k] = vh_p[k] = uh_p[k] = P_p[k] = Q_p[k] = R_pp[k] = T(0);
  }
  // cA: rdyCu, rdxCu, rdxCv, rdyCv, a1, a2, rh of the fed row;  cB: b1, b2, rq, cu1, cu2, cv1, cv2 of the row before
  // FMA (the backward kernels, where nothing is bit-identical with numpy anyway): the second product of every stress / divergence line
  // rides on a fused multiply-add -- the operation order of k_cgrid_ring (gcmf_cgrid_ring.hip), which gives the same bits
  template <bool FMA = false>
  __device__ __forceinline__ void feed(const T (&su)[VEC], const T (&sv)[VEC], const T (&cA)[7][VEC], const T (&cB)[7][VEC],
                                       T (&lu)[VEC], T (&lv)[VEC]) {
    T ut[VEC], uh[VEC], vt[VEC], vh[VEC], Pr[VEC], Qr[VEC], Rm[VEC], Sm[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      ut[k] = su[k] * cA[0][k];
      uh[k] = su[k] * cA[1][k];
      vt[k] = sv[k] * cA[2][k];
      vh[k] = sv[k] * cA[3][k];
    }
    const T ut_w = from_lower_lane0(ut[VEC - 1]);
    const T vh_e = from_upper_lane0(vh_p[0]);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T utw = (k == 0) ? ut_w : ut[k > 0 ? k - 1 : 0];
      if constexpr (FMA) Pr[k] = rfma(cA[4][k], ut[k] - utw, -(cA[5][k] * (vt[k] - vt_p[k])));
      else Pr[k] = cA[4][k] * (ut[k] - utw) - cA[5][k] * (vt[k] - vt_p[k]);
      Qr[k] = cA[6][k] * Pr[k];
      const T vhe = (k == VEC - 1) ? vh_e : vh_p[k < VEC - 1 ? k + 1 : k];
      if constexpr (FMA) Rm[k] = rfma(cB[0][k], vhe - vh_p[k], cB[1][k] * (uh[k] - uh_p[k]));
      else Rm[k] = cB[0][k] * (vhe - vh_p[k]) + cB[1][k] * (uh[k] - uh_p[k]);
      Sm[k] = cB[2][k] * Rm[k];
    }
    const T P_e = from_upper_lane0(P_p[0]);
    const T S_w = from_lower_lane0(Sm[VEC - 1]);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T pe = (k == VEC - 1) ? P_e : P_p[k < VEC - 1 ? k + 1 : k];
      const T sw = (k == 0) ? S_w : Sm[k > 0 ? k - 1 : 0];
      if constexpr (FMA) {
        lu[k] = rfma(cB[3][k], P_p[k] - pe, cB[4][k] * (R_pp[k] - Rm[k]));
        lv[k] = rfma(cB[5][k], sw - Sm[k], -(cB[6][k] * (Q_p[k] - Qr[k])));
      } else {
        lu[k] = cB[3][k] * (P_p[k] - pe) + cB[4][k] * (R_pp[k] - Rm[k]);
        lv[k] = cB[5][k] * (sw - Sm[k]) - cB[6][k] * (Q_p[k] - Qr[k]);
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
    }
  }
};

// PRIV (single-level fields): one-wave workgroups, each an independent (window, strip) group; the wave fetches all 14
// coefficient rows itself, level 1 uses them straight from registers and levels 2..S read them back from the wave's own
// LDS ring of S-1 slots, refilled at the end of the iteration (no barrier, no shadow waves).
// CLEN: the polynomial evaluated backwards (Clenshaw, see gcmf_ringc_impl.hpp): the state is (b_{k+1}, b_{k+2}), the conveyor that
// carries fbar from level to level carries the row of the constant input f instead (b_k = p_k f + 2 A(b_{k+1}) - b_{k+2}), nothing
// is accumulated and only the last launch writes a result: 54 instead of 78 bytes per cell, level and pass for f32 state.
template <typename T, typename FB, int VEC, int S, int D, bool PRIV, bool CLEN>
__device__ __forceinline__ void cgrid_stream2_body(const CStream2P<T, FB> &P) {
  static_assert(!CLEN || std::is_same<FB, T>::value, "backward evaluation: the conveyor has the state's type");
  constexpr int M = (S + VEC - 1) / VEC * VEC;  // level j is stale j cells per side; windows start on a VEC boundary
  constexpr int W = 64 * VEC, WI = W - 2 * M;
  constexpr int NS = PRIV ? S - 1 : S + 1;
  constexpr int NSHARE = PRIV ? 14 : 4;
  constexpr int WPB = PRIV ? 1 : 4;  // waves per workgroup
  extern __shared__ __align__(16) unsigned char s_raw[];
  typedef MPack<T, VEC> CoefSlot[14][64];

  const int lane = threadIdx.x & 63, wv = PRIV ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  CoefSlot *s_coef = reinterpret_cast<CoefSlot *>(s_raw);
  const int blk = blockIdx.x;
  const int xcd = blk & 7, slot = (blk >> 3) * WPB + wv;
  const int nlevp = PRIV ? P.nlev : P.nlev4;
  const int group = (slot / nlevp) * 8 + xcd;
  int lev = slot % nlevp;
  if (group >= P.ngroups) return;  // shared mode: whole workgroups exit together; private mode has no barriers
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
  const bool first = P.first, last = P.last;

  struct Row {
    T u[VEC], v[VEC];      // T_{k-1} row r
    T up[VEC], vp[VEC];    // T_{k-2} row r-1
    FB fu[VEC], fv[VEC];   // fbar    row r-1
    T share[NSHARE][VEC];  // this wave's quarter of the coefficient rows (planes wv, wv+4, wv+8, wv+12); PRIV: all 14
  };
  // this wave's coefficient planes, resolved once: indexing the kernel argument inside the row loop costs a
  // dependent memory load plus a full vmcnt(0) drain per plane and row
  const T *cp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cp[q] = P.coef[wv + 4 * q < 14 ? wv + 4 * q : 0];
  auto row_index = [&](int r) {
    if (P.wrap) {
      r = r < 0 ? r + rows : (r >= rows ? r - rows : r);
      return r < 0 ? r + rows : (r >= rows ? r - rows : r);  // |overshoot| <= S + 1 may exceed one period on tiny grids
    }
    return r < 0 ? 0 : (r >= rows ? rows - 1 : r);
  };
  const T *upp = (CLEN && first) ? P.u0 : P.up, *vpp = (CLEN && first) ? P.v0 : P.vp;
  const T dscale = (CLEN && first) ? (T)P.p0 : T(1);
  auto load_row = [&](Row &x, int r) {
    const long long ro = (long long)row_index(r) * nx + col;
    const long long rc = (long long)row_index(r - 1) * nx + col;
    if (!shadow) {   // (wave-uniform) levels that pad the last workgroup of a tile are HELPER waves: they fetch and publish their
      //               share of the coefficient rows, keep the barriers, and load / compute nothing else
      mload<T, VEC>(x.u, P.u0 + boff + ro);
      mload<T, VEC>(x.v, P.v0 + boff + ro);
      if (!first || CLEN) {  // (backward, first launch: d_n = b_n = p_n f -- the rows of f again, scaled at the use; a select there made
        //                        the compiler wait for the load it had just issued: 17 % on the B-grid twin of this kernel)
        mload<T, VEC>(x.up, upp + boff + rc);
        mload<T, VEC>(x.vp, vpp + boff + rc);
      }
      if (!first || CLEN) {  // fbar -- or, backward evaluation, the row of the constant input
        mload<FB, VEC>(x.fu, P.fu_in + boff + rc);
        mload<FB, VEC>(x.fv, P.fv_in + boff + rc);
      }
    }
    if (PRIV) {
#pragma unroll
      for (int q = 0; q < NSHARE; ++q) mload<T, VEC>(x.share[q], P.coef[q < 14 ? q : 0] + (q < 7 ? ro : rc));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pidx = wv + 4 * q;
        if (pidx < 14) mload<T, VEC>(x.share[q], cp[q] + (pidx < 7 ? ro : rc));
      }
    }
  };

  // Level j (1..S) consumes the output of level j-1 (level 0 = the delivered T_{k-1}):
  //   fed with its newest row, "-x" is its row of one iteration ago (o1), "-T_{k-2}" is level j-2's row of two
  //   iterations ago (o2; for level 1 the lagged T_{k-2} load).  fbar after level j waits one iteration for level j+1.
  CgLevel<T, VEC> L[S];
  T o1u[S][VEC], o1v[S][VEC], o2u[S][VEC], o2v[S][VEC];
  T dou[S + 1][VEC], dov[S + 1][VEC];   // backward evaluation: the row of d = b_k + b_{k+1} level j produced one iteration ago (see below)
  FB accu[S][VEC], accv[S][VEC];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    L[j].init();
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      o1u[j][k] = o1v[j][k] = o2u[j][k] = o2v[j][k] = T(0);
      dou[j][k] = dov[j][k] = dou[j + 1][k] = dov[j + 1][k] = T(0);
      accu[j][k] = accv[j][k] = FB(0);
    }
  }
  int cur = 0;  // LDS ring slot of this iteration

  auto publish = [&](Row &x) {
#pragma unroll
    for (int q = 0; q < NSHARE; ++q) {
      const int pidx = PRIV ? q : wv + 4 * q;
      if (pidx < 14) {
        MPack<T, VEC> pk;
#pragma unroll
        for (int k = 0; k < VEC; ++k) pk.s[k] = x.share[q][k];
        s_coef[cur][pidx][lane] = pk;
      }
    }
  };

  auto step = [&](Row &x, int r) {
    // ---- exchange the coefficient rows of this iteration through LDS ----
    if (!PRIV) {
      publish(x);
      __syncthreads();
      if (shadow) {   // a helper wave: its rows are published, nothing to compute
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        return;
      }
    }

    T cu[S + 1][VEC], cv[S + 1][VEC];  // newest row of every level this iteration
    FB nau[S + 1][VEC], nav[S + 1][VEC];
    T dnu[S + 1][VEC], dnv[S + 1][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      // (backward evaluation, first launch: the delivered rows of f become rows of b_n = p_n f)
      cu[0][k] = (CLEN && first) ? (T)P.p0 * x.u[k] : x.u[k];
      cv[0][k] = (CLEN && first) ? (T)P.p0 * x.v[k] : x.v[k];
    }

#pragma unroll
    for (int j = 1; j <= S; ++j) {
      int sl = cur - (j - 1);
      if (sl < 0) sl += NS;
      T cA[7][VEC], cB[7][VEC];
      if (PRIV && j == 1) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) { cA[q][k] = x.share[q][k]; cB[q][k] = x.share[PRIV ? 7 + q : 0][k]; }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const MPack<T, VEC> pa = s_coef[sl][q][lane];
          const MPack<T, VEC> pb = s_coef[sl][7 + q][lane];
#pragma unroll
          for (int k = 0; k < VEC; ++k) { cA[q][k] = pa.s[k]; cB[q][k] = pb.s[k]; }
        }
      }
      T su[VEC], sv[VEC], lu[VEC], lv[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) { su[k] = c2san(cu[j - 1][k]); sv[k] = c2san(cv[j - 1][k]); }
      L[j - 1].template feed<CLEN>(su, sv, cA, cB, lu, lv);
      const double pkj = P.pk[j - 1];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T xu = o1u[j - 1][k], xv = o1v[j - 1][k];
        const T avu = -xu - c * lu[k], avv = -xv - c * lv[k];
        if constexpr (CLEN) {
          // REINSCH'S FORM of Clenshaw's recurrence (round 5).  A = -1 - cL maps the large scales -- most of a filtered field -- to the end
          // x = -1 of the Chebyshev interval, where b_k = p_k f + 2 A b_{k+1} - b_{k+2} subtracts nearly equal numbers and its rounding
          // errors grow like k.  Carrying d_k = b_k + b_{k+1} instead of b_{k+2},
          //       d_k = p_k f - 2 c L(b_{k+1}) - d_{k+1},      b_k = d_k - b_{k+1},      result = p_0 f - c L(b_1) - d_1,
          // is the same polynomial, the same two state planes (b_{k+1}, d_{k+1}) and the same operation count without that cancellation:
          // f32 fields at n_steps 44 come out 1.8e-6 from f64 arithmetic instead of 4.9e-6 -- the reference's own f32 path (f32 T_k, f64
          // running sum): 2.7e-6; at n_steps 98: 5.0e-6 / 1.9e-5 / 8.5e-6.  The state planes hold (b_{k+1}, d_{k+1}); b_n = d_n = p_n f.
          const T dpu = (j == 1) ? dscale * x.up[k] : dou[j - 1][k];   // (dscale = 1 but for the first launch: exact)
          const T dpv = (j == 1) ? dscale * x.vp[k] : dov[j - 1][k];
          const T fiu = (T)((j == 1) ? x.fu[k] : accu[j - 1][k]);
          const T fiv = (T)((j == 1) ? x.fv[k] : accv[j - 1][k]);
          const bool fin = last && j == S;               // the last level of the last launch is the result: c L, not 2 c L, and no b_0
          const T mtc = fin ? -c : T(-2) * c;
          // (nothing here is bit-identical with numpy: every multiply-add pair is one fma -- fewer instructions, fewer roundings)
          const T dku = rfma((T)pkj, fiu, rfma(mtc, lu[k], -dpu)), dkv = rfma((T)pkj, fiv, rfma(mtc, lv[k], -dpv));
          dnu[j][k] = dku;
          dnv[j][k] = dkv;
          cu[j][k] = fin ? dku : dku - xu;
          cv[j][k] = fin ? dkv : dkv - xv;
          nau[j][k] = (FB)fiu;   // the row of f travels on with its row of the state
          nav[j][k] = (FB)fiv;
        } else if (j == 1 && first) {
          cu[j][k] = avu;
          cv[j][k] = avv;
          if (std::is_same<FB, T>::value) {
            nau[j][k] = (FB)((T)P.p0 * xu + (T)pkj * avu);
            nav[j][k] = (FB)((T)P.p0 * xv + (T)pkj * avv);
          } else {
            nau[j][k] = (FB)(P.p0 * (double)xu + pkj * (double)avu);
            nav[j][k] = (FB)(P.p0 * (double)xv + pkj * (double)avv);
          }
        } else {
          const T x2u = (j == 1) ? x.up[k] : o2u[j >= 2 ? j - 2 : 0][k];
          const T x2v = (j == 1) ? x.vp[k] : o2v[j >= 2 ? j - 2 : 0][k];
          const FB fiu = (j == 1) ? x.fu[k] : accu[j - 1][k];
          const FB fiv = (j == 1) ? x.fv[k] : accv[j - 1][k];
          cu[j][k] = T(2) * avu - x2u;
          cv[j][k] = T(2) * avv - x2v;
          if (std::is_same<FB, T>::value) {
            nau[j][k] = fiu + (FB)((T)pkj * cu[j][k]);
            nav[j][k] = fiv + (FB)((T)pkj * cv[j][k]);
          } else {
            nau[j][k] = fiu + (FB)(pkj * (double)cu[j][k]);
            nav[j][k] = fiv + (FB)(pkj * (double)cv[j][k]);
          }
        }
      }
      if (j >= (CLEN ? S : S - 1) && keep && r - j >= a && r - j < b) {
        const long long off = boff + (long long)(r - j) * nx + col;
        if (!last) {
          if constexpr (CLEN) {   // both states of the next launch are level S's: b (-> u2o) and d (-> u1o), the same row
            mstore<T, VEC>(P.u2o + off, cu[j]);
            mstore<T, VEC>(P.v2o + off, cv[j]);
            mstore<T, VEC>(P.u1o + off, dnu[j]);
            mstore<T, VEC>(P.v1o + off, dnv[j]);
          } else {
            mstore<T, VEC>((j == S ? P.u2o : P.u1o) + off, cu[j]);
            mstore<T, VEC>((j == S ? P.v2o : P.v1o) + off, cv[j]);
          }
        }
        if (j == S && !CLEN) {
          mstore<FB, VEC>(P.fu_out + off, nau[j]);
          mstore<FB, VEC>(P.fv_out + off, nav[j]);
        }
        if (CLEN && j == S && last) {
          if (P.du_out) {  // wave-uniform: f64 result from f32 state (one component at a time: registers)
            double dd[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) dd[k] = (double)cu[j][k];
            mstore<double, VEC>(P.du_out + off, dd);
#pragma unroll
            for (int k = 0; k < VEC; ++k) dd[k] = (double)cv[j][k];
            mstore<double, VEC>(P.dv_out + off, dd);
          } else {
            mstore<T, VEC>(reinterpret_cast<T *>(P.fu_out) + off, cu[j]);
            mstore<T, VEC>(reinterpret_cast<T *>(P.fv_out) + off, cv[j]);
          }
        }
      }
    }

    // ---- rotate ----
#pragma unroll
    for (int j = 0; j < S; ++j) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        o2u[j][k] = o1u[j][k];
        o2v[j][k] = o1v[j][k];
        o1u[j][k] = cu[j][k];
        o1v[j][k] = cv[j][k];
        if (j >= 1) { accu[j][k] = nau[j][k]; accv[j][k] = nav[j][k]; }
        if (CLEN) { dou[j + 1][k] = dnu[j + 1][k]; dov[j + 1][k] = dnv[j + 1][k]; }
      }
    }
    if (PRIV) publish(x);  // after level S has read the slot this overwrites (same wave: LDS executes in order)
    cur = (cur + 1 == NS) ? 0 : cur + 1;
  };

  const int r_begin = a - S, r_end = b + S;  // rows delivered: [a-S, b+S-1]
  if (D == 1) {  // one row of operands in flight per wave; the other waves of the SIMD hide the rest of the latency
    Row nxt;
    load_row(nxt, r_begin);
    // unrolled by hand (a loop with a barrier is not unrolled by the compiler when its trip count is unknown); the march
    // is padded to a whole number of bodies: the extra iterations re-load the last row and store nothing
    constexpr int U = GCMF_CG_UNROLL;
    const int r_pad = r_begin + (r_end - r_begin + U - 1) / U * U;
    for (int r = r_begin; r < r_pad; r += U) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        Row now = nxt;
        load_row(nxt, min(r + q + 1, r_end - 1));
        step(now, r + q);
      }
    }
  } else {  // two rows in flight
    Row q0, q1;
    load_row(q0, r_begin);
    load_row(q1, min(r_begin + 1, r_end - 1));
    for (int r = r_begin; r < r_end; r += 2) {
      {
        Row now = q0;
        load_row(q0, min(r + 2, r_end - 1));
        step(now, r);
      }
      if (r + 1 < r_end) {
        Row now = q1;
        load_row(q1, min(r + 3, r_end - 1));
        step(now, r + 1);
      }
    }
  }
}

template <typename T, typename FB, int VEC, int S, int D, bool PRIV>
__global__ __launch_bounds__((PRIV ? 64 : 256), (sizeof(T) * VEC == 16 && (PRIV || S > 2) ? 1 : 2)) void k_cgrid_stream2(const CStream2P<T, FB> P) {
  cgrid_stream2_body<T, FB, VEC, S, D, PRIV, false>(P);
}
template <typename T, int VEC, int S, int D, bool PRIV>
__global__ __launch_bounds__((PRIV ? 64 : 256), (sizeof(T) * VEC == 16 && (PRIV || S > 2) ? 1 : 2)) void k_cgrid_stream2c(const CStream2P<T, T> P) {
  cgrid_stream2_body<T, T, VEC, S, D, PRIV, true>(P);
}

static bool c2al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool cgrid_multi_supported(const gcmf_plan *pl, int64_t nbatch, int S, bool backward) {
  if (pl->kind != K_CGRID || pl->cgrid_tile) return false;
  // f64: more than two levels spill registers.  f32: S = 6 / 8 fit only one wave per SIMD and measured slower than
  // S = 4 at two (234-252 G against 268-274 G cell.steps/s on config 5)
  // f32: S = 5 holds 238 VGPRs at two waves per SIMD and measured 282 G against 250 G for S = 4 on the same box
  // (config 5); S = 6 spills (229 G).  f64: S <= 4 (one wave per SIMD from S = 3 on)
  // single-level fields (wave-private LDS rings): S = 5 leaves five waves per CU and measured 90 G against 133 G at S = 4
  // (six levels: only k_cgrid_ring, i.e. the backward evaluation of batched f32 levels)
  if (S < 2 || (S > ((pl->d.dtype == GCMF_F64 || nbatch == 1) ? 4 : 5) && !(backward && cgrid_ring_supported(pl, nbatch, S)))) return false;
  const int vec = pl->d.dtype == GCMF_F64 ? 2 : 4;
  if (pl->g.nx % vec || pl->g.nx < vec || pl->g.rows < S + 2) return false;
  // any batch size: the lock-step workgroups of 4 levels are padded with shadow waves that repeat the last level
  // without storing.  Even 1 level + 3 shadows beats the single-step kernel (83 G against 40 G cell.steps/s on
  // 2400x3600 f32; 2 levels 163 against 54; 5 levels 198 against 62): the shadows' loads hit in L2.
  if (nbatch < 1) return false;
  for (int k = 0; k < MAX_COEF; ++k)
    if (!c2al16(pl->g.coef[k])) return false;
  return true;
}

template <typename T, typename FB, int VEC, int S, int D, bool PRIV, bool CLEN = false> static int launch_c2(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  constexpr int M = (S + VEC - 1) / VEC * VEC, W = 64 * VEC, WI = W - 2 * M;
  const Geom &g = pl->g;
  CStream2P<T, FB> P;
  P.u0 = (const T *)a.u0[0];  P.v0 = (const T *)a.u0[1];
  P.up = (const T *)a.uprev[0];  P.vp = (const T *)a.uprev[1];
  P.fu_in = (const FB *)a.fb_in[0];  P.fv_in = (const FB *)a.fb_in[1];
  P.u1o = (T *)a.u1o[0];  P.v1o = (T *)a.u1o[1];
  P.u2o = (T *)a.u2o[0];  P.v2o = (T *)a.u2o[1];
  P.fu_out = (FB *)a.fb_out[0];  P.fv_out = (FB *)a.fb_out[1];
  P.du_out = P.dv_out = nullptr;
  if (CLEN && a.last && sizeof(T) == 4 && !a.fb_is_f32) {  // f32 state, f64 result (NumPy >= 2 promotion of the reference)
    P.du_out = (double *)a.fb_out[0];
    P.dv_out = (double *)a.fb_out[1];
  }
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
  int H = pl->strip_rows;
  if (H <= 0) {
    // strips as tall as possible (a strip marches H + 2S rows) while the launch still fills whole rounds of the
    // 2048 resident waves (2 per SIMD): the fewest strips of <= 96 rows (64-96 measured best on config 5: 273 G
    // against 262 G at 160) fix the number of rounds, then the strip
    // count grows to fill the last round
    long long cap = (sizeof(T) * VEC == 16 && S > 2) ? 1024 : 2048;  // 16 bytes per lane beyond two levels: one wave per SIMD
    if (PRIV) {  // one-wave workgroups: registers (f64: one wave per SIMD) or the LDS ring bound the residency
      const long long by_lds = (160 * 1024) / ((long long)(S - 1) * 14 * 64 * sizeof(MPack<T, VEC>));
      const long long by_reg = sizeof(T) == 8 ? 4 : 8;
      cap = 256 * (by_lds < by_reg ? by_lds : by_reg) * 85 / 100;  // a little headroom measured best (139 vs 132 G)
    }
    const long long per_strip = (long long)P.nwx * (PRIV ? P.nlev : P.nlev4), hmax = 96;
    const long long ns_min = (nrows + hmax - 1) / hmax;
    const long long rounds = (ns_min * per_strip + cap - 1) / cap;
    long long ns = rounds * cap / per_strip;
    if (ns < ns_min) ns = ns_min;
    H = (int)((nrows + ns - 1) / ns);
    if (H < 16) H = 16;
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.ngroups = P.nwx * ((nrows + H - 1) / H);
  P.wrap = g.south_wrap && g.north_wrap;
  P.first = a.first;
  P.last = a.last;
  P.bstride = (long long)g.rows * g.nx;
  P.p0 = a.p0;
  for (int t = 0; t < 6; ++t) P.pk[t] = a.pk[t];
  P.c = a.c;
  const long long groups_per_xcd = (P.ngroups + 7) / 8;
  const long long blocks_per_xcd = PRIV ? groups_per_xcd * P.nlev : (groups_per_xcd * P.nlev4 + 3) / 4;
  dim3 block(PRIV ? 64 : 256), grid((unsigned)(blocks_per_xcd * 8));
  const size_t lds = (size_t)(PRIV ? S - 1 : S + 1) * 14 * 64 * sizeof(MPack<T, VEC>);
  static bool attr_set = false;  // per instantiation
  if constexpr (CLEN) {
    if (!attr_set && lds > 48 * 1024) {
      GCMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cgrid_stream2c<T, VEC, S, D, PRIV>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    hipLaunchKernelGGL((k_cgrid_stream2c<T, VEC, S, D, PRIV>), grid, block, lds, s, P);
    note_kernel(pl, std::string("gcmf::k_cgrid_stream2c<") + tyname<T>() + ", " + std::to_string(VEC) + ", " + std::to_string(S) + ", " +
                        std::to_string(D) + ", " + (PRIV ? "true" : "false") + ">", S,
                launch_geom(P.H, (nrows + H - 1) / H, P.nwx, 1, grid.x, grid.y, nrows));
  } else {
  if (!attr_set && lds > 48 * 1024) {
    GCMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cgrid_stream2<T, FB, VEC, S, D, PRIV>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_cgrid_stream2<T, FB, VEC, S, D, PRIV>), grid, block, lds, s, P);
  note_kernel(pl, std::string("gcmf::k_cgrid_stream2<") + tyname<T>() + ", " + tyname<FB>() + ", " + std::to_string(VEC) + ", " +
                      std::to_string(S) + ", " + std::to_string(D) + ", " + (PRIV ? "true" : "false") + ">", S,
              launch_geom(P.H, (nrows + H - 1) / H, P.nwx, 1, grid.x, grid.y, nrows));
  }
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

template <typename T, typename FB, int S> static int launch_c2_sel(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  // single-level fields: private coefficient rings (no shadow waves); env GCMF_VEC_PRIV=0 keeps the padded lock-step form
  static const bool priv_ok = !(getenv("GCMF_VEC_PRIV") && atoi(getenv("GCMF_VEC_PRIV")) == 0);
  if (a.nbatch == 1 && priv_ok) return launch_c2<T, FB, 2, S, 1, true>(pl, a, s);  // 14 rows per operand row: one in flight
  // two operand rows in flight per wave unless tuned down (f64 at S = 2: the registers of two waves per SIMD allow one)
  if constexpr (sizeof(T) == 8 && S == 2) {
    return launch_c2<T, FB, 2, S, 1, false>(pl, a, s);
  } else {
    if (pl->prefetch_rows == 1) return launch_c2<T, FB, 2, S, 1, false>(pl, a, s);
    return launch_c2<T, FB, 2, S, 2, false>(pl, a, s);
  }
}

int launch_cgrid_multi(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  if (a.clen && cgrid_ring_supported(pl, a.nbatch, a.S) && cgrid_ring_args_aligned(a)) return launch_cgrid_ring(pl, a, s);   // batched f32 levels, deep launches
  if (!a.clen && cgrid_ringf_supported(pl, a)) return launch_cgrid_ringf(pl, a, s);   // ... of the forward recurrence with an f64 running sum (round 6)
  if (a.clen) {  // backward evaluation: the conveyor has the state's type; single-level fields: private coefficient rings as above
    static const bool priv_ok = !(getenv("GCMF_VEC_PRIV") && atoi(getenv("GCMF_VEC_PRIV")) == 0);
    const bool priv = a.nbatch == 1 && priv_ok;
    if (pl->d.dtype == GCMF_F64) {
      switch (a.S) {
        case 2: return priv ? launch_c2<double, double, 2, 2, 1, true, true>(pl, a, s) : launch_c2<double, double, 2, 2, 1, false, true>(pl, a, s);
        case 3: return priv ? launch_c2<double, double, 2, 3, 1, true, true>(pl, a, s) : launch_c2<double, double, 2, 3, 2, false, true>(pl, a, s);
        case 4: return priv ? launch_c2<double, double, 2, 4, 1, true, true>(pl, a, s) : launch_c2<double, double, 2, 4, 2, false, true>(pl, a, s);
      }
      return GCMF_ERR_INVALID_ARG;
    }
    switch (a.S) {
      case 2: return priv ? launch_c2<float, float, 2, 2, 1, true, true>(pl, a, s) : launch_c2<float, float, 2, 2, 2, false, true>(pl, a, s);
      case 3: return priv ? launch_c2<float, float, 2, 3, 1, true, true>(pl, a, s) : launch_c2<float, float, 2, 3, 2, false, true>(pl, a, s);
      // (round 4: FOUR cells per lane -- 16-byte accesses, 256-cell windows, 416 registers = one wave per SIMD like the f64 kernel -- measured
      // 331 G against 358 G on config 5, same bits; experiments/README.md)
      case 4: return priv ? launch_c2<float, float, 2, 4, 1, true, true>(pl, a, s) : launch_c2<float, float, 2, 4, 2, false, true>(pl, a, s);
      case 5: return launch_c2<float, float, 2, 5, 1, false, true>(pl, a, s);
    }
    return GCMF_ERR_INVALID_ARG;
  }
  if (pl->d.dtype == GCMF_F64) {
    switch (a.S) {
      case 2: return launch_c2_sel<double, double, 2>(pl, a, s);
      case 3: return launch_c2_sel<double, double, 3>(pl, a, s);
      case 4: return launch_c2_sel<double, double, 4>(pl, a, s);
    }
    return GCMF_ERR_INVALID_ARG;
  }
  // f32: two cells per lane (8-byte accesses).  Four would need > 256 registers per lane already for two levels;
  // with two cells S = 2 runs three waves per SIMD and measured 115 G cell.steps/s against 87 G (config 5).
  switch (a.S * 2 + (a.fb_is_f32 ? 1 : 0)) {
    case 4: return launch_c2_sel<float, double, 2>(pl, a, s);
    case 5: return launch_c2_sel<float, float, 2>(pl, a, s);
    case 6: return launch_c2_sel<float, double, 3>(pl, a, s);
    case 7: return launch_c2_sel<float, float, 3>(pl, a, s);
    case 8: return launch_c2_sel<float, double, 4>(pl, a, s);
    case 9: return launch_c2_sel<float, float, 4>(pl, a, s);
    case 10: return launch_c2<float, double, 2, 5, 1, false>(pl, a, s);
    case 11: return launch_c2<float, float, 2, 5, 1, false>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
