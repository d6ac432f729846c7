// Streaming B-grid kernel, S Chebyshev steps per pass over HBM (VECTOR_B_GRID, reference gcm_filters/kernels.py:740-837
// inside the recurrence of filter.py:225-283).
//
// The temporally blocked counterpart of k_bgrid_stream, built like k_cgrid_stream2 (gcmf_cgrid_stream2.hip): a wave owns
// 64*VEC contiguous cells and marches north; level j = 1..S rides one row behind level j-1 in registers (3-row window of
// the sanitised u and v, the last two raw output rows for the "-x" / "-T_{k-2}" operands, one delayed fbar accumulator);
// the 4 waves of a workgroup are 4 levels of one (window, strip) group in lock-step, each fetches 2 of the 8 coefficient
// rows and they are exchanged through an LDS ring of S+1 slots (level j needs the rows delivered j-1 iterations ago).
// A pass reads T_{k-1}, T_{k-2}, fbar and 1/4 of the coefficient rows and writes T_{k+S-1}, T_{k+S-2}, fbar:
// 8w + 4f + 8w/4 bytes per cell.level whatever S is.  Batches that do not fill the 4 levels are padded with shadow waves.
// Per level the summation order is the reference's, term by term: bit-identical to S single steps and to numpy.
//
// CLEN (round 3, opt-in with GCMF_CLENSHAW=2 like the other grid types that are bit-exact with numpy forward): the polynomial evaluated
// backwards, b_k = p_k f + 2 A(b_{k+1}) - b_{k+2} (gcmf_ringc_impl.hpp) -- the conveyor that carries fbar from level to level carries the
// row of the constant input instead, nothing is accumulated, only the last launch writes a result: 4w + 2w (+ coefficients) read and 4w
// written per cell and level instead of 4w + 2f / 4w + 2f, every multiply-add pair one fma.  Not bit-identical with numpy (<= 1e-14).
// Round 5: in REINSCH's form -- the state planes hold b_{k+1} and d_{k+1} = b_{k+1} + b_{k+2}; d_k = p_k f - 2c L(b_{k+1}) - d_{k+1},
// b_k = d_k - b_{k+1}, result = p_0 f - c L(b_1) - d_1.  Same planes, same traffic, one more subtraction; the cancellation of
// 2 b_{k+1} - b_{k+2} (the terms of a smooth filter nearly cancel there) no longer happens in the state's precision: f32 fields are
// 2.5e-6 from f64 arithmetic at n_steps 44 instead of 6.8e-6 (the reference's own f32 path: 1.2e-6).
#include "gcmf_multi_common.hpp"
#include "gcmf_recurrence.hpp"
#include <cstdlib>

namespace gcmf {

template <typename T, typename FB> struct BStream2P {
  const T *u0, *v0;          // T_{k-1}
  const T *up, *vp;          // T_{k-2}       (unused when first)
  const FB *fu_in, *fv_in;   //               (unused when first)
  T *u1o, *v1o;              // T_{k+S-2} out (unused when last)
  T *u2o, *v2o;              // T_{k+S-1} out (unused when last)
  FB *fu_out, *fv_out;
  const T *coef[8];          // cc, DUN, DUS, DUE, DUW, DMC, DMN, DME
  int nx, rows, out_lo, out_hi;
  int H, nwx, ngroups, nlev, nlev4, wrap, first, last;
  long long bstride;
  double p0, pk[6], c;
  // backward (Clenshaw) evaluation: fu_in / fv_in = the constant input fields, p0 = p_n (first launch); last launch: the result goes
  // to du_out / dv_out as f64 (f32 state, default output) or to fu_out / fv_out (state dtype)
  double *du_out, *dv_out;
};

template <typename T> __device__ __forceinline__ T b2san(T x) {
  const bool isn = (x != x);
  const bool big = (mabs(x) > MLim<T>::big());
  const T clamped = big ? (x > T(0) ? MLim<T>::big() : -MLim<T>::big()) : x;
  return isn ? T(0) : clamped;
}

// one level of the B-grid operator: feed it the sanitised row r, get L_u, L_v of row r-1 (coefficients of row r-1)
template <typename T, int VEC> struct BgLevel {
  T uS[VEC], uC[VEC], vS[VEC], vC[VEC];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < VEC; ++k) uS[k] = uC[k] = vS[k] = vC[k] = T(0);
  }
  // FMA (the backward kernels, round 6: nothing there is bit-identical with numpy anyway): the same terms in the same order, every
  // multiply-add pair one fused multiply-add -- 10 instead of 19 dependent instructions per component
  template <bool FMA = false>
  __device__ __forceinline__ void feed(const T (&uN)[VEC], const T (&vN)[VEC], const T (&K)[8][VEC], T (&lu)[VEC],
                                       T (&lv)[VEC]) {
    const T uw_ = from_lower_lane0(uC[VEC - 1]), ue_ = from_upper_lane0(uC[0]);
    const T vw_ = from_lower_lane0(vC[VEC - 1]), ve_ = from_upper_lane0(vC[0]);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T uc = uC[k], vc = vC[k];
      const T uw = (k == 0) ? uw_ : uC[k > 0 ? k - 1 : 0], ue = (k == VEC - 1) ? ue_ : uC[k < VEC - 1 ? k + 1 : k];
      const T vw = (k == 0) ? vw_ : vC[k > 0 ? k - 1 : 0], ve = (k == VEC - 1) ? ve_ : vC[k < VEC - 1 ? k + 1 : k];
      const T cc = K[0][k], dun = K[1][k], dus = K[2][k], due = K[3][k], duw = K[4][k];
      const T dmc = K[5][k], dmn = K[6][k], dme = K[7][k];
      const T dms = -dmn, dmw = -dme;
      // reference summation order (kernels.py:811-835)
      T a, b;
      if constexpr (FMA) {
        a = rfma(dun, uN[k], cc * uc);
        a = rfma(dus, uS[k], a); a = rfma(due, ue, a); a = rfma(duw, uw, a); a = rfma(dmc, vc, a);
        a = rfma(dmn, vN[k], a); a = rfma(dms, vS[k], a); a = rfma(dme, ve, a); a = rfma(dmw, vw, a);
        b = rfma(dun, vN[k], cc * vc);
        b = rfma(dus, vS[k], b); b = rfma(due, ve, b); b = rfma(duw, vw, b); b = rfma(dmc, uc, b);
        b = rfma(dmn, uN[k], b); b = rfma(dms, uS[k], b); b = rfma(dme, ue, b); b = rfma(dmw, uw, b);
      } else {
        a = cc * uc + dun * uN[k];
        a = a + dus * uS[k]; a = a + due * ue; a = a + duw * uw; a = a + dmc * vc;
        a = a + dmn * vN[k]; a = a + dms * vS[k]; a = a + dme * ve; a = a + dmw * vw;
        b = cc * vc + dun * vN[k];
        b = b + dus * vS[k]; b = b + due * ve; b = b + duw * vw; b = b + dmc * uc;
        b = b + dmn * uN[k]; b = b + dms * uS[k]; b = b + dme * ue; b = b + dmw * uw;
      }
      lu[k] = a;
      lv[k] = b;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      uS[k] = uC[k]; uC[k] = uN[k];
      vS[k] = vC[k]; vC[k] = vN[k];
    }
  }
};

// PRIV (single-level fields): one-wave workgroups, each an independent (window, strip) group; the wave fetches all 8
// coefficient rows itself, level 1 uses them straight from registers and levels 2..S read them back from the wave's own
// LDS ring of S-1 slots, refilled at the end of the iteration (no barrier, no shadow waves).
template <typename T, typename FB, int VEC, int S, int D, bool PRIV, bool CLEN>
__device__ __forceinline__ void bgrid_stream2_body(const BStream2P<T, FB> &P) {
  static_assert(!CLEN || std::is_same<FB, T>::value, "backward evaluation: the conveyor has the state's type");
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int W = 64 * VEC, WI = W - 2 * M;
  constexpr int NS = PRIV ? S - 1 : S + 1;
  constexpr int NSHARE = PRIV ? 8 : 2;
  constexpr int WPB = PRIV ? 1 : 4;  // waves per workgroup
  extern __shared__ __align__(16) unsigned char s_raw[];
  typedef MPack<T, VEC> CoefSlot[8][64];
  CoefSlot *s_coef = reinterpret_cast<CoefSlot *>(s_raw);

  const int lane = threadIdx.x & 63, wv = PRIV ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    T share[NSHARE][VEC];  // this wave's quarter of the coefficient rows r-1 (planes wv, wv+4); PRIV: all 8
  };
  // this wave's coefficient planes, resolved once (never index the kernel argument inside the row loop)
  const T *cp[2] = {P.coef[wv], P.coef[wv + 4]};
  (void)cp;
  auto row_index = [&](int r) {
    if (P.wrap) {
      r = r < 0 ? r + rows : (r >= rows ? r - rows : r);
      return r < 0 ? r + rows : (r >= rows ? r - rows : r);
    }
    return r < 0 ? 0 : (r >= rows ? rows - 1 : r);
  };
  const T *upp = (CLEN && first) ? P.u0 : P.up, *vpp = (CLEN && first) ? P.v0 : P.vp;
  const T dscale = (CLEN && first) ? (T)P.p0 : T(1);
  auto load_row = [&](Row &x, int r) {
    const long long ro = (long long)row_index(r) * nx + col;
    const long long rc = (long long)row_index(r - 1) * nx + col;
    mload<T, VEC>(x.u, P.u0 + boff + ro);
    mload<T, VEC>(x.v, P.v0 + boff + ro);
    if (!first || CLEN) {   // (backward, first launch: d_n = b_n = p_n f -- the same rows of f again, scaled below; no select in the loop)
      mload<T, VEC>(x.up, upp + boff + rc);
      mload<T, VEC>(x.vp, vpp + boff + rc);
    }
    if (!first || CLEN) {   // fbar -- or, backward evaluation, the row of the constant input
      mload<FB, VEC>(x.fu, P.fu_in + boff + rc);
      mload<FB, VEC>(x.fv, P.fv_in + boff + rc);
    }
    if (PRIV) {
#pragma unroll
      for (int q = 0; q < NSHARE; ++q) mload<T, VEC>(x.share[q], P.coef[q < 8 ? q : 0] + rc);
    } else {
      mload<T, VEC>(x.share[0], cp[0] + rc);
      mload<T, VEC>(x.share[1], cp[1] + rc);
    }
  };

  BgLevel<T, VEC> L[S];
  T o1u[S][VEC], o1v[S][VEC], o2u[S][VEC], o2v[S][VEC];
  T d1u[S][VEC], d1v[S][VEC];  // backward evaluation: last iteration's d rows (o2u / o2v are then unused)
  FB accu[S][VEC], accv[S][VEC];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    L[j].init();
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      o1u[j][k] = o1v[j][k] = o2u[j][k] = o2v[j][k] = d1u[j][k] = d1v[j][k] = T(0);
      accu[j][k] = accv[j][k] = FB(0);
    }
  }
  int cur = 0;  // LDS ring slot of this iteration

  auto publish = [&](Row &x) {
#pragma unroll
    for (int q = 0; q < NSHARE; ++q) {
      MPack<T, VEC> pk;
#pragma unroll
      for (int k = 0; k < VEC; ++k) pk.s[k] = x.share[q][k];
      s_coef[cur][PRIV ? q : wv + 4 * q][lane] = pk;
    }
  };

  auto step = [&](Row &x, int r) {
    if (!PRIV) {
      publish(x);
      __syncthreads();
    }

    T cu[S + 1][VEC], cv[S + 1][VEC];  // newest row of every level this iteration
    T dcu[S + 1][VEC], dcv[S + 1][VEC];  // (backward evaluation) and of d = b_k + b_{k+1}
    FB nau[S + 1][VEC], nav[S + 1][VEC];
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
      T K[8][VEC];
      if (PRIV && j == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) K[q][k] = x.share[PRIV ? q : 0][k];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const MPack<T, VEC> pq = s_coef[sl][q][lane];
#pragma unroll
          for (int k = 0; k < VEC; ++k) K[q][k] = pq.s[k];
        }
      }
      T su[VEC], sv[VEC], lu[VEC], lv[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) { su[k] = b2san(cu[j - 1][k]); sv[k] = b2san(cv[j - 1][k]); }
      L[j - 1].template feed<CLEN>(su, sv, K, lu, lv);
      const double pkj = P.pk[j - 1];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T xu = o1u[j - 1][k], xv = o1v[j - 1][k];
        const T avu = -xu - c * lu[k], avv = -xv - c * lv[k];
        if constexpr (CLEN) {
          // Reinsch's form (round 5, see the header): d_{k+1} of this row came out of level j - 1 one iteration ago
          const T dpu = (j == 1) ? dscale * x.up[k] : d1u[j - 1][k];   // (dscale = 1 but for the first launch: exact)
          const T dpv = (j == 1) ? dscale * x.vp[k] : d1v[j - 1][k];
          const T fiu = (T)((j == 1) ? x.fu[k] : accu[j - 1][k]);
          const T fiv = (T)((j == 1) ? x.fv[k] : accv[j - 1][k]);
          const bool fin = last && j == S;       // the last level of the last launch is the result: p_0 f - c L(b_1) - d_1
          const T mtc = fin ? -c : T(-2) * c;
          const T dku = rfma((T)pkj, fiu, rfma(mtc, lu[k], -dpu));
          const T dkv = rfma((T)pkj, fiv, rfma(mtc, lv[k], -dpv));
          dcu[j][k] = dku;
          dcv[j][k] = dkv;
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
          mstore<T, VEC>((j == S ? P.u2o : P.u1o) + off, cu[j]);
          mstore<T, VEC>((j == S ? P.v2o : P.v1o) + off, cv[j]);
          if constexpr (CLEN) {   // both state planes from level S: b_k and d_k = b_k + b_{k+1}
            mstore<T, VEC>(P.u1o + off, dcu[j]);
            mstore<T, VEC>(P.v1o + off, dcv[j]);
          }
        }
        if (j == S && !CLEN) {
          mstore<FB, VEC>(P.fu_out + off, nau[j]);
          mstore<FB, VEC>(P.fv_out + off, nav[j]);
        }
        if (CLEN && j == S && last) {
          if (P.du_out) {  // wave-uniform: f64 result from f32 state
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

#pragma unroll
    for (int j = 0; j < S; ++j) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        o2u[j][k] = o1u[j][k];
        o2v[j][k] = o1v[j][k];
        o1u[j][k] = cu[j][k];
        o1v[j][k] = cv[j][k];
        if (j >= 1) { accu[j][k] = nau[j][k]; accv[j][k] = nav[j][k]; }
        if (CLEN && j >= 1) { d1u[j][k] = dcu[j][k]; d1v[j][k] = dcv[j][k]; }
      }
    }
    if (PRIV) publish(x);  // after level S has read the slot this overwrites (same wave: LDS executes in order)
    cur = (cur + 1 == NS) ? 0 : cur + 1;
  };

  const int r_begin = a - S, r_end = b + S;  // rows delivered: [a-S, b+S-1]
  if (D == 1) {
    Row nxt;
    load_row(nxt, r_begin);
    for (int r = r_begin; r < r_end; ++r) {
      Row now = nxt;
      load_row(nxt, min(r + 1, r_end - 1));
      step(now, r);
    }
  } else {
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
__global__ __launch_bounds__((PRIV ? 64 : 256), (sizeof(T) == 8 && (S > 3 || (PRIV && S > 2)) ? 1 : 2)) void k_bgrid_stream2(const BStream2P<T, FB> P) {
  bgrid_stream2_body<T, FB, VEC, S, D, PRIV, false>(P);
}
template <typename T, int VEC, int S, int D, bool PRIV>
__global__ __launch_bounds__((PRIV ? 64 : 256), (sizeof(T) == 8 && (S > 3 || (PRIV && S > 2)) ? 1 : 2)) void k_bgrid_stream2c(const BStream2P<T, T> P) {
  bgrid_stream2_body<T, T, VEC, S, D, PRIV, true>(P);
}

static bool b2al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool bgrid_multi_supported(const gcmf_plan *pl, int64_t nbatch, int S) {
  if (pl->kind != K_BGRID || nbatch < 1) return false;
  if (S < 2 || S > ((pl->d.dtype == GCMF_F64 || nbatch == 1) ? 4 : 6)) return false;  // see cgrid_multi_supported
  const int vec = pl->d.dtype == GCMF_F64 ? 2 : 4;
  if (pl->g.nx % vec || pl->g.nx < vec || pl->g.rows < S + 2) return false;
  for (int k = 0; k < 8; ++k)
    if (!b2al16(pl->g.coef[k])) return false;
  return true;
}

template <typename T, typename FB, int VEC, int S, int D, bool PRIV, bool CLEN = false> static int launch_b2(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  constexpr int M = (S + VEC - 1) / VEC * VEC, W = 64 * VEC, WI = W - 2 * M;
  const Geom &g = pl->g;
  BStream2P<T, FB> P;
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
  for (int k = 0; k < 8; ++k) P.coef[k] = (const T *)g.coef[k];
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
  if (H <= 0) {  // see launch_c2 (gcmf_cgrid_stream2.hip)
    long long cap = (sizeof(T) == 8 && S > 3) ? 1024 : 2048;  // f64 at four levels: one wave per SIMD
    if (PRIV) {  // one-wave workgroups: registers or the LDS ring bound the residency
      const long long by_lds = (160 * 1024) / ((long long)(S - 1) * 8 * 64 * sizeof(MPack<T, VEC>));
      const long long by_reg = (sizeof(T) == 8 && S > 2) ? 4 : 8;
      cap = 256 * (by_lds < by_reg ? by_lds : by_reg) * 85 / 100;
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
  const size_t lds = (size_t)(PRIV ? S - 1 : S + 1) * 8 * 64 * sizeof(MPack<T, VEC>);
  if constexpr (CLEN) {
    hipLaunchKernelGGL((k_bgrid_stream2c<T, VEC, S, D, PRIV>), grid, block, lds, s, P);
    note_kernel(pl, std::string("gcmf::k_bgrid_stream2c<") + tyname<T>() + ", " + std::to_string(VEC) + ", " + std::to_string(S) + ", " +
                        std::to_string(D) + ", " + (PRIV ? "true" : "false") + ">", S);
    GCMF_HIP(hipGetLastError());
    return GCMF_OK;
  }
  hipLaunchKernelGGL((k_bgrid_stream2<T, FB, VEC, S, D, PRIV>), grid, block, lds, s, P);
  note_kernel(pl, std::string("gcmf::k_bgrid_stream2<") + tyname<T>() + ", " + tyname<FB>() + ", " + std::to_string(VEC) + ", " +
                      std::to_string(S) + ", " + std::to_string(D) + ", " + (PRIV ? "true" : "false") + ">", S);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

template <typename T, typename FB, int S, int DMAX> static int launch_b2_sel(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  // single-level fields: private coefficient rings (no shadow waves); env GCMF_VEC_PRIV=0 keeps the padded lock-step form
  static const bool priv_ok = !(getenv("GCMF_VEC_PRIV") && atoi(getenv("GCMF_VEC_PRIV")) == 0);
  if (a.nbatch == 1 && priv_ok) return launch_b2<T, FB, 2, S, 1, true>(pl, a, s);
  if (DMAX == 1 || pl->prefetch_rows == 1) return launch_b2<T, FB, 2, S, 1, false>(pl, a, s);
  return launch_b2<T, FB, 2, S, DMAX, false>(pl, a, s);
}

int launch_bgrid_multi(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  if (a.clen) {  // backward evaluation (opt-in): the conveyor has the state's type; single-level fields: private coefficient rings
    static const bool priv_ok = !(getenv("GCMF_VEC_PRIV") && atoi(getenv("GCMF_VEC_PRIV")) == 0);
    const bool priv = a.nbatch == 1 && priv_ok;
    if (pl->d.dtype == GCMF_F64) {
      switch (a.S) {
        case 2: return priv ? launch_b2<double, double, 2, 2, 1, true, true>(pl, a, s) : launch_b2<double, double, 2, 2, 2, false, true>(pl, a, s);
        case 3: return priv ? launch_b2<double, double, 2, 3, 1, true, true>(pl, a, s) : launch_b2<double, double, 2, 3, 1, false, true>(pl, a, s);
        case 4: return priv ? launch_b2<double, double, 2, 4, 1, true, true>(pl, a, s) : launch_b2<double, double, 2, 4, 2, false, true>(pl, a, s);
      }
      return GCMF_ERR_INVALID_ARG;
    }
    switch (a.S) {
      case 2: return priv ? launch_b2<float, float, 2, 2, 1, true, true>(pl, a, s) : launch_b2<float, float, 2, 2, 2, false, true>(pl, a, s);
      case 3: return priv ? launch_b2<float, float, 2, 3, 1, true, true>(pl, a, s) : launch_b2<float, float, 2, 3, 2, false, true>(pl, a, s);
      case 4: return priv ? launch_b2<float, float, 2, 4, 1, true, true>(pl, a, s) : launch_b2<float, float, 2, 4, 2, false, true>(pl, a, s);
    }
    return GCMF_ERR_INVALID_ARG;
  }
  if (pl->d.dtype == GCMF_F64) {
    switch (a.S) {
      case 2: return launch_b2_sel<double, double, 2, 2>(pl, a, s);
      case 3: return launch_b2_sel<double, double, 3, 1>(pl, a, s);
      case 4: return launch_b2_sel<double, double, 4, 2>(pl, a, s);  // one wave per SIMD: registers to spare
    }
    return GCMF_ERR_INVALID_ARG;
  }
  switch (a.S * 2 + (a.fb_is_f32 ? 1 : 0)) {
    case 4: return launch_b2_sel<float, double, 2, 2>(pl, a, s);
    case 5: return launch_b2_sel<float, float, 2, 2>(pl, a, s);
    case 6: return launch_b2_sel<float, double, 3, 2>(pl, a, s);
    case 7: return launch_b2_sel<float, float, 3, 2>(pl, a, s);
    case 8: return launch_b2_sel<float, double, 4, 2>(pl, a, s);
    case 9: return launch_b2_sel<float, float, 4, 2>(pl, a, s);
    case 10: return launch_b2_sel<float, double, 5, 1>(pl, a, s);
    case 11: return launch_b2_sel<float, float, 5, 1>(pl, a, s);
    case 12: return launch_b2_sel<float, double, 6, 1>(pl, a, s);
    case 13: return launch_b2_sel<float, float, 6, 1>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
