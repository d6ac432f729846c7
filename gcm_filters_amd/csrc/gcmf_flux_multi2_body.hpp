// The general (NaN / inf aware) two-rows-per-iteration march of the deep flux-form kernels as a device function:
// k_flux_multi2 (gcmf_flux_multi2.hip) is just this; k_ring (gcmf_ring_impl.hpp) falls back to it for a strip in
// which a non-finite value turned up.  See gcmf_flux_multi2.hip for the description.
#pragma once
#include "gcmf_multi_common.hpp"

namespace gcmf {

template <typename T, typename FB, int S, int VEC = 16 / (int)sizeof(T)>   // (VEC: cells per lane = the caller's window geometry)
__device__ __forceinline__ void flux_multi2_march(const MultiP<T, FB> &P, const int wid) {
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;

  const int lane = threadIdx.x & 63;
  // wid: the wave's (window, strip) index, wave-uniform (scalar row / pointer arithmetic); the caller decides how workgroups map
  // to strips (k_ring reorders them per XCD)
  if (wid >= P.nwaves) return;
  const int wx = wid % P.nwx, st = wid / P.nwx;
  const int nx = P.nx, rows = P.rows;
  const int a = P.out_lo + st * P.H;
  const int b = min(a + P.H, P.out_hi);
  const long long boff = (long long)blockIdx.y * P.bstride;
  const int pos = wx * WI - M + lane * VEC;
  int col = pos % nx;
  if (col < 0) col += nx;
  const bool keep = (lane * VEC >= M) && (lane * VEC < W - M) && (pos < nx);
  const T c = (T)P.c;
  const bool first = P.first, last = P.last;

  // ---- register-resident state; q = 0 / 1 is the first / second row of the pair (rows r, r+1) ----
  T G[S][4][VEC];     // level t (0..S-1) output rows: sub-iteration q reads slots (q, q+1, q+2) = (old, mid, new)
  unsigned Rf[S];     // flags (what nan_to_num removed) of slots 0 and 1; nf[q][t]: of the row level t produced in q
  unsigned nf[2][S];
  // lag lines, indexed by the lag relative to row r+1: in sub-iteration q level t works on lag t + 1 - q
  T cEq[S + 2][VEC], cNq[S + 3][VEC], raq[S + 2][VEC];
  FB Fq[S + 2][VEC];
#pragma unroll
  for (int t = 0; t < S; ++t) {
    Rf[t] = nf[0][t] = nf[1][t] = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) G[t][0][k] = G[t][1][k] = G[t][2][k] = G[t][3][k] = T(0);
  }
#pragma unroll
  for (int l = 0; l < S + 3; ++l) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      cNq[l][k] = T(0);
      if (l < S + 2) { cEq[l][k] = T(0); raq[l][k] = T(0); Fq[l][k] = FB(0); }
    }
  }

  struct Row {
    T u[VEC], v[VEC], ce[VEC], cn[VEC], ra[VEC];
    FB fb[VEC];
    bool closed;
  };
  auto row_index = [&](int r, bool &outside) {
    int jr = r;
    outside = false;
    if (P.wrap) {
      jr = r < 0 ? r + rows : (r >= rows ? r - rows : r);
    } else if (r < 0 || r >= rows) {
      outside = true;
      jr = r < 0 ? 0 : rows - 1;
    }
    return jr;
  };
  // row r of T_{k-1} travels with the centre-only operands of row r-1 (T_{k-2}, fbar, coefficients)
  auto load_row = [&](Row &x, int r) {
    bool out_u, out_c;
    const long long ro = (long long)row_index(r, out_u) * nx + col;
    const long long rc = (long long)row_index(r - 1, out_c) * nx + col;
    mload<T, VEC>(x.u, P.u0 + boff + ro);
    if (!first) {
      mload<T, VEC>(x.v, P.v0 + boff + rc);
      mload<FB, VEC>(x.fb, P.fb_in + boff + rc);
    }
    mload<T, VEC>(x.ce, P.cE + rc);
    mload<T, VEC>(x.cn, P.cN + rc);
    mload<T, VEC>(x.ra, P.ra + rc);
    x.closed = out_c;
  };

  constexpr unsigned OLD_MASK = (1u << (2 * VEC)) - 1u;
  T Vp[VEC];                  // raw T_{k-2} of the row level 1 works on in this sub-iteration
  T out_v[VEC], out_u[VEC];   // raw outputs of levels S-1 and S of this sub-iteration

  auto consume = [&](auto qq, const Row &cur) {
    constexpr int q = decltype(qq)::value;
    nf[q][0] = 0u;
    bool odd = false;
#pragma unroll
    for (int k = 0; k < VEC; ++k) odd = odd || !(mabs(cur.u[k]) <= MLim<T>::big());
    if (__any(odd)) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        unsigned f;
        G[0][2 + q][k] = msan_flag(cur.u[k], f);
        nf[q][0] |= f << (2 * k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) G[0][2 + q][k] = cur.u[k];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      cEq[2 - q][k] = cur.closed ? T(0) : cur.ce[k];
      cNq[2 - q][k] = cur.closed ? T(0) : cur.cn[k];
      raq[2 - q][k] = cur.closed ? T(0) : cur.ra[k];
      Fq[2 - q][k] = first ? FB(0) : cur.fb[k];
      Vp[k] = first ? T(0) : cur.v[k];
    }
  };

  auto level = [&](auto tt, auto qq, auto flagged_c) {
    constexpr int t = decltype(tt)::value;
    constexpr int q = decltype(qq)::value;
    constexpr bool FLAGGED = decltype(flagged_c)::value;
    constexpr int e = t + 1 - q;  // lag of this level's row
    const T(&gS)[VEC] = G[t - 1][q];
    const T(&gC)[VEC] = G[t - 1][q + 1];
    const T(&gN)[VEC] = G[t - 1][q + 2];
    const T ev = from_upper_lane0(gC[0]);
    T fev[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T xE = (k == VEC - 1) ? ev : gC[k < VEC - 1 ? k + 1 : k];
      fev[k] = (xE - gC[k]) * cEq[e][k];
    }
    const T few = from_lower_lane0(fev[VEC - 1]);
    const unsigned fmid = (q == 0) ? (Rf[t - 1] >> (2 * VEC)) : nf[0][t - 1];
    const unsigned fold2 = (t >= 2) ? ((q == 0) ? (Rf[t >= 2 ? t - 2 : 0] & OLD_MASK) : (Rf[t >= 2 ? t - 2 : 0] >> (2 * VEC))) : 0u;
    T tkv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T xC = gC[k];
      const T fe = fev[k];
      const T fw = (k == 0) ? few : fev[k > 0 ? k - 1 : 0];
      const T fn = (gN[k] - xC) * cNq[e][k];
      const T fs = (xC - gS[k]) * cNq[e + 1][k];
      const T L = ((fe - fw) + (fn - fs)) * raq[e][k];
      const T x = FLAGGED ? unsan(xC, (fmid >> (2 * k)) & 3u) : xC;
      const T av = cheb_a<true>(x, c, L);
      T tk;
      if (t == 1 && first) {
        tk = av;
        Fq[e][k] = cheb_acc_first<true, T, FB>(P.p0, P.pk[0], x, av);
      } else {
        T x2;
        if (t == 1) x2 = Vp[k];
        else x2 = FLAGGED ? unsan(G[t >= 2 ? t - 2 : 0][q][k], (fold2 >> (2 * k)) & 3u) : G[t >= 2 ? t - 2 : 0][q][k];
        tk = cheb_t<true>(av, x2);
        Fq[e][k] = cheb_acc<true, T, FB>(Fq[e][k], P.pk[t - 1], tk);
      }
      tkv[k] = tk;
      if (t == S - 1) out_v[k] = tk;
      if (t == S) out_u[k] = tk;
    }
    if (t < S) {
      unsigned f2 = 0u;
      bool odd = false;
#pragma unroll
      for (int k = 0; k < VEC; ++k) odd = odd || !(mabs(tkv[k]) <= MLim<T>::big());
      if (__any(odd)) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          unsigned f;
          G[t < S ? t : 0][q + 2][k] = msan_flag(tkv[k], f);
          f2 |= f << (2 * k);
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) G[t < S ? t : 0][q + 2][k] = tkv[k];
      }
      nf[q][t < S ? t : 0] = f2;
    }
  };

  auto level_all = [&](auto qq, auto flagged_c) {
#define GCMF_LEVEL2(t_)                                                              \
  if constexpr (S >= (t_)) level(std::integral_constant<int, (t_)>{}, qq, flagged_c);
    GCMF_LEVEL2(1) GCMF_LEVEL2(2) GCMF_LEVEL2(3) GCMF_LEVEL2(4) GCMF_LEVEL2(5) GCMF_LEVEL2(6) GCMF_LEVEL2(7) GCMF_LEVEL2(8)
#undef GCMF_LEVEL2
  };

  auto compute = [&](auto qq, int rq) {
    constexpr int q = decltype(qq)::value;
    unsigned anyf = nf[q][0];
#pragma unroll
    for (int t = 0; t < S; ++t) anyf |= Rf[t] | (q == 1 ? nf[0][t] : 0u);
    if (__any(anyf != 0u)) level_all(qq, std::true_type{});
    else level_all(qq, std::false_type{});
    // stores: T_{k-1+S} row rq-S, T_{k-2+S} row rq-S+1, fbar row rq-S
    const int ju = rq - S;
    if (keep && ju >= a && ju < b) {
      const long long off = boff + (long long)ju * nx + col;
      if (!last) mstore<T, VEC>(P.uo + off, out_u);
      mstore<FB, VEC>(P.fb_out + off, Fq[S + 1 - q]);
    }
    const int jv = rq - S + 1;
    if (!last && keep && jv >= a && jv < b) mstore<T, VEC>(P.vo + boff + (long long)jv * nx + col, out_v);
  };

  // ---- march north, two rows per iteration, two rows of operands in flight ----
  const int r_begin = a - S, r_last = b + S - 1;           // rows loaded by this strip: [a-S, b+S)
  const int npair = (r_last - r_begin + 2) / 2;            // an odd row count marches one row further (nothing stored)
  Row qa, qb;
  load_row(qa, r_begin);
  load_row(qb, min(r_begin + 1, r_last));
  for (int ip = 0, r = r_begin; ip < npair; ++ip, r += 2) {
    consume(std::integral_constant<int, 0>{}, qa);
    load_row(qa, min(r + 2, r_last));
    compute(std::integral_constant<int, 0>{}, r);
    consume(std::integral_constant<int, 1>{}, qb);
    load_row(qb, min(r + 3, r_last));
    compute(std::integral_constant<int, 1>{}, r + 1);
    // shift everything by two rows
#pragma unroll
    for (int t = 0; t < S; ++t) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        G[t][0][k] = G[t][2][k];
        G[t][1][k] = G[t][3][k];
      }
      Rf[t] = (nf[0][t] & OLD_MASK) | (nf[1][t] << (2 * VEC));
    }
#pragma unroll
    for (int l = S + 2; l >= 3; --l) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        cNq[l][k] = cNq[l - 2][k];
        if (l < S + 2) {
          cEq[l][k] = cEq[l - 2][k];
          raq[l][k] = raq[l - 2][k];
          Fq[l][k] = Fq[l - 2][k];
        }
      }
    }
  }
}


}  // namespace gcmf
