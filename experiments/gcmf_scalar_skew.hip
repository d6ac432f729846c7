// Temporally blocked scalar kernel, "double-skew" variant: like k_scalar_multi (gcmf_scalar_multi.hip), but level t
// trails level t-1 by TWO rows instead of one:
//
//     iteration r (row r of T_{k-1} delivered):   level t produces row r - 2t + 1   (t = 1..S)
//
// so level t only reads rows that level t-1 finished in EARLIER iterations.  Inside one iteration the S levels are
// therefore independent instruction streams: with one wave per SIMD (the register budget of S >= 6) the in-order
// issue no longer waits on the S-long dependent chain DPP -> flux -> A -> T of the single-skew kernel, and the
// finite-value test needs ONE wave-uniform branch per iteration instead of one per level.  The price is longer
// lag windows (coefficients and fbar by lag 1..2S, 4-row windows per level) and S-1 extra drain iterations per strip.
// Arithmetic per level is unchanged, results stay bit-identical to S single steps.
#include "gcmf_multi_common.hpp"

namespace gcmf {

template <typename T, typename FB, int KIND, int S, int D>
__global__ __launch_bounds__(256, 1) void k_scalar_skew(const MultiP<T, FB> P) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;
  constexpr bool SAN = (KIND != K_REG);
  constexpr int NL = 2 * S;  // largest lag kept (south-face coefficient of level S)

  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
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

  // ---- register-resident state ----
  // G[t] (t = 0..S-1): rows r-2t-3, r-2t-2, r-2t-1, r-2t of level t as the stencil sees them.  Its consumer,
  // level t+1, uses slots 1..3 as south / centre / north; level t+2 reads slot 0 as its T_{k-2} operand.
  T G[S][4][VEC];
  unsigned Rf[S];  // 2 flag bits per (slot, cell): what nan_to_num removed (to rebuild raw values)
  T Vp[VEC];       // raw T_{k-2} of row r-1 (level 1's T_{k-2} operand)
  T cEq[NL + 1][VEC], cNq[NL + 1][VEC], raq[NL + 1][VEC];  // coefficient rows by lag 1..2S (row r - lag)
  unsigned Bq[NL + 1];
  FB Fq[NL + 1][VEC];  // fbar accumulators by lag 1..2S-1
#pragma unroll
  for (int t = 0; t < S; ++t) {
    Rf[t] = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) G[t][q][k] = T(0);
    }
  }
#pragma unroll
  for (int l = 0; l <= NL; ++l) {
    Bq[l] = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) { cEq[l][k] = T(0); cNq[l][k] = T(0); raq[l][k] = T(0); Fq[l][k] = FB(0); }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) Vp[k] = T(0);

  struct Row {
    T u[VEC], v[VEC], ce[VEC], cn[VEC], ra[VEC], ar[VEC];
    FB fb[VEC];
    unsigned bits;
    bool closed;
  };
  const int last_needed = b + S - 1;  // rows beyond it cannot influence rows < b of level S: re-load it instead
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
  auto load_row = [&](Row &x, int r) {
    r = min(r, last_needed);
    bool out_u, out_c;
    const long long ro = (long long)row_index(r, out_u) * nx + col;
    const long long rc = (long long)row_index(r - 1, out_c) * nx + col;
    mload<T, VEC>(x.u, P.u0 + boff + ro);
    if (!first) {
      mload<T, VEC>(x.v, P.v0 + boff + rc);
      mload<FB, VEC>(x.fb, P.fb_in + boff + rc);
    }
    if (KIND == K_FLUX) {
      mload<T, VEC>(x.ce, P.cE + rc);
      mload<T, VEC>(x.cn, P.cN + rc);
      mload<T, VEC>(x.ra, P.ra + rc);
      x.closed = out_c;
    }
    if (KIND == K_MASK) {
      unsigned bb = 0;
      const uint8_t *mp = P.mbits + rc;
      if (VEC == 2) bb = *reinterpret_cast<const unsigned short *>(mp);
      else bb = *reinterpret_cast<const unsigned *>(mp);
      x.bits = out_c ? 0u : bb;
    }
    if (first && P.area_weighted) mload<T, VEC>(x.ar, P.area + ro);
  };

  // flags: bits [2*(q*VEC + k), +1] of Rf[t] belong to slot q, cell k
  auto slot_flags = [&](int t, int q, int k) { return (Rf[t] >> (2 * (q * VEC + k))) & 3u; };

  // ---- part 1: deliver row r: level-0 window shifts by one row, lag-1 slots receive the centre-only operands ----
  auto consume = [&](const Row &cur) {
    T uu[VEC];
    bool odd = false;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      uu[k] = cur.u[k];
      if (first && P.area_weighted) uu[k] = uu[k] * cur.ar[k];
      odd = odd || !(mabs(uu[k]) <= MLim<T>::big());
    }
    unsigned nf = 0u;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) G[0][q][k] = G[0][q + 1][k];
    }
    if (SAN && __any(odd)) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        unsigned f;
        G[0][3][k] = msan_flag(uu[k], f);
        nf |= f << (2 * k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) G[0][3][k] = uu[k];
    }
    Rf[0] = (Rf[0] >> (2 * VEC)) | (nf << (2 * 3 * VEC));
    if (KIND == K_FLUX) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        cEq[1][k] = cur.closed ? T(0) : cur.ce[k];
        cNq[1][k] = cur.closed ? T(0) : cur.cn[k];
        raq[1][k] = cur.closed ? T(0) : cur.ra[k];
      }
    }
    if (KIND == K_MASK) Bq[1] = cur.bits;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      Fq[1][k] = first ? FB(0) : cur.fb[k];
      Vp[k] = first ? T(0) : cur.v[k];
    }
  };

  T tk[S + 1][VEC];  // raw value produced by level t in this iteration (row r-2t+1)

  auto level = [&](auto tt, auto flagged_c) {
    constexpr int t = decltype(tt)::value;
    constexpr bool FLAGGED = decltype(flagged_c)::value;
    constexpr int lag = 2 * t - 1;
    const T(&gS)[VEC] = G[t - 1][1];
    const T(&gC)[VEC] = G[t - 1][2];
    const T(&gN)[VEC] = G[t - 1][3];
    const T wv = from_lower_lane(gC[VEC - 1]);
    const T ev = from_upper_lane(gC[0]);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T xC = gC[k];
      const T xW = (k == 0) ? wv : gC[k > 0 ? k - 1 : 0];
      const T xE = (k == VEC - 1) ? ev : gC[k < VEC - 1 ? k + 1 : k];
      T L;
      if (KIND == K_REG) {
        L = T(-4) * xC + xE;
        L = L + xW;
        L = L + gN[k];
        L = L + gS[k];
      } else if (KIND == K_MASK) {
        const unsigned bb = (Bq[lag] >> (8 * k)) & 0xFFu;
        const T mC = (bb & 1u) ? xC : T(0);
        const T wf = (T)__popc((bb >> 1) & 0xFu);
        L = -wf * mC + ((bb & 2u) ? xE : T(0));
        L = L + ((bb & 4u) ? xW : T(0));
        L = L + ((bb & 8u) ? gN[k] : T(0));
        L = L + ((bb & 16u) ? gS[k] : T(0));
        L = (bb & 1u) ? L : T(0);
      } else {
        const T cw = (k == 0) ? from_lower_lane(cEq[lag][VEC - 1]) : cEq[lag][k > 0 ? k - 1 : 0];
        const T fe = (xE - xC) * cEq[lag][k];
        const T fw = (xC - xW) * cw;
        const T fn = (gN[k] - xC) * cNq[lag][k];
        const T fs = (xC - gS[k]) * cNq[lag + 1][k];
        L = (fe - fw + fn - fs) * raq[lag][k];
      }
      const T x = FLAGGED ? unsan(xC, slot_flags(t - 1, 2, k)) : xC;
      const T av = -x - c * L;
      T v;
      if (t == 1 && first) {
        v = av;
        if (std::is_same<FB, T>::value) Fq[1][k] = (FB)((T)P.p0 * x + (T)P.pk[0] * av);
        else Fq[1][k] = (FB)(P.p0 * (double)x + P.pk[0] * (double)av);
      } else {
        T x2;
        if (t == 1) x2 = Vp[k];
        else x2 = FLAGGED ? unsan(G[t >= 2 ? t - 2 : 0][0][k], slot_flags(t >= 2 ? t - 2 : 0, 0, k)) : G[t >= 2 ? t - 2 : 0][0][k];
        v = T(2) * av - x2;
        if (std::is_same<FB, T>::value) Fq[lag][k] = Fq[lag][k] + (FB)((T)P.pk[t - 1] * v);
        else Fq[lag][k] = Fq[lag][k] + (FB)(P.pk[t - 1] * (double)v);
      }
      tk[t][k] = v;
    }
  };
  auto level_all = [&](auto flagged_c) {
    level(std::integral_constant<int, 1>{}, flagged_c);
    if constexpr (S >= 2) level(std::integral_constant<int, 2>{}, flagged_c);
    if constexpr (S >= 3) level(std::integral_constant<int, 3>{}, flagged_c);
    if constexpr (S >= 4) level(std::integral_constant<int, 4>{}, flagged_c);
    if constexpr (S >= 5) level(std::integral_constant<int, 5>{}, flagged_c);
    if constexpr (S >= 6) level(std::integral_constant<int, 6>{}, flagged_c);
    if constexpr (S >= 7) level(std::integral_constant<int, 7>{}, flagged_c);
    if constexpr (S >= 8) level(std::integral_constant<int, 8>{}, flagged_c);
  };

  // ---- part 2: all levels (independent of each other), stores, window rotation ----
  auto compute = [&](int r) {
    unsigned anyf = 0u;
#pragma unroll
    for (int t = 0; t < S; ++t) anyf |= Rf[t];
    if (SAN && __any(anyf != 0u)) level_all(std::true_type{});
    else level_all(std::false_type{});

    {  // stores: T_{k-1+S} row r-2S+1, T_{k-2+S} row r-2S+3, fbar row r-2S+1
      const int ju = r - 2 * S + 1;
      if (keep && ju >= a && ju < b) {
        const long long off = boff + (long long)ju * nx + col;
        if (!last) {
          mstore<T, VEC>(P.uo + off, tk[S]);
        } else if (P.area_weighted) {
          T ar[VEC];
          mload<T, VEC>(ar, P.area + (long long)ju * nx + col);
#pragma unroll
          for (int k = 0; k < VEC; ++k) Fq[2 * S - 1][k] = Fq[2 * S - 1][k] / (FB)ar[k];
        }
        mstore<FB, VEC>(P.fb_out + off, Fq[2 * S - 1]);
      }
      const int jv = r - 2 * S + 3;
      if (!last && S >= 2 && keep && jv >= a && jv < b) mstore<T, VEC>(P.vo + boff + (long long)jv * nx + col, tk[S >= 2 ? S - 1 : 1]);
    }

    // rotate: every level's window takes its new row (sanitised only if some lane actually holds a NaN/inf)
    bool odd = false;
#pragma unroll
    for (int t = 1; t < S; ++t) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) odd = odd || !(mabs(tk[t][k]) <= MLim<T>::big());
    }
    const bool dirty = SAN && __any(odd);
#pragma unroll
    for (int t = 1; t < S; ++t) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) G[t][q][k] = G[t][q + 1][k];
      }
      unsigned nf = 0u;
      if (dirty) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          unsigned f;
          G[t][3][k] = msan_flag(tk[t][k], f);
          nf |= f << (2 * k);
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) G[t][3][k] = tk[t][k];
      }
      Rf[t] = (Rf[t] >> (2 * VEC)) | (nf << (2 * 3 * VEC));
    }
#pragma unroll
    for (int l = NL; l >= 2; --l) {
      Bq[l] = Bq[l - 1];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        cEq[l][k] = cEq[l - 1][k];
        cNq[l][k] = cNq[l - 1][k];
        raq[l][k] = raq[l - 1][k];
        Fq[l][k] = Fq[l - 1][k];
      }
    }
  };

  // ---- march north ----
  const int r_begin = a - S, r_end = b + 2 * S - 1;
  Row q0, q1, q2;
  load_row(q0, r_begin);
  if (D >= 2) load_row(q1, min(r_begin + 1, r_end - 1));
  if (D >= 3) load_row(q2, min(r_begin + 2, r_end - 1));
#define GCMF_SLOT(Q, dd)                                          \
  if (r + (dd) < r_end) {                                         \
    consume(Q);                                                   \
    load_row(Q, min(r + (dd) + D, r_end - 1));                    \
    compute(r + (dd));                                            \
  }
  for (int r = r_begin; r < r_end; r += D) {
    GCMF_SLOT(q0, 0)
    if (D >= 2) { GCMF_SLOT(q1, 1) }
    if (D >= 3) { GCMF_SLOT(q2, 2) }
  }
#undef GCMF_SLOT
}

// ------------------------------------------------------------------------------------------------------
template <typename T, typename FB, int KIND, int S, int D>
static int launch_skew_s(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;
  const Geom &g = pl->g;
  MultiP<T, FB> P;
  P.u0 = (const T *)a.u0;
  P.v0 = (const T *)a.v0;
  P.uo = (T *)a.uo;
  P.vo = (T *)a.vo;
  P.fb_in = (const FB *)a.fb_in;
  P.fb_out = (FB *)a.fb_out;
  P.cE = (const T *)g.coef[0];
  P.cN = (const T *)g.coef[1];
  P.ra = (const T *)g.coef[2];
  P.mbits = g.mbits;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  int H = pl->strip_rows;
  if (H <= 0) {  // one resident round of waves, one wave per SIMD
    long long want = 1024 / ((long long)P.nwx * a.nbatch);
    if (want < 1) want = 1;
    H = (int)((nrows + want - 1) / want);
    if (H < 3 * S) H = 3 * S;
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.nstrips = (nrows + H - 1) / H;
  P.nwaves = P.nwx * P.nstrips;
  P.wrap = g.south_wrap && g.north_wrap;
  P.first = a.first;
  P.last = a.last;
  P.area_weighted = g.area_weighted;
  P.bstride = (long long)g.rows * g.nx;
  for (int t = 0; t < MAX_S; ++t) P.pk[t] = t < S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  dim3 block(256), grid((P.nwaves + 3) / 4, (unsigned)a.nbatch);
  hipLaunchKernelGGL((k_scalar_skew<T, FB, KIND, S, D>), grid, block, 0, s, P);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

template <typename T, typename FB, int KIND> static int launch_skew_k(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 4: return launch_skew_s<T, FB, KIND, 4, 2>(pl, a, s);
    case 6: return launch_skew_s<T, FB, KIND, 6, 2>(pl, a, s);
  }
  set_error("launch_scalar_skew: unsupported S=%d", a.S);
  return GCMF_ERR_INVALID_ARG;
}

template <typename T, typename FB> static int launch_skew_t(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (pl->kind) {
    case K_REG: return launch_skew_k<T, FB, K_REG>(pl, a, s);
    case K_MASK: return launch_skew_k<T, FB, K_MASK>(pl, a, s);
    case K_FLUX: return launch_skew_k<T, FB, K_FLUX>(pl, a, s);
  }
  set_error("launch_scalar_skew: plan is not a scalar kind");
  return GCMF_ERR_INVALID_ARG;
}

bool skew_supported(const gcmf_plan *pl, int S) {
  if (!(S == 4 || S == 6)) return false;
  if (!multi_supported(pl, S)) return false;
  return pl->g.rows >= 2 * S + 2;  // row wrap by one conditional add needs |r| < rows over [a-S-1, b+2S)
}

int launch_scalar_skew(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_skew_t<double, double>(pl, a, s);
  if (a.fb_is_f32) return launch_skew_t<float, float>(pl, a, s);
  return launch_skew_t<float, double>(pl, a, s);
}

}  // namespace gcmf
