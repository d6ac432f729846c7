// Temporally blocked scalar Chebyshev kernels: S recurrence steps per pass over HBM.
//
// The single-step kernel (gcmf_scalar.hip) moves B_alg = 5w + coefficients bytes per cell per step and is
// already at ~0.9 of the achievable HBM rate, so the only way to go faster is to touch HBM less often.
// Here ONE launch advances the recurrence by S steps ("levels") while streaming every plane once:
//
//   * a wave owns a window of 64*VEC contiguous cells in x and MARCHES north through a strip of rows;
//   * time skewing along y: in the iteration that loads row r of T_{k-1}, level t (1..S) produces row r-t of
//     T_{k-1+t} from the 3-row register window of level t-1 -- all levels live in registers, nothing but the
//     final two states and fbar is ever written back;
//   * shrinking window along x: east/west neighbours come from the adjacent lane (wave shuffles); the
//     outermost lanes go stale by one cell per level, so windows overlap by M >= S cells per side and only
//     the interior 64*VEC - 2M cells are stored.  No LDS, no barriers, no inter-wave communication;
//   * strips overlap by S rows per side in y for the same reason (periodic wrap, or ghost rows in the
//     multi-GPU slab case -- this kernel IS the "exchange every S steps" ghost-zone scheme, at wave level).
//
// HBM traffic per cell per step drops from 8w (flux form) to ~(9w / S) * overlap, e.g. 64 B -> ~21 B at S=4.
// Arithmetic per level is identical to the single-step kernel (same operation order, fbar accumulated step
// by step in its storage precision), so results are BIT-IDENTICAL to S single steps.
//
// Reference semantics per level: gcm_filters/filter.py:162-175,192-206 + the Laplacians of kernels.py (see
// gcmf_scalar.hip for the per-kind citations).
#pragma once
#include "gcmf_multi_common.hpp"
#include <cstdlib>

#ifndef GCMF_NO_SKIP
#define GCMF_NO_SKIP 0
#endif

namespace gcmf {

// resident waves per SIMD the register budget allows: the flux form carries 3 coefficient lag windows and fits two
// waves only up to S = 4; the coefficient-free kinds fit two waves at every depth
// K_MASKZ: the land-mask kind when the caller guarantees that land cells of the input states are zero (gcmf_apply and
// the slab driver zero them after the first launch, see gcmf_plan::lbits).  Land then stays zero on its own
// (L is forced to 0 there, 2(-0 - c 0) - 0 = 0), so the stencil needs no per-neighbour wet test: 18 of ~45 VALU
// instructions per cell and level go away.  Same sums on wet cells (a land neighbour contributed 0 before as well).
constexpr int K_MASKZ = 5;
template <int KIND> struct IsMask { static constexpr bool value = (KIND == K_MASK || KIND == K_MASKZ); };

template <typename T, int KIND, int S> struct WavesPerSimd {
  // f32 state carries 4 cells per lane: the flux form spills at two waves per SIMD from S = 3 on (328 B of scratch at
  // S = 4 made a 4-step remainder launch cost more than an 8-step one), the land-mask form from S = 6 on
  static constexpr int value = (KIND == K_FLUX && (S > 4 || (sizeof(T) == 4 && S > 2))) ? 1
                               : ((IsMask<KIND>::value && S > 5 && sizeof(T) == 4) ? 1 : 2);
};

// the march itself as a device function: k_scalar_multi is just this; the static-ring kernels (gcmf_ring_impl.hpp) fall
// back to it for a strip in which a non-finite value turned up
template <typename T, typename FB, int KIND, int S, int D>
__device__ __forceinline__ void scalar_multi_march(const MultiP<T, FB> &P, const int wid) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;  // x margin, multiple of VEC so that windows stay 16-byte aligned
  constexpr int WI = W - 2 * M;
  constexpr bool SAN = (KIND != K_REG);

  const int lane = threadIdx.x & 63;
  // wid: the wave's (window, strip) index, wave-uniform (scalar row / pointer arithmetic); see flux_multi2_march
  if (wid >= P.nwaves) return;
  const int wx = wid % P.nwx, st = wid / P.nwx;
  const int nx = P.nx, rows = P.rows;
  const int a = P.out_lo + st * P.H;
  const int b = min(a + P.H, P.out_hi);
  const long long boff = (long long)blockIdx.y * P.bstride;

  // this lane's VEC columns (periodic in x) and whether it stores
  const int pos = wx * WI - M + lane * VEC;  // unwrapped position of the lane's first cell
  int col = pos % nx;
  if (col < 0) col += nx;
  const bool keep = (lane * VEC >= M) && (lane * VEC < W - M) && (pos < nx);
  const T c = (T)P.c;
  const bool first = P.first, last = P.last;

  // ---- register-resident state ----
  T G[S][3][VEC];      // level t (0..S-1): rows (old, mid, new) as the stencil sees them (nan_to_num'ed)
  unsigned Rf[S];      // 2 flag bits per (slot old/mid, cell) of the same levels: what nan_to_num removed, so
                       // that the raw "-x" and "T_{k-2}" operands (which keep NaN/inf) can be rebuilt
  T Vp[VEC];           // raw T_{k-2} of the row that is `mid` at level 0
  T cEq[S + 1][VEC], cNq[S + 2][VEC], raq[S + 1][VEC];  // coefficient rows by lag >= 1 (row r - lag)
  unsigned Bq[S + 1];  // mask bits by lag
  FB Fq[S + 1][VEC];   // fbar accumulators by lag
#pragma unroll
  for (int t = 0; t < S; ++t) {
    Rf[t] = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) G[t][0][k] = G[t][1][k] = G[t][2][k] = T(0);
  }
#pragma unroll
  for (int l = 0; l <= S; ++l) {
    Bq[l] = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) { cEq[l][k] = T(0); raq[l][k] = T(0); Fq[l][k] = FB(0); }
  }
#pragma unroll
  for (int l = 0; l <= S + 1; ++l) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) cNq[l][k] = T(0);
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) Vp[k] = T(0);

  // one row of operands in flight
  struct Row {
    T u[VEC], v[VEC], ce[VEC], cn[VEC], ra[VEC], ar[VEC];
    FB fb[VEC];
    unsigned bits;
    bool closed;
  };
  // Row r of T_{k-1} travels with the centre-only operands of row r-1 (T_{k-2}, fbar, coefficients, mask bits):
  // level 1 is the first consumer of those and it works on row r-1, so delivering them one row late saves a
  // whole lag-0 register stage.
  auto row_index = [&](int r, bool &outside) {
    int jr = r;
    outside = false;
    if (P.wrap) {  // |r| never leaves (-rows, 2 rows): one conditional add instead of an integer division
      jr = r < 0 ? r + rows : (r >= rows ? r - rows : r);
    } else if (r < 0 || r >= rows) {
      outside = true;
      jr = r < 0 ? 0 : rows - 1;
    }
    return jr;
  };
  auto load_row = [&](Row &x, int r) {
    bool out_u, out_c;
    const long long ro = (long long)row_index(r, out_u) * nx + col;
    const long long rc = (long long)row_index(r - 1, out_c) * nx + col;
    mload<T, VEC>(x.u, P.u0 + boff + ro);
    // unconditional (the first launch reads stand-ins it then ignores): a conditionally loaded 16-byte operand ends
    // up in scratch and is re-read from there every row (seen in the K_REG / K_MASK instantiations)
    // (not in the flux kernel: its instantiations keep these operands in registers as they are, and its schedule is
    // sensitive to any change)
    if (KIND != K_FLUX) {
      mload<T, VEC>(x.v, (first ? P.u0 : P.v0) + boff + rc);
      mload<FB, VEC>(x.fb, (first ? (const FB *)P.fb_out : P.fb_in) + boff + rc);
    } else if (!first) {
      mload<T, VEC>(x.v, P.v0 + boff + rc);
      mload<FB, VEC>(x.fb, P.fb_in + boff + rc);
    }
    if (KIND == K_FLUX) {
      mload<T, VEC>(x.ce, P.cE + rc);
      mload<T, VEC>(x.cn, P.cN + rc);
      mload<T, VEC>(x.ra, P.ra + rc);
      x.closed = out_c;  // beyond a closed boundary: no flux (coefficients zeroed on delivery)
    }
    if (IsMask<KIND>::value) {
      unsigned bb = 0;
      const uint8_t *mp = P.mbits + rc;
      if (VEC == 2) bb = *reinterpret_cast<const unsigned short *>(mp);
      else bb = *reinterpret_cast<const unsigned *>(mp);
      x.bits = out_c ? 0u : bb;
    }
    // same reason: unconditional, from the address of x.u when there is no area to apply (an L1 hit)
    if (KIND != K_FLUX) mload<T, VEC>(x.ar, (first && P.area_weighted) ? P.area + ro : P.u0 + boff + ro);
    else if (first && P.area_weighted) mload<T, VEC>(x.ar, P.area + ro);
  };

  // flag layout in Rf[t]: bits [2k, 2k+1] = cell k of slot `old`, bits [2*VEC + 2k, ..+1] = cell k of slot `mid`
  constexpr unsigned OLD_MASK = (1u << (2 * VEC)) - 1u;

  // ---- one row-iteration, part 1: move the delivered row into the lag-0 / level-0 slots (frees its prefetch
  //      registers so that the next load into them can be issued before the arithmetic starts) ----
  unsigned newflags[S];  // flags of the value each level (0..S-1) produced for its `new` slot
  T out_v[VEC], out_u[VEC];  // raw outputs of levels S-1 and S (the two states written back)
  auto consume = [&](const Row &cur) {
    newflags[0] = 0u;
    T uu[VEC];
    bool odd = false;  // any NaN / inf in this lane's cells?
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      uu[k] = cur.u[k];
      if (first && P.area_weighted) uu[k] = uu[k] * cur.ar[k];  // prepare(): field * area (kernels.py:100-101)
      odd = odd || !(mabs(uu[k]) <= MLim<T>::big());
    }
    // nan_to_num is the identity on finite values: only waves that actually hold a NaN/inf (land cells of a
    // NaN-masked field) pay for the selects and the flag bookkeeping
    if (SAN && __any(odd)) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        unsigned f;
        G[0][2][k] = msan_flag(uu[k], f);
        newflags[0] |= f << (2 * k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) G[0][2][k] = uu[k];
    }
    if (KIND == K_FLUX) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        cEq[1][k] = cur.closed ? T(0) : cur.ce[k];
        cNq[1][k] = cur.closed ? T(0) : cur.cn[k];
        raq[1][k] = cur.closed ? T(0) : cur.ra[k];
      }
    }
    if (IsMask<KIND>::value) Bq[1] = cur.bits;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      Fq[1][k] = first ? FB(0) : cur.fb[k];
      Vp[k] = first ? T(0) : cur.v[k];
    }
  };

  // one level, in one of three wave-uniform modes:
  //   0  no NaN / inf anywhere in this wave's windows: raw == sanitised, nothing to keep track of;
  //   1  only NaN flags (ocean fields with NaN on land): a NaN cell stays NaN at every later level, so the arithmetic
  //      runs on the sanitised values, a flagged cell's new window value is 0 and its flag is copied; its outputs
  //      (states, fbar) are overwritten with NaN when they are stored -- no raw operands are rebuilt;
  //   2  an inf is around: the raw operands are rebuilt from the flags (the general, slow form).
  auto level = [&](auto tt, auto mode_c) {
    constexpr int t = decltype(tt)::value;
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool FLAGGED = (MODE == 2);
    const T(&gS)[VEC] = G[t - 1][0];
    const T(&gC)[VEC] = G[t - 1][1];
    const T(&gN)[VEC] = G[t - 1][2];
    const T ev = from_upper_lane0(gC[0]);
    // flux form: the west-face flux of a cell IS the east-face flux of its western neighbour (same operands, same
    // rounding), so every east flux is computed once and the lane's first cell takes its west flux from the
    // lower lane -- one DPP hop of the flux instead of hops of the value and of the coefficient
    T fev[VEC], few = T(0), wv = T(0);
    if (KIND == K_FLUX) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T xE = (k == VEC - 1) ? ev : gC[k < VEC - 1 ? k + 1 : k];
        fev[k] = (xE - gC[k]) * cEq[t][k];
      }
      few = from_lower_lane0(fev[VEC - 1]);
    } else {
      wv = from_lower_lane0(gC[VEC - 1]);
    }
    T tkv[VEC];
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
      } else if (KIND == K_MASKZ) {
        const unsigned bb = (Bq[t] >> (8 * k)) & 0xFFu;
        const T wf = (T)(bb >> 5);  // wet-neighbour count, precomputed in bits 5-7
        L = -wf * xC + xE;
        L = L + xW;
        L = L + gN[k];
        L = L + gS[k];
        L = (bb & 1u) ? L : T(0);
      } else if (KIND == K_MASK) {
        const unsigned bb = (Bq[t] >> (8 * k)) & 0xFFu;
        const T mC = (bb & 1u) ? xC : T(0);
        const T wf = (T)(bb >> 5);  // wet-neighbour count, precomputed in bits 5-7
        L = -wf * mC + ((bb & 2u) ? xE : T(0));
        L = L + ((bb & 4u) ? xW : T(0));
        L = L + ((bb & 8u) ? gN[k] : T(0));
        L = L + ((bb & 16u) ? gS[k] : T(0));
        L = (bb & 1u) ? L : T(0);
      } else {
        const T fe = fev[k];
        const T fw = (k == 0) ? few : fev[k > 0 ? k - 1 : 0];
        const T fn = (gN[k] - xC) * cNq[t][k];
        const T fs = (xC - gS[k]) * cNq[t + 1][k];
        L = ((fe - fw) + (fn - fs)) * raq[t][k];
      }
      // raw centre of level t-1 (NaN/inf survive in "-x", filter.py:171-173)
      const T x = FLAGGED ? unsan(xC, (Rf[t - 1] >> (2 * VEC + 2 * k)) & 3u) : xC;
      constexpr bool FUSED = (KIND == K_FLUX);  // see gcmf_recurrence.hpp
      const T av = cheb_a<FUSED>(x, c, L);
      T tk;
      if (t == 1 && first) {
        tk = av;
        Fq[1][k] = cheb_acc_first<FUSED, T, FB>(P.p0, P.pk[0], x, av);
      } else {
        T x2;
        if (t == 1) x2 = Vp[k];
        else x2 = FLAGGED ? unsan(G[t >= 2 ? t - 2 : 0][0][k], (Rf[t >= 2 ? t - 2 : 0] >> (2 * k)) & 3u) : G[t >= 2 ? t - 2 : 0][0][k];
        tk = cheb_t<FUSED>(av, x2);
        Fq[t][k] = cheb_acc<FUSED, T, FB>(Fq[t][k], P.pk[t - 1], tk);
      }
      tkv[k] = tk;
      if (MODE == 1 && t >= S - 1) {
        // mode 1 ran on sanitised operands: what leaves the wave is NaN on a NaN cell (in mode 2 it already is)
        const bool isn = (Rf[t - 1] >> (2 * VEC + 2 * k)) & 1u;
        const T val = isn ? (T)__builtin_nan("") : tk;
        if (t == S - 1) out_v[k] = val;
        if (t == S) {
          out_u[k] = val;
          Fq[S][k] = isn ? (FB)__builtin_nan("") : Fq[S][k];
        }
      } else {
        if (t == S - 1) out_v[k] = tk;
        if (t == S) out_u[k] = tk;
      }
    }
    if (t < S && MODE == 1) {  // becomes the `new` row of this level's window; flagged cells: 0 and the same flag
      unsigned nf = 0u;
      bool odd = false;
      const unsigned fx = (Rf[t - 1] >> (2 * VEC)) & OLD_MASK;  // flags of the centre row (NaN bits only in this mode)
#pragma unroll
      for (int k = 0; k < VEC; ++k) odd = odd || (!((fx >> (2 * k)) & 1u) && !(mabs(tkv[k]) <= MLim<T>::big()));
      if (__any(odd)) {  // a NaN / inf appeared on a cell that was finite: classify properly (next iteration: mode 2)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          unsigned f;
          const T raw = ((fx >> (2 * k)) & 1u) ? (T)__builtin_nan("") : tkv[k];
          G[t < S ? t : 0][2][k] = msan_flag(raw, f);
          nf |= f << (2 * k);
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) G[t < S ? t : 0][2][k] = ((fx >> (2 * k)) & 1u) ? T(0) : tkv[k];
        nf = fx;
      }
      newflags[t < S ? t : 0] = nf;
    } else if (t < S) {
      unsigned nf = 0u;
      bool odd = false;
#pragma unroll
      for (int k = 0; k < VEC; ++k) odd = odd || !(mabs(tkv[k]) <= MLim<T>::big());
      if (SAN && __any(odd)) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          unsigned f;
          G[t < S ? t : 0][2][k] = msan_flag(tkv[k], f);
          nf |= f << (2 * k);
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) G[t < S ? t : 0][2][k] = tkv[k];
      }
      newflags[t < S ? t : 0] = nf;
    }
  };

  // Only the trapezoid of (level, row) pairs that can reach an output row of this strip is evaluated: level t
  // matters on rows [a-S+t, b+S-t).  The skipped corners (warm-up / drain iterations) are ~10 % of the
  // level-iterations at S=8.  The test is wave-uniform (scalar branch) -- but the branches also fence the
  // compiler's cross-level scheduling, which the one-wave-per-SIMD flux kernel needs more than it needs the saved
  // work (measured 330 -> 304 G with skipping), so it is enabled only where two waves share a SIMD.
  constexpr bool SKIP = (WavesPerSimd<T, KIND, S>::value == 2) && !GCMF_NO_SKIP;
  auto level_all = [&](auto mode_c, int r) {
#define GCMF_LEVEL(t_)                                                                           \
  if constexpr (S >= (t_)) {                                                                     \
    if (!SKIP || (r - (t_) >= a - S + (t_) && r - (t_) < b + S - (t_))) level(std::integral_constant<int, (t_)>{}, mode_c); \
  }
    GCMF_LEVEL(1) GCMF_LEVEL(2) GCMF_LEVEL(3) GCMF_LEVEL(4) GCMF_LEVEL(5) GCMF_LEVEL(6) GCMF_LEVEL(7) GCMF_LEVEL(8)
#undef GCMF_LEVEL
  };

  // ---- part 2: levels 1..S (level t produces row r-t), stores, window rotation ----
  auto compute = [&](int r) {
    unsigned anyf = newflags[0];
#pragma unroll
    for (int t = 0; t < S; ++t) anyf |= Rf[t];
    const bool flagged_any = SAN && __any(anyf != 0u);
    if (flagged_any) {
      constexpr unsigned INF_BITS = 0xAAAAAAAAu;  // bit 1 of every 2-bit flag
      // mode 1 only for the land-mask kinds: in the flux kernel (one wave per SIMD, 340+ registers) a third copy of the
      // levels costs the finite path 7-9 % (scratch appears) and gains 3 % on NaN input
      if (!IsMask<KIND>::value || __any((anyf & INF_BITS) != 0u)) level_all(std::integral_constant<int, 2>{}, r);
      else level_all(std::integral_constant<int, (IsMask<KIND>::value ? 1 : 2)>{}, r);
    } else {
      level_all(std::integral_constant<int, 0>{}, r);
    }

    // stores: T_{k-1+S} row r-S, T_{k-2+S} row r-S+1, fbar row r-S
    {
      const int ju = r - S;
      if (keep && ju >= a && ju < b) {
        const long long off = boff + (long long)ju * nx + col;
        if (!last) {
          mstore<T, VEC>(P.uo + off, out_u);
        } else if (P.area_weighted) {  // finalize(): / area (kernels.py:103-104)
          T ar[VEC];
          mload<T, VEC>(ar, P.area + (long long)ju * nx + col);
#pragma unroll
          for (int k = 0; k < VEC; ++k) Fq[S][k] = Fq[S][k] / (FB)ar[k];
        }
        mstore<FB, VEC>(P.fb_out + off, Fq[S]);
      }
      const int jv = r - S + 1;
      if (!last && keep && jv >= a && jv < b) mstore<T, VEC>(P.vo + boff + (long long)jv * nx + col, out_v);
    }

    // rotate the windows: old <- mid <- new, lags shift by one row
#pragma unroll
    for (int t = 0; t < S; ++t) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        G[t][0][k] = G[t][1][k];
        G[t][1][k] = G[t][2][k];
      }
      Rf[t] = ((Rf[t] >> (2 * VEC)) & OLD_MASK) | (newflags[t] << (2 * VEC));
    }
#pragma unroll
    for (int l = S; l >= 2; --l) {
      Bq[l] = Bq[l - 1];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        cEq[l][k] = cEq[l - 1][k];
        raq[l][k] = raq[l - 1][k];
        Fq[l][k] = Fq[l - 1][k];
      }
    }
#pragma unroll
    for (int l = S + 1; l >= 2; --l) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) cNq[l][k] = cNq[l - 1][k];
    }
  };

  // ---- march north with D rows of operands in flight (explicit slots so that they stay in registers) ----
  const int r_begin = a - S, r_end = b + S;  // rows loaded by this strip: [a-S, b+S)
  Row q0, q1, q2, q3;
  load_row(q0, r_begin);
  if (D >= 2) load_row(q1, min(r_begin + 1, r_end - 1));
  if (D >= 3) load_row(q2, min(r_begin + 2, r_end - 1));
  if (D >= 4) load_row(q3, min(r_begin + 3, r_end - 1));
#define GCMF_SLOT(Q, dd)                                          \
  if (r + (dd) < r_end) {                                         \
    consume(Q); /* waits for this slot only */                    \
    load_row(Q, min(r + (dd) + D, r_end - 1)); /* tail: harmless re-load of the last row */ \
    compute(r + (dd));                                            \
  }
  for (int r = r_begin; r < r_end; r += D) {
    GCMF_SLOT(q0, 0)
    if (D >= 2) { GCMF_SLOT(q1, 1) }
    if (D >= 3) { GCMF_SLOT(q2, 2) }
    if (D >= 4) { GCMF_SLOT(q3, 3) }
  }
#undef GCMF_SLOT
}

template <typename T, typename FB, int KIND, int S, int D>
__global__ __launch_bounds__(256, (WavesPerSimd<T, KIND, S>::value)) void k_scalar_multi(const MultiP<T, FB> P) {
  scalar_multi_march<T, FB, KIND, S, D>(P, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// ------------------------------------------------------------------------------------------------------
template <typename T, typename FB, int KIND, int S, int D>
static int launch_multi_s(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
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
  P.zrow = nullptr;
  P.lbits = nullptr;
  P.nfb = nullptr;
  P.xcd_per = 0;
  P.zigzag = 0;
  P.mbits = g.mbits;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  // strip height: one resident wave per register-file slot (2 waves/SIMD up to S=4, 1 above) and no second
  // round -- all strips march in lock-step, so a partial second round would idle most of the chip.  Measured
  // on MI355X (2400x3600 f64, S=4): 67 strips x 30 windows = 2010 waves (H=36) is the sweet spot.
  int H = pl->strip_rows;
  if (H <= 0) {
    const long long cap = 1024 * WavesPerSimd<T, KIND, S>::value;
    long long want = cap / ((long long)P.nwx * a.nbatch);
    if (want < 1) want = 1;
    H = (int)((nrows + want - 1) / want);
    if (H < 2 * S) H = 2 * S;  // keep the 2S warm-up rows per strip below half of the work
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
  for (int t = 0; t < MAX_PK; ++t) P.pk[t] = t < S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  dim3 block(256), grid((P.nwaves + 3) / 4, (unsigned)a.nbatch);
  hipLaunchKernelGGL((k_scalar_multi<T, FB, KIND, S, D>), grid, block, 0, s, P);
  note_kernel(pl, std::string("gcmf::k_scalar_multi<") + tyname<T>() + ", " + tyname<FB>() + ", " + std::to_string(KIND) + ", " +
                      std::to_string(S) + ", " + std::to_string(D) + ">", S);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

template <typename T, typename FB, int KIND> static int launch_multi_k(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 2: return launch_multi_s<T, FB, KIND, 2, 2>(pl, a, s);
    case 3: return launch_multi_s<T, FB, KIND, 3, 2>(pl, a, s);
    case 4: return pl->prefetch_rows == 1 ? launch_multi_s<T, FB, KIND, 4, 1>(pl, a, s) : launch_multi_s<T, FB, KIND, 4, 2>(pl, a, s);
    case 5: return launch_multi_s<T, FB, KIND, 5, 1>(pl, a, s);  // 5 and 7 serve remainders (63 = 7 x 8 + 7)
    case 6: return pl->prefetch_rows == 2 ? launch_multi_s<T, FB, KIND, 6, 2>(pl, a, s) : launch_multi_s<T, FB, KIND, 6, 1>(pl, a, s);
    case 7: return launch_multi_s<T, FB, KIND, 7, 1>(pl, a, s);
    case 8:
      // one wave per SIMD is issue-bound, not latency-bound: a shallower prefetch frees registers (fewer
      // VGPR<->AGPR moves) and measures 4-5 % faster than 2 rows in flight
      if (pl->prefetch_rows == 2) return launch_multi_s<T, FB, KIND, 8, 2>(pl, a, s);
      if (pl->prefetch_rows == 3) return launch_multi_s<T, FB, KIND, 8, 3>(pl, a, s);
      return launch_multi_s<T, FB, KIND, 8, 1>(pl, a, s);
  }
  set_error("launch_scalar_multi: unsupported S=%d", a.S);
  return GCMF_ERR_INVALID_ARG;
}

// one translation unit per stencil kind (compile time): the dtype dispatch of launch_scalar_multi
template <int KIND> static int launch_multi_kind(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_multi_k<double, double, KIND>(pl, a, s);
  if (a.fb_is_f32) return launch_multi_k<float, float, KIND>(pl, a, s);
  return launch_multi_k<float, double, KIND>(pl, a, s);
}

}  // namespace gcmf
