// k_ringc: the static-ring kernel (gcmf_ring_impl.hpp) evaluating the filter polynomial BACKWARDS (Clenshaw):
//
//     b_{n+1} = b_{n+2} = 0,   b_k = p_k f + 2 A(b_{k+1}) - b_{k+2}  (k = n .. 1),   result = p_0 f + A(b_1) - b_2,
//     A(x) = -x - c L(x)                                   == sum_k p_k T_k(A) f  of reference filter.py:162-212
//
// The forward recurrence carries three planes per field (T_{k-1}, T_{k-2}, fbar: fbar is read AND written by every launch); the
// backward one carries two (b_{k+1}, b_{k+2}) and re-reads the constant input f: one plane less per launch (8 instead of 9 for
// the flux kinds, 5 + 1 byte instead of 6 + 1 byte for the land-mask kinds), and the kernels run at the rate the memory
// system gives.  Same stencils, same time skewing, rings, halo geometry and launch structure as k_ring:
// b_n = p_n f needs no stencil: the first launch forms it as it loads f (its "input state" is (b_n, 0)); "level" l = 1 .. n of a
// filter produces b_{n-l}, level n the result (its 2 A(.) is A(.)); the n levels are cut into launches of S = 5 .. 8 -- as
// many Laplacian applications and launches as the forward recurrence.
//
//   * same NaN semantics as the reference: the stencil sees nan_to_num of its operands, "-x", b_{k+2} and p_k f keep a NaN, so
//     an input NaN in a wet cell stays in that cell's state and result and its column of the stencil is zero -- exactly
//     what happens to T_k in the forward recurrence.  The fast march only WATCHES (last level); a strip that meets a
//     non-finite value is redone by the SAME march with nan_to_num on the stencil operands (SANI = true), which needs no
//     other kernel: no fbar is accumulated, so a strip is always re-computable from its inputs.
//   * isolated (land) cells: f is masked to zero as it is loaded (every launch: +1 byte per cell), so they stay zero in the
//     state; k_land_fix writes their own polynomial (forward recurrence, as the reference computes it) at the end.
//   * prepare() / finalize() of the area-weighted types: f * area as it is loaded, result / area as it is stored.
//   * rounding differs from the forward recurrence in the last bits (measured <= 3e-15 relative): results are NOT bit-identical
//     with the single-step kernels; they are bit-identical across different cuts of the levels into launches.
//
// f64 state (all scalar kinds) and f32 state of the flux kinds (four cells per lane, f64 result).  On tripolar plans the launch stops S rows
// below the seam and k_fold_band's backward form (gcmf_foldband.hip) advances those rows beside it.
#pragma once
#include "gcmf_ring_impl.hpp"

namespace gcmf {

// the VEC bytes (mask bits / land bits) of a lane's cells in one load
template <int VEC> __device__ __forceinline__ unsigned cell_bytes(const uint8_t *p) {
  if constexpr (VEC == 2) return *reinterpret_cast<const unsigned short *>(p);
  else return *reinterpret_cast<const unsigned *>(p);
}

// one march of a strip; returns whether this wave met a non-finite value in its last level (wave-uniform)
// XE: early exits also for the flux kinds (k_ringcs: slabs that do not own a tripole seam, see below)
// XE6: ONE early exit, in the middle of the ring period (whole f64 flux grids at nine levels, round 6: a strip marches a multiple of six
// rows instead of twelve)
// ZIP: two strips of one window that share a boundary ("seam") march AWAY from it, side by side in one workgroup (k_ringcz, round 6): neither
// warms its levels up over S ghost rows on that side -- at its first row every level takes the flux across the seam from the row the partner
// has just produced (one row per level through LDS, a workgroup barrier per level: S - 1 of them in the first ring period), so a strip
// marches H + S + 1 rows instead of H + 2 S and all its levels start within S + 1 phases.  Same operands, same operations: same bits.
// (wx: the wave's window; [a, b): the rows it owns; boff: its field's offset in the planes; odd: odd strips of the flux kinds march upwards;
// zmine / zpart: ZIP -- this wave's and its partner's S - 1 rows of LDS)
// (Tried on top of it, round 6: the three coefficient rings of a wave in LDS instead of accumulation registers -- a VALU instruction cannot
// read an accumulation register, so every level pays twelve v_accvgpr_read per row, a third of its non-arithmetic VALU work; from LDS
// they are three 16-byte reads.  3 983 -> 1 460 v_accvgpr_read in the eight-level kernel, 30 registers fewer, and SLOWER everywhere: 300 x
// 3600 424 -> 360 G, 1080 x 1440 457 -> 432 G, 1440 x 2880 617 -> 535 G -- one wave per SIMD has nothing to hide the LDS latency behind.
// Removed.)
template <bool ZIP> constexpr bool ringc_ramp_on(int t, int ph) { return ZIP ? (t == 1 ? ph >= 1 : ph >= t + 1) : ph >= 2 * t - 1; }

// ZIP, fold (wave-uniform, k_ringcz on the plan that owns the TRIPOLE SEAM): the strip starts at the grid's top row and its partner is the
// strip of the MIRROR window -- the northern neighbour of cell (rows - 1, i) is (rows - 1, nx - 1 - i) (reference kernels.py:33-40,
// 517-585) -- so the row the partner hands over is read with the lanes (and a lane's cells) reversed, level 1's seam flux comes from the
// partner's row of the input state as well (one more exchange), and the face coefficient is the top row's own north face (folded at plan
// time).  pos_at: the window's first column instead of wx * WI - M; [klo, khi): the columns this window keeps.
template <typename T, int KIND, int S, bool FIRST, bool SANI, bool XE = false, bool XE6 = false, bool ZIP = false>
__device__ __forceinline__ bool ringc_march(const MultiP<T, T> &P, const int wx, const int a, const int b, const long long boff, const bool odd,
                                            T *zmine = nullptr, const T *zpart = nullptr, const bool fold = false, const int pos_at = 0,
                                            const int klo = -(1 << 30), const int khi = 1 << 30) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;
  constexpr int R = RingGeom::R, D = RingGeom::D, RU = RingGeom::RU, RV = RingGeom::RV;
  static_assert(R >= S + D && R % 3 == 0 && R % RU == 0 && R % RV == 0 && RU >= 3 + D && RV >= 1 + D, "ring periods");
  constexpr bool FLUX = (KIND == K_FLUX), MASK = (KIND == K_MASKZ);
  static_assert(!ZIP || FLUX, "the seam exchange is the flux kinds' (a carried face flux per level)");
  constexpr bool WATCH = (KIND != K_REG) && !SANI;  // K_REG has no nan_to_num in the reference: NaN spreads by plain arithmetic
  constexpr bool FUSED = true;   // nothing here is bit-identical with numpy anyway: every multiply-add pair is one fma

  const int lane = threadIdx.x & 63;
  const int nx = P.nx, rows = P.rows;
  const int pos = ((ZIP && fold) ? pos_at : wx * WI - M) + lane * VEC;
  int col_s = pos % nx;
  if (col_s < 0) col_s += nx;
  const unsigned col = (unsigned)col_s;
  const unsigned colT = col * (unsigned)sizeof(T);
  const bool keep = (lane * VEC >= M) && (lane * VEC < W - M) && (pos < nx) && (!ZIP || (pos >= klo && pos < khi));
  const T c = (T)P.c;
  const bool last = P.last;

  T G0[RU][VEC];     // rows of b_{k+1} (the newer input state), slot = (row - r_begin) mod RU
  T G[S][3][VEC];    // G[t], t = 1..S-1: rows of level t, slot = (row - r_begin) mod 3
  T FN[S + 1][VEC];  // K_FLUX: carried north-face flux per level
  T cE[R][VEC], cN[R][VEC], ra[R][VEC];
  unsigned B[R];     // K_MASKZ: mask bytes
  T Ff[R][VEC];      // rows of the constant input f (prepared: * area, land -> 0), slot as the coefficient rows
  unsigned Zc[R];    // land bits of those rows
  T ARc[R][VEC];     // area of those rows (area-weighted types)
  T V[RV][VEC];      // rows of b_{k+2}
  unsigned Zu[RU];   // FIRST: land bits of the rows of f that become b_n (slots of G0)
  T ARu[RU][VEC];    // FIRST, area-weighted types: their area
  T cNs[VEC];        // ZIP: the seam face's coefficient
#pragma unroll
  for (int k = 0; k < VEC; ++k) cNs[k] = T(0);
#pragma unroll
  for (int t = 0; t < S; ++t) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) G[t][0][k] = G[t][1][k] = G[t][2][k] = FN[t + 1][k] = T(0);
  }
#pragma unroll
  for (int l = 0; l < RU; ++l) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) G0[l][k] = ARu[l][k] = T(0);
    Zu[l] = 0u;
  }
#pragma unroll
  for (int l = 0; l < R; ++l) {  // rings start at zero (see k_ring: the NaN watch also sees the levels of the first rows)
    B[l] = 0u;
    Zc[l] = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cE[l][k] = cN[l][k] = ra[l][k] = Ff[l][k] = ARc[l][k] = T(0);
  }
#pragma unroll
  for (int l = 0; l < RV; ++l) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) V[l][k] = T(0);
  }

  const bool wrap = P.wrap;
  const int r_begin = ZIP ? a - 1 : a - S, r_last = b + S - 1;   // (ZIP: march row a - 1 is the partner's first row)
  // Odd strips of the flux kinds march UPWARDS (P.zigzag): the march below runs in "march order" m = r_begin - 1, r_begin, ... and
  // visits the grid row  mir - m  instead of m  (mir = a + b - 1 maps the strip and its ghost rows onto themselves).  Two strips that
  // share a boundary then reach it at the same time -- both at the start or both at the end of their marches -- so that the 2 S rows
  // each of them re-reads from the other's territory are found in the L2 / the memory-side cache instead of in HBM (same-direction
  // marches read them a whole march apart).  The arithmetic is the same instruction stream: "north" in march order is the grid's
  // south, the row of cN that travels with a centre row j is that of row j - 1, the carried flux is minus the grid's north flux, and
  // (fe - fw) + (fn - fs) is bit-for-bit symmetric under that exchange.
  const bool up = FLUX && (ZIP || P.zigzag) && odd;
  const int mir = a + b - 1;
  int cj, cj_prev;
  bool cout_, cout_prev;
  int cr = up ? mir - (r_begin - 1) : r_begin - 1;   // the cursor's grid row
  {
    const int r = cr;
    const bool lo = r < 0, hi = r >= rows;
    cj = wrap ? (r + (lo ? rows : 0) - (hi ? rows : 0)) : (lo ? 0 : (hi ? rows - 1 : r));
    cout_ = !wrap && (lo || hi);
    cj_prev = cj;
    cout_prev = cout_;
  }
  auto advance = [&]() {
    cj_prev = cj;
    cout_prev = cout_;
    if (!up) {
      ++cr;
      const int jn = cj + ((wrap || cr > 0) ? 1 : 0);
      const bool hit = (jn >= rows);
      cj = hit ? (wrap ? 0 : rows - 1) : jn;
    } else {
      --cr;
      const int jn = cj - ((wrap || cr < rows - 1) ? 1 : 0);
      const bool hit = (jn < 0);
      cj = hit ? (wrap ? rows - 1 : 0) : jn;
    }
    cout_ = !wrap && (cr < 0 || cr >= rows);
  };
  const T *fplane = P.fb_in + boff;  // the constant input f (Clenshaw has no fbar: the pointer slot is reused)
  const bool has_land = P.lbits != nullptr;
  const uint8_t *zbase = has_land ? P.lbits : reinterpret_cast<const uint8_t *>(P.fb_in);  // (valid bytes, ignored)
  const bool weigh = !FLUX && P.area_weighted;
  const T *abase = weigh ? P.area : P.fb_in;  // (an unconditional load, ignored when there is no area)

  auto load_u = [&](auto slot_c) {  // the cursor's row of b_{k+1}; in a first launch that is b_n = p_n f, formed from the row of f
    constexpr int sl = decltype(slot_c)::value;
    const long long rcur = (long long)(cj * nx);
    if constexpr (!FIRST) {
      mload<T, VEC>(G0[sl], lane_ptr(P.u0 + boff + rcur, colT));
    } else {
      mload<T, VEC>(G0[sl], lane_ptr(fplane + rcur, colT));
      Zu[sl] = cell_bytes<VEC>(lane_ptr(zbase + rcur, col));
      if constexpr (!FLUX) mload<T, VEC>(ARu[sl], lane_ptr(abase + rcur, colT));
    }
  };
  auto load_centre = [&](auto slot_c, auto vslot_c) {  // what travels with the row before it: b_{k+2}, f, coefficients / mask bits
    constexpr int sl = decltype(slot_c)::value;
    constexpr int vs = decltype(vslot_c)::value;
    const bool out_c = cout_prev;
    const long long rc = (long long)(cj_prev * nx);
    if constexpr (FLUX) {
      const T *pE = out_c ? P.zrow : P.cE + rc;
      // (an upward march: the face between the centre row and the row the cursor is on -- grid row j - 1's north face)
      const T *pN = !up ? (out_c ? P.zrow : P.cN + rc) : ((out_c || cout_) ? P.zrow : P.cN + (long long)(cj * nx));
      const T *pA = out_c ? P.zrow : P.ra + rc;
      mload<T, VEC>(cE[sl], lane_ptr(pE, colT));
      mload<T, VEC>(cN[sl], lane_ptr(pN, colT));
      mload<T, VEC>(ra[sl], lane_ptr(pA, colT));
    }
    if constexpr (MASK) {
      const uint8_t *mp = lane_ptr(out_c ? (const uint8_t *)P.zrow : P.mbits + rc, col);
      B[sl] = cell_bytes<VEC>(mp);
    }
    mload<T, VEC>(Ff[sl], lane_ptr(fplane + rc, colT));
    Zc[sl] = cell_bytes<VEC>(lane_ptr(zbase + rc, col));
    if constexpr (!FLUX) mload<T, VEC>(ARc[sl], lane_ptr(abase + rc, colT));
    if constexpr (!FIRST) mload<T, VEC>(V[vs], lane_ptr(P.v0 + boff + rc, colT));  // a first launch: b_{n+1} = 0
  };

  bool bad = false;
  T out_v[VEC], out_u[VEC];

  auto level = [&](auto tt, auto ph_c) {
    constexpr int t = decltype(tt)::value;
    constexpr int ph = decltype(ph_c)::value;
    constexpr int sS = pmod(ph - t - 1, 3), sC = pmod(ph - t, 3), sN = pmod(ph - t + 1, 3);
    constexpr int sl = pmod(ph - t + 1, R);
    const T(&rS)[VEC] = (t == 1) ? G0[pmod(ph - 2, RU)] : G[t >= 2 ? t - 1 : 1][sS];
    const T(&rC)[VEC] = (t == 1) ? G0[pmod(ph - 1, RU)] : G[t >= 2 ? t - 1 : 1][sC];
    const T(&rN)[VEC] = (t == 1) ? G0[pmod(ph, RU)] : G[t >= 2 ? t - 1 : 1][sN];
    // what the stencil sees: its operands through nan_to_num in the redo pass (kernels.py:175, 300), as they are otherwise
    T gS[VEC], gC[VEC], gN[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      gS[k] = SANI ? msan(rS[k]) : rS[k];
      gC[k] = SANI ? msan(rC[k]) : rC[k];
      gN[k] = SANI ? msan(rN[k]) : rN[k];
    }
    const T ev = from_upper_lane0(gC[0]);
    T fev[VEC], few = T(0), wv = T(0);
    if constexpr (FLUX) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T xE = (k == VEC - 1) ? ev : gC[k < VEC - 1 ? k + 1 : k];
        fev[k] = (xE - gC[k]) * cE[sl][k];
      }
      few = from_lower_lane0(fev[VEC - 1]);
    } else {
      wv = from_lower_lane0(gC[VEC - 1]);
    }
    // the last level of the last launch is the result  p_0 f + A(b_1) - b_2: A, not 2 A
    const T two = (last && t == S) ? T(1) : T(2);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T xC = gC[k];
      T L;
      if constexpr (FLUX) {
        const T fe = fev[k];
        const T fw = (k == 0) ? few : fev[k > 0 ? k - 1 : 0];
        const T fn = (gN[k] - xC) * cN[sl][k];
        L = ((fe - fw) + (fn - FN[t][k])) * ra[sl][k];
        FN[t][k] = fn;
      } else {
        const T xW = (k == 0) ? wv : gC[k > 0 ? k - 1 : 0];
        const T xE = (k == VEC - 1) ? ev : gC[k < VEC - 1 ? k + 1 : k];
        if constexpr (MASK) {
          const unsigned bb = (B[sl] >> (8 * k)) & 0xFFu;
          const T wf = (T)(bb >> 5);
          L = rfma(-wf, xC, xE);
          L = L + xW;
          L = L + gN[k];
          L = L + gS[k];
          L = (bb & 1u) ? L : T(0);
        } else {
          L = rfma(T(-4), xC, xE);
          L = L + xW;
          L = L + gN[k];
          L = L + gS[k];
        }
      }
      const T av = cheb_a<FUSED>(rC[k], c, L);  // "-x" takes the raw value: a NaN stays in its cell (filter.py:166-175)
      const T x2 = (t == 1) ? V[ph % RV][k] : (t == 2 ? G0[pmod(ph - 2, RU)][k] : G[t >= 3 ? t - 2 : 1][sC][k]);
      T tk;
      if (FUSED) {
        tk = rfma(two, av, -x2);
        tk = rfma((T)P.pk[t - 1], Ff[sl][k], tk);
      } else {
        tk = two * av - x2;
        tk = tk + (T)P.pk[t - 1] * Ff[sl][k];
      }
      if (t < S) G[t < S ? t : 0][sC][k] = tk;
      if (t == S - 1) out_v[k] = tk;
      if (t == S) {
        out_u[k] = tk;
        if (WATCH) bad = bad || !(mabs(tk) <= MLim<T>::big());
      }
    }
  };

  auto phase = [&](auto ph_c, int r, auto pro_c) {
    constexpr int ph = decltype(ph_c)::value;
    // the strip's first period: level t starts with phase 2 t - 1 (below).  Flux kinds only: the land-mask / REGULAR kernels lose 3-4 % at
    // full size with the peeled period (config 2: 82 -> 85 us per launch on one box) and gain 3 % on 1/4-degree grids.
    constexpr bool PRO = decltype(pro_c)::value && FLUX;
    advance();
    load_centre(ic<(ph + D) % R>{}, ic<(ph + D) % RV>{});
    load_u(ic<(ph + D) % RU>{});
    if constexpr (FIRST) {  // the row of f that has just arrived becomes a row of b_n = p_n f (prepared, land out)
      constexpr int su = ph % RU;
      const T pn = (T)P.p0;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (!FLUX) G0[su][k] = weigh ? G0[su][k] * ARu[su][k] : G0[su][k];
        G0[su][k] = (!has_land || ((Zu[su] >> (8 * k)) & 1u)) ? pn * G0[su][k] : T(0);
      }
    }
    {  // the f row level 1 uses now: prepare() and land out (once per row; the later levels find it that way in the ring)
      constexpr int s1 = ph % R;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (!FLUX) Ff[s1][k] = weigh ? Ff[s1][k] * ARc[s1][k] : Ff[s1][k];
        Ff[s1][k] = (!has_land || ((Zc[s1] >> (8 * k)) & 1u)) ? Ff[s1][k] : T(0);
      }
    }
    // The ramp: phase q of a strip (q = 0 at row r_begin = a - S) produces row a - S + q - t of level t, and the strip needs level t from
    // row a - (S - t) on -- phase 2 t.  One phase earlier the level has to run for the flux it carries to its next row (its value there
    // is never used); before that it would compute rows nobody reads.  S^2 of the (H + 2 S) S level-rows of a strip: 8 % at 80 rows, a
    // quarter of the instruction stream of a 12-row strip (8-way slabs, 1/4-degree grids).
    // ZIP: the strip starts AT the seam (q = 0 at march row a - 1, the partner's first row): level 1 runs from phase 1 (for the seam face's
    // flux, out of its own two rows), level t >= 2 from phase t + 1 -- its first row, with the seam face's flux from the exchange below.
    if constexpr (ZIP && PRO && ph == 1) {
      if (fold) mload<T, VEC>(cNs, lane_ptr(P.cN + (long long)(rows - 1) * nx, colT));   // the top row's north face: the fold
      else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) cNs[k] = cN[1][k];   // (the coefficient row of march row a - 1: its face towards row a)
      }
    }
    if constexpr (!PRO || ringc_ramp_on<ZIP>(1, ph)) level(ic<1>{}, ph_c);
    if constexpr (S >= 2 && (!PRO || ringc_ramp_on<ZIP>(2, ph))) level(ic<2>{}, ph_c);
    if constexpr (S >= 3 && (!PRO || ringc_ramp_on<ZIP>(3, ph))) level(ic<3>{}, ph_c);
    if constexpr (S >= 4 && (!PRO || ringc_ramp_on<ZIP>(4, ph))) level(ic<4>{}, ph_c);
    if constexpr (S >= 5 && (!PRO || ringc_ramp_on<ZIP>(5, ph))) level(ic<5>{}, ph_c);
    if constexpr (S >= 6 && (!PRO || ringc_ramp_on<ZIP>(6, ph))) level(ic<6>{}, ph_c);
    if constexpr (S >= 7 && (!PRO || ringc_ramp_on<ZIP>(7, ph))) level(ic<7>{}, ph_c);
    if constexpr (S >= 8 && (!PRO || ringc_ramp_on<ZIP>(8, ph))) level(ic<8>{}, ph_c);
    if constexpr (S >= 9 && (!PRO || ringc_ramp_on<ZIP>(9, ph))) level(ic<9>{}, ph_c);
    if constexpr (ZIP && PRO && ph == 1) {
      // (launch-uniform: in a launch with fold strips EVERY workgroup meets at this barrier -- pairs and fold strips share workgroups)
      if (P.fold_rows > 0 && !fold) __syncthreads();
      if (fold) {   // level 1's seam flux: the partner's top row of the input state (slot 1 of the ring of b_{k+1}; a first launch has just formed it)
        T own[VEC], oth[VEC], rev[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) own[k] = G0[1][k];
        mstore<T, VEC>(zmine + (S - 1) * W + lane * VEC, own);
        __syncthreads();
        mload<T, VEC>(rev, zpart + (S - 1) * W + (63 - lane) * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) oth[k] = rev[VEC - 1 - k];
#pragma unroll
        for (int k = 0; k < VEC; ++k) FN[1][k] = ((SANI ? msan(own[k]) : own[k]) - (SANI ? msan(oth[k]) : oth[k])) * cNs[k];
      }
    }
    if constexpr (ZIP && PRO && ph >= 2 && ph <= S) {
      // level ph - 1 has just produced the strip's first row (slot 1 of its ring): the partner's level ph needs it for the flux across the
      // seam, this wave's level ph needs the partner's -- what the march row before the first one would have left in FN[ph]
      constexpr int t = ph - 1;
      T own[VEC], oth[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) own[k] = G[t < S ? t : 1][1][k];
      mstore<T, VEC>(zmine + (t - 1) * W + lane * VEC, own);
      __syncthreads();
      {
        T got[VEC];
        mload<T, VEC>(got, zpart + (t - 1) * W + (fold ? 63 - lane : lane) * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) oth[k] = fold ? got[VEC - 1 - k] : got[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) FN[t + 1][k] = ((SANI ? msan(own[k]) : own[k]) - (SANI ? msan(oth[k]) : oth[k])) * cNs[k];
    }
    const int ju = up ? mir - (r - S) : r - S;
    if (ju >= a && ju < b) {  // wave-uniform
      const long long off = boff + (long long)ju * nx;
      if (keep) {
        if (!last) {
          mstore<T, VEC>(lane_ptr(P.uo + off, colT), out_u);
        } else {
          if (!FLUX && P.area_weighted) {  // finalize(): / area (kernels.py:103-104)
            T ar[VEC];
            mload<T, VEC>(ar, lane_ptr(P.area + (long long)ju * nx, colT));
#pragma unroll
            for (int k = 0; k < VEC; ++k) out_u[k] = out_u[k] / ar[k];
          }
          if (P.d_out) {   // (wave-uniform) f32 state, f64 result: NumPy >= 2 promotes p[k] * T
            double dd[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) dd[k] = (double)out_u[k];
            mstore<double, VEC>(lane_ptr(P.d_out + off, 2 * colT), dd);
          } else {
            mstore<T, VEC>(lane_ptr(P.fb_out + off, colT), out_u);
          }
        }
      }
    }
    const int jv = up ? mir - (r - S + 1) : r - S + 1;
    if (!last && jv >= a && jv < b) {
      T *rowp = P.vo + boff + (long long)jv * nx;
      if (keep) mstore<T, VEC>(lane_ptr(rowp, colT), out_v);
    }
  };

  advance();
  load_centre(ic<0>{}, ic<0>{});
  load_u(ic<0>{});
  advance();
  load_centre(ic<1>{}, ic<1>{});
  load_u(ic<1>{});
  advance();
  load_centre(ic<2>{}, ic<2>{});
  load_u(ic<2>{});
  static_assert(D == 3, "prologue");
  // The flux kinds march whole periods: every early exit costs this kernel dozens of registers (146 -> 204 AGPRs with an exit
  // every four rows), and on tripolar plans k_fold_band's waves have to fit on the SIMDs NEXT to these (gcmf_foldband.hip).
  constexpr bool EARLY = (KIND != K_FLUX) || XE;
  // k_ringcz: an exit every second row.  Nine levels: 52 bytes of scratch that way; a second form with its exits after rows 2, 6, 10 instead of
  // 4, 8, 12 (for strips whose march is 4 k + 2 rows) fits (506 registers) but runs 7 % longer per row -- 1080 x 1440: 26 rows in 34.4 us
  // against 28 rows in 32.5 us -- and was dropped.
  constexpr bool XE2 = ZIP && XE && S <= 8;
  if constexpr (FLUX) {  // the first period, peeled: the ramp of the levels
    constexpr std::integral_constant<bool, true> pro{};
    phase(ic<0>{}, r_begin, pro);
    phase(ic<1>{}, r_begin + 1, pro);
    phase(ic<2>{}, r_begin + 2, pro);
    phase(ic<3>{}, r_begin + 3, pro);
    phase(ic<4>{}, r_begin + 4, pro);
    phase(ic<5>{}, r_begin + 5, pro);
    phase(ic<6>{}, r_begin + 6, pro);
    phase(ic<7>{}, r_begin + 7, pro);
    phase(ic<8>{}, r_begin + 8, pro);
    phase(ic<9>{}, r_begin + 9, pro);
    phase(ic<10>{}, r_begin + 10, pro);
    phase(ic<11>{}, r_begin + 11, pro);
    if (WATCH && __any(bad)) return true;
    if (r_begin + 11 >= r_last) return false;   // (a last strip of one or two rows at S = 5)
  }
  constexpr std::integral_constant<bool, false> run{};
  for (int r0 = FLUX ? r_begin + R : r_begin;; r0 += R) {
    bool done = true;
    do {
      phase(ic<0>{}, r0, run);
      phase(ic<1>{}, r0 + 1, run);
      if (XE2 && r0 + 1 >= r_last) break;
      phase(ic<2>{}, r0 + 2, run);
      phase(ic<3>{}, r0 + 3, run);
      if (EARLY && r0 + 3 >= r_last) break;     // (the last period is left after the strip's last row, see k_ring)
      phase(ic<4>{}, r0 + 4, run);
      phase(ic<5>{}, r0 + 5, run);
      if ((XE6 || XE2) && r0 + 5 >= r_last) break;
      phase(ic<6>{}, r0 + 6, run);
      phase(ic<7>{}, r0 + 7, run);
      if (EARLY && r0 + 7 >= r_last) break;
      phase(ic<8>{}, r0 + 8, run);
      phase(ic<9>{}, r0 + 9, run);
      if (XE2 && r0 + 9 >= r_last) break;
      phase(ic<10>{}, r0 + 10, run);
      phase(ic<11>{}, r0 + 11, run);
      done = (r0 + 11 >= r_last);
    } while (false);
    if (WATCH && __any(bad)) return true;
    if (done) break;
  }
  return false;
}

// What wave `wid` of a launch marches.  gridDim.y = the batch (P.npack == 0): the strip st = wid / nwx of field blockIdx.y.  Packed batches
// (P.npack > 0, round 6): the fields of the batch are ONE column of npack * nrows rows per window, cut into runs of P.H rows; a wave walks
// its run, which crosses at most one field boundary (H <= nrows): up to two (field, row range) segments, each a march of its own with its
// 2 S warm-up rows.  16 fields x 33 windows of a 300-row slab tile 1024 wave slots at ~70 % as whole strips (a strip cannot cross from one
// field into the next) and at ~95 % this way.
template <typename T, int KIND, int S, bool FIRST, bool XE, bool XE6, bool PACK>
__device__ __forceinline__ void ringc_walk(const MultiP<T, T> &P, const int wid) {
  const int wx = wid % P.nwx, st = wid / P.nwx;
  if constexpr (!PACK) {   // one strip of one field (the instruction stream of rounds 2-5: the walk below costs the land-mask kernel 5 %)
    const int a = P.out_lo + st * P.H;
    const int b = min(a + P.H, P.out_hi);
    const long long boff = (long long)blockIdx.y * P.bstride;
    if (ringc_march<T, KIND, S, FIRST, false, XE, XE6>(P, wx, a, b, boff, (st & 1) != 0)) {
      if constexpr (KIND != K_REG) {
        if (P.nfb && (threadIdx.x & 63) == 0) atomicAdd(P.nfb, 1u);                      // instrumentation: gcmf_ring_fallbacks
        ringc_march<T, KIND, S, FIRST, true, XE, XE6>(P, wx, a, b, boff, (st & 1) != 0);   // the same strip again, operands through nan_to_num
      }
    }
    return;
  }
  const int nrows = P.out_hi - P.out_lo;
  int v0 = st * P.H;   // (rows of the batch's column: the launcher keeps batch x rows below 2^30)
  const int v1 = min(v0 + P.H, P.npack * nrows);
  while (v0 < v1) {   // (wave-uniform: scalar registers)
    const int fld = v0 / nrows, r = v0 - fld * nrows;
    const int len = min(v1 - v0, nrows - r);
    const int a = P.out_lo + r, b = a + len;
    const long long boff = (long long)fld * P.bstride;
    if (ringc_march<T, KIND, S, FIRST, false, XE, XE6>(P, wx, a, b, boff, (st & 1) != 0)) {
      if constexpr (KIND != K_REG) {
        if (P.nfb && (threadIdx.x & 63) == 0) atomicAdd(P.nfb, 1u);
        ringc_march<T, KIND, S, FIRST, true, XE, XE6>(P, wx, a, b, boff, (st & 1) != 0);
      }
    }
    v0 += len;
  }
}

template <typename T, int KIND, int S, bool FIRST>
__global__ __launch_bounds__(256, 1) void k_ringc(const MultiP<T, T> P) {
  int bx = blockIdx.x;
  if (P.xcd_per > 0 && bx < 8 * P.xcd_per) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= P.nwaves) return;
  ringc_walk<T, KIND, S, FIRST, false, false, false>(P, wid);
}

// ... and for packed batches (ringc_walk<PACK>): XE = the early-exit form of the flux kinds (k_ringcs)
template <typename T, int KIND, int S, bool FIRST, bool XE>
__global__ __launch_bounds__(256, 1) void k_ringcp(const MultiP<T, T> P) {
  int bx = blockIdx.x;
  if (P.xcd_per > 0 && bx < 8 * P.xcd_per) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= P.nwaves) return;
  ringc_walk<T, KIND, S, FIRST, XE, false, true>(P, wid);
}

// The flux kinds with early exits: for ROW SLABS that own no tripole seam (the ranks of a multi-GPU run, the row blocks of the host
// pipeline).  Their strips are short -- 300 rows of an 8-way slab: 11 + 2 S rows per strip -- so rows up to the next whole ring period
// are a third of the march, and nothing has to fit beside these waves (the extra ~60 registers of the exits are free here).
template <typename T, int S, bool FIRST>
__global__ __launch_bounds__(256, 1) void k_ringcs(const MultiP<T, T> P) {
  int bx = blockIdx.x;
  if (P.xcd_per > 0 && bx < 8 * P.xcd_per) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= P.nwaves) return;
  ringc_walk<T, K_FLUX, S, FIRST, true, false, false>(P, wid);
}

// k_ringcz: strips ZIPPED in pairs at a shared seam (ringc_march<ZIP>), early exits: mid-size whole grids and row slabs of the f64 flux kinds
// without a tripole seam, where a strip is as short as its ghost zones.  The rows [out_lo, out_hi) of a window are cut into P.nstrips / 2
// pairs of (almost) equal height; a pair is cut in the middle: the lower strip marches down the grid from the cut, the upper one up.  A
// workgroup = the two pairs of two neighbouring windows (pair (2 bx + w / 2), member w % 2); a pair past the end idles through the barriers.
// A non-finite value met by ANY wave of the workgroup sends all four through the nan_to_num march (they meet at its barriers).
template <typename T, int S, bool FIRST, bool XE>
__global__ __launch_bounds__(256, 1) void k_ringcz(const MultiP<T, T> P) {
  constexpr int VEC = 16 / sizeof(T), W = 64 * VEC, M = (S + VEC - 1) / VEC * VEC, WI = W - 2 * M;
  __shared__ __attribute__((aligned(16))) T zl[4][S * W];   // (row S - 1: the fold strips' exchange of the input state)
  int bx = blockIdx.x;
  if (P.xcd_per > 0 && bx < 8 * P.xcd_per) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int np = P.nstrips >> 1, nnorm = P.nwx * np;
  const int unit = bx * 2 + (w >> 1);   // a pair of waves: a pair of strips, or (after the pairs) a window of the top rows and its mirror image
  const bool fold = unit >= nnorm;
  const long long boff = (long long)blockIdx.y * P.bstride;
  const bool upper = (w & 1) != 0;
  bool active, odd;
  int wx = 0, a, b, pos_at = 0, klo = -(1 << 30), khi = 1 << 30;
  if (!fold) {
    const int pid = unit;
    active = true;
    wx = pid % P.nwx;
    const int pp = pid / P.nwx;
    const long long nrows = P.out_hi - P.fold_rows - P.out_lo;
    const int lo = P.out_lo + (int)((long long)pp * nrows / np), hi = P.out_lo + (int)((long long)(pp + 1) * nrows / np), mid = lo + (hi - lo) / 2;
    a = upper ? mid : lo;
    b = upper ? hi : mid;
    odd = !upper;
  } else {   // a window of the western half (even wave) and its mirror image (odd wave), both marching down the grid from the seam
    const int fid = unit - nnorm;
    active = fid < P.nfw;
    a = P.out_hi - P.fold_rows;
    b = P.out_hi;
    odd = true;
    const int pw = fid * WI - M;
    pos_at = upper ? P.nx - pw - W : pw;
    klo = upper ? P.nx / 2 : 0;
    khi = upper ? P.nx : P.nx / 2;
  }
  const int nbar = P.fold_rows > 0 ? S : S - 1;   // (barriers of a march: one more in a launch with fold strips)
  bool bad = false;
  if (active) bad = ringc_march<T, K_FLUX, S, FIRST, false, XE, false, true>(P, wx, a, b, boff, odd, zl[w], zl[w ^ 1], fold, pos_at, klo, khi);
  else
    for (int k = 0; k < nbar; ++k) __syncthreads();
  if (__syncthreads_or(bad ? 1 : 0)) {
    if (active) {
      if (P.nfb && (threadIdx.x & 63) == 0) atomicAdd(P.nfb, 1u);
      ringc_march<T, K_FLUX, S, FIRST, true, XE, false, true>(P, wx, a, b, boff, odd, zl[w], zl[w ^ 1], fold, pos_at, klo, khi);
    } else {
      for (int k = 0; k < nbar; ++k) __syncthreads();
    }
  }
}

template <typename T, int S, bool FIRST>
static int launch_ringc_zip_sf(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T), W = 64 * VEC, M = (S + VEC - 1) / VEC * VEC, WI = W - 2 * M;
  const Geom &g = pl->g;
  MultiP<T, T> P;
  P.u0 = (const T *)a.u0;
  P.v0 = (const T *)a.v0;
  P.uo = (T *)a.uo;
  P.vo = (T *)a.vo;
  P.fb_in = (const T *)a.fb_in;
  P.fb_out = (T *)a.fb_out;
  P.d_out = nullptr;
  if (sizeof(T) == 4 && !a.fb_is_f32) {
    P.d_out = (double *)a.fb_out;
    P.fb_out = nullptr;
  }
  P.cE = (const T *)g.coef[0];
  P.cN = (const T *)g.coef[1];
  P.ra = (const T *)g.coef[2];
  P.zrow = (const T *)pl->zero_row;
  P.nfb = pl->ring_nfb;
  P.mbits = g.mbits;
  P.lbits = (pl->n_land > 0) ? pl->lbits : nullptr;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  int np = ringc_zip_pairs(P.nwx, a.nbatch, nrows, S, nullptr);
  P.fold_rows = 0;
  P.nfw = 0;
  if (a.zip_fold) {   // the top rows: strips that start at the tripole seam, zipped with their mirror windows (one more "half pair" per window)
    P.nfw = (g.nx / 2 + WI - 1) / WI;
    // as many pairs as fill whole rounds of the 256 CUs together with the fold strips (two units per workgroup, pairs and fold strips mixed:
    // 257 workgroups would be two rounds -- config 4 measured 1.32 ms that way against 0.90)
    long long npmax = 0;
    for (long long k = 1; k <= 8 && npmax < 1; ++k) {
      const long long cap = 256 * k / std::max<long long>(1, std::min<long long>(a.nbatch, 256 * k));   // workgroups per field
      npmax = 2 * cap > P.nfw ? (2 * cap - P.nfw) / P.nwx : 0;                                        // (two units per workgroup)
    }
    np = (int)std::max(1LL, std::min<long long>(npmax, (nrows - S) / 4));
    // (at least S rows: the ghost rows the pairs below march beyond their last row must stay on this side of the seam)
    P.fold_rows = std::max(S, (int)((nrows + 2 * np) / (2 * np + 1)));
    np = (int)std::max(1LL, std::min<long long>(np, (nrows - P.fold_rows) / 4));
  }
  if (np < 1 || nrows - P.fold_rows < 2 * np) {
    set_error("k_ringcz: %d rows cannot be cut into pairs of strips", nrows);
    return GCMF_ERR_INVALID_ARG;
  }
  P.nstrips = 2 * np;
  P.H = (nrows - P.fold_rows + 2 * np - 1) / (2 * np);
  P.npack = 0;
  P.nwaves = P.nwx * P.nstrips;
  P.wrap = g.south_wrap && g.north_wrap;
  P.first = FIRST ? 1 : 0;
  P.last = a.last;
  P.area_weighted = 0;
  P.bstride = (long long)g.rows * g.nx;
  for (int t = 0; t < MAX_PK; ++t) P.pk[t] = t < S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  dim3 block(256), grid((P.nwx * np + P.nfw + 1) / 2, (unsigned)a.nbatch);
  P.xcd_per = pl->xcd_remap ? (int)(grid.x / 8) : 0;
  P.zigzag = 1;
  bool xe = true;
  ringc_zip_rows(std::max(P.H, P.fold_rows) + S + 1, S, &xe);
  if (pl->ringc_zip == 2) xe = true;    // (tuning: 2 = always the early-exit form, 3 = always whole periods)
  if (pl->ringc_zip == 3) xe = false;
  if (xe) hipLaunchKernelGGL((k_ringcz<T, S, FIRST, true>), grid, block, 0, s, P);
  else hipLaunchKernelGGL((k_ringcz<T, S, FIRST, false>), grid, block, 0, s, P);
  GCMF_HIP(hipGetLastError());
  note_kernel(pl, std::string("gcmf::k_ringcz<") + tyname<T>() + ", " + std::to_string(S) + ", " + (FIRST ? "true" : "false") + ", " + (xe ? "true" : "false") + ">", S,
              launch_geom(P.H, P.nstrips, P.nwx, P.xcd_per > 0, grid.x, grid.y, nrows));
  return GCMF_OK;
}

// k_ringc with the one mid-period exit (see ringc_march)
template <typename T, int KIND, int S, bool FIRST>
__global__ __launch_bounds__(256, 1) void k_ringc6(const MultiP<T, T> P) {
  int bx = blockIdx.x;
  if (P.xcd_per > 0 && bx < 8 * P.xcd_per) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= P.nwaves) return;
  ringc_walk<T, KIND, S, FIRST, false, true, false>(P, wid);
}

template <typename T, int KIND, int S, bool FIRST, bool XE = false, bool XE6 = false>
static int launch_ringc_sf(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;
  constexpr int R = RingGeom::R;
  const Geom &g = pl->g;
  MultiP<T, T> P;
  P.u0 = (const T *)a.u0;
  P.v0 = (const T *)a.v0;
  P.uo = (T *)a.uo;
  P.vo = (T *)a.vo;
  P.fb_in = (const T *)a.fb_in;   // the constant input f
  P.fb_out = (T *)a.fb_out;       // the result (last launch)
  P.d_out = nullptr;
  if (sizeof(T) == 4 && !a.fb_is_f32) {   // f32 state: the result is f64 unless the caller asked for f32 (GCMF_OUT_F32)
    P.d_out = (double *)a.fb_out;
    P.fb_out = nullptr;
  }
  P.cE = (const T *)g.coef[0];
  P.cN = (const T *)g.coef[1];
  P.ra = (const T *)g.coef[2];
  P.zrow = (const T *)pl->zero_row;
  P.nfb = pl->ring_nfb;
  P.mbits = g.mbits;
  P.lbits = (pl->n_land > 0) ? pl->lbits : nullptr;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  int H = pl->strip_rows;
  if (H <= 0) {
    constexpr int EXITP = (KIND == K_FLUX && !XE) ? (XE6 ? R / 2 : R) : 4;   // rows between two exits of the march
    const long long want = strips_per_column((long long)P.nwx * a.nbatch, nrows, S, EXITP);
    H = (int)((nrows + want - 1) / want);
    if (H < 4) H = 4;   // (short strips for small grids: see k_ring)
    if (KIND == K_FLUX && !XE) H += (EXITP - (H + 2 * S) % EXITP) % EXITP;   // whole (half) periods (no early exit here): let the padding carry real rows
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.nstrips = (nrows + H - 1) / H;
  P.npack = 0;
  if (!XE6 && a.nbatch > 1 && pl->strip_rows <= 0 && pl->pack_batch && (long long)a.nbatch * nrows < (1LL << 30)) {
    // Packed batch (see ringc_walk): as many runs per window as fill whole rounds of the 1024 wave slots, never longer than a field.
    // Taken when its rounds x (run + warm-up rows, one field boundary in most runs) beat the whole strips chosen above -- short grids
    // (the 300-row slab of one of 8 ranks, 16 fields: 626-651 -> 708-710 G; tools/measure_batched_scaling.py).
    constexpr int EXITP = (KIND == K_FLUX && !XE) ? (XE6 ? R / 2 : R) : 4;
    auto padded = [&](long long m) { return (m + EXITP - 1) / EXITP * EXITP; };
    const long long total = (long long)a.nbatch * nrows, slots = std::max(1LL, 1024LL / P.nwx);
    const long long rounds_u = ((long long)P.nwx * a.nbatch * P.nstrips + 1023) / 1024;
    const double cost_u = (double)(rounds_u * padded(H + 2 * S)) * (1.0 + 0.04 * (rounds_u - 1));
    double best = cost_u;
    long long best_q = 0, best_w = 0;
    for (long long k = 1; k <= 16; ++k) {
      const long long w = std::min(total, slots * k);                 // runs per window
      const long long q = (total + w - 1) / w;                       // rows per run
      if (q > nrows || q > 320) continue;   // (tall runs lose: 2400 x 3600 x 8 fields as 30 runs of 640 rows per window 768 G against 805 G as whole strips)
      const long long rounds = (w * P.nwx + 1023) / 1024;
      const bool crosses = (nrows % q) != 0;                          // (runs aligned with the fields cross nothing)
      const double cost = (double)(rounds * (padded(q + 2 * S) + (crosses ? padded(2 * S + EXITP / 2) : 0))) * (1.0 + 0.04 * (rounds - 1));
      if (cost < 0.97 * best) { best = cost; best_q = q; best_w = (total + q - 1) / q; }
      if (w >= total) break;
    }
    if (best_q > 0) {
      P.H = (int)best_q;
      P.nstrips = (int)best_w;
      P.npack = (int)a.nbatch;
    }
  }
  P.nwaves = P.nwx * P.nstrips;
  P.wrap = g.south_wrap && g.north_wrap;
  P.first = FIRST ? 1 : 0;
  P.last = a.last;
  P.area_weighted = (KIND == K_FLUX) ? 0 : g.area_weighted;
  P.bstride = (long long)g.rows * g.nx;
  for (int t = 0; t < MAX_PK; ++t) P.pk[t] = t < S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  dim3 block(256), grid((P.nwaves + 3) / 4, P.npack > 0 ? 1u : (unsigned)a.nbatch);
  P.xcd_per = pl->xcd_remap ? (int)(grid.x / 8) : 0;
  P.zigzag = pl->zigzag;
  if constexpr (!XE6) {
    if (P.npack > 0) {
      hipLaunchKernelGGL((k_ringcp<T, KIND, S, FIRST, XE>), grid, block, 0, s, P);
      GCMF_HIP(hipGetLastError());
      note_kernel(pl, std::string("gcmf::k_ringcp<") + tyname<T>() + ", " + std::to_string(KIND) + ", " + std::to_string(S) + ", " +
                          (FIRST ? "true" : "false") + ", " + (XE ? "true" : "false") + ">", S,
                  launch_geom(P.H, P.nstrips, P.nwx, P.xcd_per > 0, grid.x, grid.y, nrows));
      return GCMF_OK;
    }
  }
  if constexpr (XE) {
    hipLaunchKernelGGL((k_ringcs<T, S, FIRST>), grid, block, 0, s, P);
    GCMF_HIP(hipGetLastError());
    note_kernel(pl, std::string("gcmf::k_ringcs<") + tyname<T>() + ", " + std::to_string(S) + ", " + (FIRST ? "true" : "false") + ">", S,
                launch_geom(P.H, P.nstrips, P.nwx, P.xcd_per > 0, grid.x, grid.y, nrows));
    return GCMF_OK;
  }
  if constexpr (XE6) {
    hipLaunchKernelGGL((k_ringc6<T, KIND, S, FIRST>), grid, block, 0, s, P);
    GCMF_HIP(hipGetLastError());
    note_kernel(pl, std::string("gcmf::k_ringc6<") + tyname<T>() + ", " + std::to_string(KIND) + ", " + std::to_string(S) + ", " +
                        (FIRST ? "true" : "false") + ">", S, launch_geom(P.H, P.nstrips, P.nwx, P.xcd_per > 0, grid.x, grid.y, nrows));
    return GCMF_OK;
  }
  hipLaunchKernelGGL((k_ringc<T, KIND, S, FIRST>), grid, block, 0, s, P);
  GCMF_HIP(hipGetLastError());
  note_kernel(pl, std::string("gcmf::k_ringc<") + tyname<T>() + ", " + std::to_string(KIND) + ", " + std::to_string(S) + ", " +
                      (FIRST ? "true" : "false") + ">", S, launch_geom(P.H, P.nstrips, P.nwx, P.xcd_per > 0, grid.x, grid.y, nrows));
  return GCMF_OK;
}

// one stencil kind, one state type (their own translation units: gcmf_ringc_<kind>.hip = f64, gcmf_ringc_<kind>_f32.hip = f32 -- the
// instantiations of a kind compile for four to five minutes in one unit)
template <int KIND> static int launch_ringc_kind_f32(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  // f32 state, four cells per lane, the whole polynomial carried in f32 (f64 result unless GCMF_OUT_F32): the flux kinds since round 3,
  // the REGULAR / land-mask kinds since round 4
  switch (a.S) {
    case 5: return a.first ? launch_ringc_sf<float, KIND, 5, true>(pl, a, s) : launch_ringc_sf<float, KIND, 5, false>(pl, a, s);
    case 6: return a.first ? launch_ringc_sf<float, KIND, 6, true>(pl, a, s) : launch_ringc_sf<float, KIND, 6, false>(pl, a, s);
    case 7: return a.first ? launch_ringc_sf<float, KIND, 7, true>(pl, a, s) : launch_ringc_sf<float, KIND, 7, false>(pl, a, s);
    case 8:   // (never a first launch: clenshaw_cut starts an f32 filter with at most seven levels -- eight spill there)
      if (a.first) break;
      return launch_ringc_sf<float, KIND, 8, false>(pl, a, s);
  }
  set_error("k_ringc<float>: depth %d%s is not offered", a.S, a.first ? " as a first launch" : "");
  return GCMF_ERR_INVALID_ARG;
}

template <int KIND> static int launch_ringc_kind_f64(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 5: return a.first ? launch_ringc_sf<double, KIND, 5, true>(pl, a, s) : launch_ringc_sf<double, KIND, 5, false>(pl, a, s);
    case 6: return a.first ? launch_ringc_sf<double, KIND, 6, true>(pl, a, s) : launch_ringc_sf<double, KIND, 6, false>(pl, a, s);
    case 7: return a.first ? launch_ringc_sf<double, KIND, 7, true>(pl, a, s) : launch_ringc_sf<double, KIND, 7, false>(pl, a, s);
    case 8: return a.first ? launch_ringc_sf<double, KIND, 8, true>(pl, a, s) : launch_ringc_sf<double, KIND, 8, false>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
