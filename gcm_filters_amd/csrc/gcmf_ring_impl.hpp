// Deep temporal blocking (S = 5..8 recurrence steps per pass over HBM) with STATIC register rings: k_ring<T, FB, KIND, S>.
//
// Same algorithm, wave mapping, operands and arithmetic as k_scalar_multi<T, FB, KIND, S> / k_flux_multi2 (reference
// gcm_filters/kernels.py:113-121 (K_REG), 163-187 (K_MASKZ = the land-mask stencil on states whose land cells are zero),
// 259-315, 345-372, 402-429, 517-585 (K_FLUX) inside the recurrence of filter.py:162-212) -- results are bit-identical --
// rebuilt around what the SQ counters of those kernels showed (profiles/r02/cfg3_flux_multi2_sq_counters.txt,
// cfg2_maskz_sq_counters.txt: VALU-issue bound, only 28-38 % of the VALU instructions are arithmetic):
//
//   * NO WINDOW SHIFTING.  The per-level row windows (3 rows), the input rows (6), the T_{k-2} rows (4) and the lag lines of
//     the coefficient / mask rows and of fbar (12) are rings whose slot is (row index) mod the ring size; the row loop is
//     unrolled over the common period 12, so every slot index is a compile-time constant and nothing is ever moved.
//     Operands are loaded straight into the ring slot they are consumed from, THREE rows ahead (18 KB in flight per
//     wave): with the shifting gone the first version of this kernel (one row ahead) spent 46 % of its time in s_waitcnt
//     (profiles/r02/cfg3_flux_ring_v1_sq_counters.txt).
//   * K_FLUX: the south-face flux of a row IS the north-face flux of the row below (same operands, same rounding): it is
//     carried in a register per level instead of being recomputed (2 of 13 f64 operations per cell and level, and the
//     second lag of the north-face coefficients).
//   * NO NaN / inf BOOKKEEPING on this path.  nan_to_num is the identity on finite data, so the fast march only
//     watches for a non-finite value (one compare per cell of the LAST level -- a NaN / inf never becomes finite again on
//     this path, so every one that could reach a stored cell ends up there -- OR-ed into a wave mask, tested once per
//     unrolled body); a wave that sees one abandons its strip and redoes it with the general march of k_flux_multi2 /
//     k_scalar_multi, whose results then overwrite whatever the fast march stored (so fbar must not be accumulated in
//     place).  Land never enters the state (FIRST below), so ocean fields with NaN on land stay on the fast path.
//     K_REG has no nan_to_num in the reference (NaN spreads): no check, no fallback.
//   * the wave index goes through readfirstlane: row indices, pointers and loop bounds are scalar; neighbours by DPP
//     with zero fill (no copy of the source); rows beyond a closed boundary read their coefficients from a row of zeros.
//
//   * FIRST = true is the first launch of a filter (T_0 = the caller's field, no T_{k-2} / fbar yet, fbar = p_0 T_0 + p_1 T_1,
//     filter.py:192-199): isolated (land) cells are taken as zero while the field is loaded, so NaN on land never enters
//     the state and gcmf_apply needs no k_zero_land pass; k_land_fix writes those cells' own polynomial at the end.
//     prepare() of the area-weighted types (field * area, kernels.py:100-101) is applied as the field is loaded.
//   * all rings start at zero: the upper levels of a strip's first rows run on slots no load has filled yet; what they
//     produce is never stored but the NaN watch sees it, and stale NaNs left in the registers by an earlier kernel would
//     send clean strips to the general march.
//   * workgroups are renumbered so that every XCD owns a contiguous range of strips (one L2 for shared columns / halo rows).
//
// One wave per SIMD (up to 446 registers at S = 8 in f64).  gcmf_ring_fallbacks() counts the strips that were redone.
#pragma once
#include "gcmf_flux_multi2_body.hpp"
#include "gcmf_scalar_multi_impl.hpp"

namespace gcmf {

template <int N> using ic = std::integral_constant<int, N>;
constexpr int pmod(int a, int m) { return ((a % m) + m) % m; }

struct RingGeom {
  static constexpr int D = 3;   // rows of operands in flight
  static constexpr int R = 12;  // lag-ring slots (>= S + D) == unroll factor of the row loop == common period of all rings
  static constexpr int RU = 6;  // ring of the input rows T_{k-1}: 3 live (levels 1, 2) + D in flight
  static constexpr int RV = 4;  // ring of the T_{k-2} rows: 1 live (level 1) + D in flight
};

// (scalar row pointer) + (32-bit byte offset of the lane): the form the global_load `saddr` addressing mode takes
template <typename X> __device__ __forceinline__ const X *lane_ptr(const X *rowp, unsigned bytes) {
  return reinterpret_cast<const X *>(reinterpret_cast<const char *>(rowp) + bytes);
}
template <typename X> __device__ __forceinline__ X *lane_ptr(X *rowp, unsigned bytes) {
  return reinterpret_cast<X *>(reinterpret_cast<char *>(rowp) + bytes);
}

// VEC: cells per lane -- 16 bytes' worth by default; the f32 flux kinds run with TWO (8-byte accesses): with four their rings need 510
// registers (round 3: slower than k_flux_multi2), with two they need what the f64 kernel needs.
template <typename T, typename FB, int KIND, int S, bool FIRST, int VEC = 16 / (int)sizeof(T)>
__global__ __launch_bounds__(256, 1) void k_ring(const MultiP<T, FB> P) {
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;
  constexpr int R = RingGeom::R, D = RingGeom::D, RU = RingGeom::RU, RV = RingGeom::RV;
  static_assert(R >= S + D && R % 3 == 0 && R % RU == 0 && R % RV == 0 && RU >= 3 + D && RV >= 1 + D, "ring periods");
  static_assert(KIND == K_REG || KIND == K_MASKZ || KIND == K_FLUX, "stencil kind");
  constexpr bool FLUX = (KIND == K_FLUX), MASK = (KIND == K_MASKZ);
  constexpr bool SAN = (KIND != K_REG);  // the reference sanitises with nan_to_num (kernels.py:165, 298): watch for non-finite values
  constexpr bool FUSED = FLUX;           // gcmf_recurrence.hpp

  const int lane = threadIdx.x & 63;
  // the wave index is uniform, but the compiler cannot prove it of threadIdx.x >> 6: without the readfirstlane every row
  // index, pointer and loop bound derived from it lives in vector registers and is recomputed with vector instructions
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8) and every XCD has its own L2.  Give each XCD a
  // CONTIGUOUS range of strips instead of every eighth workgroup, so that the columns two neighbouring windows share and
  // the halo rows two neighbouring strips share are fetched through the same L2.
  int bx = blockIdx.x;
  if (P.xcd_per > 0 && bx < 8 * P.xcd_per) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= P.nwaves) return;
  const int wx = wid % P.nwx, st = wid / P.nwx;
  const int nx = P.nx, rows = P.rows;
  const int a = P.out_lo + st * P.H;
  const int b = min(a + P.H, P.out_hi);
  const long long boff = (long long)blockIdx.y * P.bstride;
  const int pos = wx * WI - M + lane * VEC;
  int col_s = pos % nx;
  if (col_s < 0) col_s += nx;
  // unsigned: a zero-extended 32-bit lane offset lets the loads take (scalar row pointer + vector offset) addressing
  const unsigned col = (unsigned)col_s;
  const unsigned colT = col * (unsigned)sizeof(T), colF = col * (unsigned)sizeof(FB);  // byte offsets of the lane's first cell
  const bool keep = (lane * VEC >= M) && (lane * VEC < W - M) && (pos < nx);
  const T c = (T)P.c;
  const bool last = P.last;

  // ---- register-resident state: rings indexed by compile-time slots ----
  T G0[RU][VEC];     // rows of the input T_{k-1}, slot = (row - r_begin) mod RU
  T G[S][3][VEC];    // G[t], t = 1..S-1: rows of T_{k-1+t}, slot = (row - r_begin) mod 3   (G[0] unused)
  T FN[S + 1][VEC];  // K_FLUX: FN[t] = north-face flux of the row level t worked on in the previous iteration
  T cE[R][VEC], cN[R][VEC], ra[R][VEC];  // K_FLUX: coefficient rows, slot = (row + 1 - r_begin) mod R
  unsigned B[R];     // K_MASKZ: mask bytes of the lane's cells (bit 0 wet, bits 5-7 wet-neighbour count), same slots
  FB F[R][VEC];      // fbar rows, same slots
  T V[RV][VEC];      // T_{k-2} rows, slot = (row + 1 - r_begin) mod RV
  unsigned Z[RU];    // FIRST: the "exchanges with a neighbour" bytes of the input rows (same slots as G0)
  T AR[RU][VEC];     // FIRST, area-weighted types: the area of the input rows (prepare(): T_0 = field * area)
#pragma unroll
  for (int t = 0; t < S; ++t) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) G[t][0][k] = G[t][1][k] = G[t][2][k] = FN[t + 1][k] = T(0);
  }
#pragma unroll
  for (int l = 0; l < RU; ++l) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) G0[l][k] = AR[l][k] = T(0);
    Z[l] = 0u;
  }
  // The first S iterations run their upper levels on ring slots no load has filled yet.  What they produce is never stored,
  // but the NaN watch sees it: registers start with whatever the previous kernel on this SIMD left in them, and 0 x (a stale
  // NaN taken as a coefficient) would send a clean strip to the general march.  (Measured: every strip of the first launch
  // after a kernel that had NaNs in flight.)  Zeros cost ~100 moves per wave, once.
#pragma unroll
  for (int l = 0; l < R; ++l) {
    B[l] = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      cE[l][k] = cN[l][k] = ra[l][k] = T(0);
      F[l][k] = FB(0);
    }
  }
#pragma unroll
  for (int l = 0; l < RV; ++l) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) V[l][k] = T(0);
  }

  // Row cursor: the rows are visited in sequence, so the (wrapped / clamped) row index of the row being loaded is kept
  // incrementally in scalar registers -- periodic wrap, or clamp beyond a closed boundary (such rows are `outside`: no
  // flux, no wet cells).  Written as selects: a branch per row costs this single-wave kernel more than the arithmetic.
  const bool wrap = P.wrap;
  const int r_begin = a - S, r_last = b + S - 1;  // rows of T_{k-1} this strip needs: [a-S, b+S)
  int cj, cj_prev;          // row index (in [0, rows)) of the next T_{k-1} row to load, and of the one before it
  bool cout_, cout_prev;    // ... lies beyond a closed boundary
  {
    const int r = r_begin - 1;  // the centre operands of the first iteration belong to row r_begin - 1
    const bool lo = r < 0, hi = r >= rows;
    cj = wrap ? (r + (lo ? rows : 0) - (hi ? rows : 0)) : (lo ? 0 : (hi ? rows - 1 : r));
    cout_ = !wrap && (lo || hi);
    cj_prev = cj;
    cout_prev = cout_;
  }
  int cr = r_begin - 1;     // unwrapped row of the cursor
  auto advance = [&]() {    // cursor -> next row
    cj_prev = cj;
    cout_prev = cout_;
    ++cr;
    const int jn = cj + ((wrap || cr > 0) ? 1 : 0);
    const bool hit = (jn >= rows);
    cj = hit ? (wrap ? 0 : rows - 1) : jn;
    cout_ = !wrap && (cr < 0 || cr >= rows);
  };
  const bool has_land = FIRST && P.lbits != nullptr;
  const uint8_t *zbase = has_land ? P.lbits : reinterpret_cast<const uint8_t *>(P.u0);  // (bytes of a valid plane, ignored)
  const bool weigh = FIRST && !FLUX && P.area_weighted;  // prepare() of the area-weighted types (kernels.py:100-101)
  const T *abase = weigh ? P.area : P.u0;                // (same trick: an unconditional load, ignored when there is no area)
  // addresses = a wave-uniform row pointer (scalar arithmetic) + this lane's column.  Loads past the strip's last row are
  // harmless (a valid row of the plane; what they feed is never stored), so the march needs no clamp of its own.
  auto load_u = [&](auto slot_c) {  // the cursor's row of T_{k-1}
    constexpr int sl = decltype(slot_c)::value;
    const T *rowp = P.u0 + boff + (long long)(cj * nx);
    mload<T, VEC>(G0[sl], lane_ptr(rowp, colT));
    if constexpr (FIRST) {  // the caller's field: land leaves the state as it is loaded (see `phase`)
      // unconditional load (a plan without land reads the field's own bytes and ignores them): a load under a wave-uniform
      // `if` has produced mis-timed waits in this code base before (compiler notes)
      const uint8_t *zp = lane_ptr(zbase + (long long)(cj * nx), col);
      if (VEC == 2) Z[sl] = *reinterpret_cast<const unsigned short *>(zp);
      else Z[sl] = *reinterpret_cast<const unsigned *>(zp);
      if constexpr (!FLUX) mload<T, VEC>(AR[sl], lane_ptr(abase + (long long)(cj * nx), colT));
    }
  };
  // the centre-only operands that travel with that row: T_{k-2}, fbar, coefficients / mask bits of the row before it
  auto load_centre = [&](auto slot_c, auto vslot_c) {
    constexpr int sl = decltype(slot_c)::value;
    constexpr int vs = decltype(vslot_c)::value;
    const bool out_c = cout_prev;
    const long long rc = (long long)(cj_prev * nx);
    if constexpr (FLUX) {  // beyond a closed boundary: no flux -- the coefficients come from a row of zeros
      const T *pE = out_c ? P.zrow : P.cE + rc;
      const T *pN = out_c ? P.zrow : P.cN + rc;
      const T *pA = out_c ? P.zrow : P.ra + rc;
      mload<T, VEC>(cE[sl], lane_ptr(pE, colT));
      mload<T, VEC>(cN[sl], lane_ptr(pN, colT));
      mload<T, VEC>(ra[sl], lane_ptr(pA, colT));
    }
    if constexpr (MASK) {  // beyond a closed boundary: land (bits 0)
      const uint8_t *mp = lane_ptr(out_c ? (const uint8_t *)P.zrow : P.mbits + rc, col);
      if (VEC == 2) B[sl] = *reinterpret_cast<const unsigned short *>(mp);
      else B[sl] = *reinterpret_cast<const unsigned *>(mp);
    }
    if constexpr (!FIRST) {  // the first launch of a filter has no T_{k-2} and no fbar yet
      const FB *pF = P.fb_in + boff + rc;
      const T *pV = P.v0 + boff + rc;
      mload<FB, VEC>(F[sl], lane_ptr(pF, colF));
      mload<T, VEC>(V[vs], lane_ptr(pV, colT));
    }
  };

  bool bad = false;         // this lane met a non-finite value (input row or a produced level)
  T out_v[VEC], out_u[VEC]; // outputs of levels S-1 and S of the current iteration

  // level t of the iteration with phase ph: row r - t of T_{k-1+t} from rows r-t-1 .. r-t+1 of level t-1
  auto level = [&](auto tt, auto ph_c) {
    constexpr int t = decltype(tt)::value;
    constexpr int ph = decltype(ph_c)::value;
    constexpr int sS = pmod(ph - t - 1, 3), sC = pmod(ph - t, 3), sN = pmod(ph - t + 1, 3);
    constexpr int sl = pmod(ph - t + 1, R);
    const T(&gS)[VEC] = (t == 1) ? G0[pmod(ph - 2, RU)] : G[t >= 2 ? t - 1 : 1][sS];
    const T(&gC)[VEC] = (t == 1) ? G0[pmod(ph - 1, RU)] : G[t >= 2 ? t - 1 : 1][sC];
    const T(&gN)[VEC] = (t == 1) ? G0[pmod(ph, RU)] : G[t >= 2 ? t - 1 : 1][sN];
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
          const T wf = (T)(bb >> 5);  // wet-neighbour count, precomputed in bits 5-7
          L = -wf * xC + xE;
          L = L + xW;
          L = L + gN[k];
          L = L + gS[k];
          L = (bb & 1u) ? L : T(0);
        } else {
          L = T(-4) * xC + xE;
          L = L + xW;
          L = L + gN[k];
          L = L + gS[k];
        }
      }
      const T av = cheb_a<FUSED>(xC, c, L);
      T tk;
      if constexpr (FIRST && t == 1) {  // T_1 = A(T_0), fbar = p_0 T_0 + p_1 T_1 (filter.py:192-199)
        tk = av;
        F[sl][k] = cheb_acc_first<FUSED, T, FB>(P.p0, P.pk[0], xC, av);
      } else {
        const T x2 = (t == 1) ? V[ph % RV][k] : (t == 2 ? G0[pmod(ph - 2, RU)][k] : G[t >= 3 ? t - 2 : 1][sC][k]);
        tk = cheb_t<FUSED>(av, x2);
        F[sl][k] = cheb_acc<FUSED, T, FB>(F[sl][k], P.pk[t - 1], tk);
      }
      if (t < S) G[t < S ? t : 0][sC][k] = tk;
      if (t == S - 1) out_v[k] = tk;
      if (t == S) {
        out_u[k] = tk;
        // A non-finite value never becomes finite again on this path: "-x" carries it to the same cell of the next level
        // (and T_{k-2} to the one after), the stencil spreads it to the neighbours.  So whatever NaN / inf could reach a
        // stored cell -- from T_{k-1}, T_{k-2} or any level -- shows up in level S of that cell: one test per cell
        // and row instead of one per cell, row and level.
        if (SAN) bad = bad || !(mabs(tk) <= MLim<T>::big());
      }
    }
  };

  // one row iteration: row r of T_{k-1} has been delivered into G0[ph % RU], its centre operands into ring slots ph
  auto phase = [&](auto ph_c, int r) {
    constexpr int ph = decltype(ph_c)::value;
    // the operands of iteration + D; every slot they land in was released in the previous iteration at the latest
    advance();
    load_centre(ic<(ph + D) % R>{}, ic<(ph + D) % RV>{});
    load_u(ic<(ph + D) % RU>{});
    if constexpr (FIRST) {
      // Land out of the state from the start: a cell that exchanges nothing with its neighbours (land under a wet mask, a
      // flux-form cell with four closed faces) is taken as zero -- it has L = 0 and evolves on its own; k_land_fix writes
      // its polynomial into the result at the end.  NaN on land therefore never enters this kernel.
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (!FLUX) G0[ph % RU][k] = weigh ? G0[ph % RU][k] * AR[ph % RU][k] : G0[ph % RU][k];
        G0[ph % RU][k] = (!has_land || ((Z[ph % RU] >> (8 * k)) & 1u)) ? G0[ph % RU][k] : T(0);
      }
    }
    level(ic<1>{}, ph_c);
    if constexpr (S >= 2) level(ic<2>{}, ph_c);
    if constexpr (S >= 3) level(ic<3>{}, ph_c);
    if constexpr (S >= 4) level(ic<4>{}, ph_c);
    if constexpr (S >= 5) level(ic<5>{}, ph_c);
    if constexpr (S >= 6) level(ic<6>{}, ph_c);
    if constexpr (S >= 7) level(ic<7>{}, ph_c);
    if constexpr (S >= 8) level(ic<8>{}, ph_c);
    // stores: T_{k-1+S} row r-S, fbar row r-S (its ring slot is complete), T_{k-2+S} row r-S+1
    const int ju = r - S;
    if (ju >= a && ju < b) {  // wave-uniform
      const long long off = boff + (long long)ju * nx;
      if (keep) {
        constexpr int fs = pmod(ph - S + 1, R);
        if (!last) {
          mstore<T, VEC>(lane_ptr(P.uo + off, colT), out_u);
        } else if (!FLUX && P.area_weighted) {  // finalize(): / area (kernels.py:103-104)
          T ar[VEC];
          mload<T, VEC>(ar, lane_ptr(P.area + (long long)ju * nx, colT));
#pragma unroll
          for (int k = 0; k < VEC; ++k) F[fs][k] = F[fs][k] / (FB)ar[k];
        }
        mstore<FB, VEC>(lane_ptr(P.fb_out + off, colF), F[fs]);
      }
    }
    const int jv = r - S + 1;
    if (!last && jv >= a && jv < b) {
      T *rowp = P.vo + boff + (long long)jv * nx;
      if (keep) mstore<T, VEC>(lane_ptr(rowp, colT), out_v);
    }
  };

  // ---- march north; the body covers one full period of every ring ----
  advance();  // cursor on r_begin
  load_centre(ic<0>{}, ic<0>{});
  load_u(ic<0>{});
  advance();
  load_centre(ic<1>{}, ic<1>{});
  load_u(ic<1>{});
  advance();
  load_centre(ic<2>{}, ic<2>{});
  load_u(ic<2>{});
  static_assert(D == 3, "prologue");
  // The last period is left after the strip's last row (a wave-uniform scalar branch per row): a short strip -- the slab of one
  // rank of an 8-GPU run marches 11 + 2 S rows -- does not pay for rows up to the next multiple of the period.
  bool dirty = false;
  for (int r0 = r_begin;; r0 += R) {
    bool done = true;
    do {
      phase(ic<0>{}, r0);
      phase(ic<1>{}, r0 + 1);
      phase(ic<2>{}, r0 + 2);
      phase(ic<3>{}, r0 + 3);
      if (r0 + 3 >= r_last) break;     // (exits every four rows: one per row costs the kernel 100 registers more)
      phase(ic<4>{}, r0 + 4);
      phase(ic<5>{}, r0 + 5);
      phase(ic<6>{}, r0 + 6);
      phase(ic<7>{}, r0 + 7);
      if (r0 + 7 >= r_last) break;
      phase(ic<8>{}, r0 + 8);
      phase(ic<9>{}, r0 + 9);
      phase(ic<10>{}, r0 + 10);
      phase(ic<11>{}, r0 + 11);
      done = (r0 + 11 >= r_last);
    } while (false);
    if (SAN && __any(bad)) {  // wave-uniform: a NaN / inf somewhere in this strip -> the general march redoes the strip
      dirty = true;
      break;
    }
    if (done) break;
  }
  if constexpr (SAN) {
    if (dirty) {
      if (P.nfb && (threadIdx.x & 63) == 0) atomicAdd(P.nfb, 1u);  // instrumentation: gcmf_ring_fallbacks
      // the general march reads P.first itself; a first launch still has land in its input: K_MASK, not K_MASKZ
      if constexpr (FLUX) flux_multi2_march<T, FB, S, VEC>(P, wid);   // the same strip and window: wid, not blockIdx (XCD order above)
      else scalar_multi_march<T, FB, (FIRST ? K_MASK : KIND), S, 1>(P, wid);
      if constexpr (FIRST) {
        // the general march carries land through the recurrence; the launches that follow were promised states whose
        // isolated cells are zero (the fast march above took them as zero, gcmf_apply then skips k_zero_land): zero them
        // in the two states of this strip (same lanes, same cells as the stores of the march: program order holds)
        if (P.lbits && !last && keep) {
          for (int j = a; j < b; ++j) {
            const long long off = boff + (long long)j * nx;
            const uint8_t *zp = lane_ptr(P.lbits + (long long)j * nx, col);
            const unsigned zb = (VEC == 2) ? (unsigned)*reinterpret_cast<const unsigned short *>(zp)
                                           : *reinterpret_cast<const unsigned *>(zp);
#pragma unroll
            for (int k = 0; k < VEC; ++k)
              if (!((zb >> (8 * k)) & 1u)) {
                lane_ptr(P.uo + off, colT)[k] = T(0);
                lane_ptr(P.vo + off, colT)[k] = T(0);
              }
          }
        }
      }
    }
  }
}

template <typename T, typename FB, int KIND, int S, bool FIRST, int VEC = 16 / (int)sizeof(T)>
static int launch_ring_sf(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  constexpr int W = 64 * VEC;
  constexpr int M = (S + VEC - 1) / VEC * VEC;
  constexpr int WI = W - 2 * M;
  constexpr int R = RingGeom::R;
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
  P.zrow = (const T *)pl->zero_row;
  P.nfb = pl->ring_nfb;
  P.mbits = g.mbits;
  P.lbits = (FIRST && pl->n_land > 0) ? pl->lbits : nullptr;
  P.area = (const T *)g.area;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  int H = pl->strip_rows;
  if (H <= 0) {  // one resident round of waves at one wave per SIMD (all strips march in lock-step)
    // (tripolar plans: k_fold_band runs beside this launch on a side stream; its 48-register waves fit on the SIMDs next to these)
    const long long want = strips_per_column((long long)P.nwx * a.nbatch, nrows, S, 4);
    H = (int)((nrows + want - 1) / want);
    // (no rounding to the ring period: the march leaves its last period early, so the slab of one rank of an 8-GPU run marches
    // 11 + 2 S rows, not 36.  Small grids get short strips -- a launch takes as long as ONE strip's march, and a grid that
    // leaves wave slots idle lives in L2 anyway: 512x512 at H = 4 runs a filter in 26 us, at H = 16 in 40 us)
    if (H < 4) H = 4;
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.nstrips = (nrows + H - 1) / H;
  P.nwaves = P.nwx * P.nstrips;
  P.wrap = g.south_wrap && g.north_wrap;
  P.first = FIRST ? 1 : 0;
  P.last = a.last;
  P.area_weighted = (KIND == K_FLUX) ? 0 : g.area_weighted;
  P.bstride = (long long)g.rows * g.nx;
  for (int t = 0; t < MAX_PK; ++t) P.pk[t] = t < S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  dim3 block(256), grid((P.nwaves + 3) / 4, (unsigned)a.nbatch);
  P.xcd_per = pl->xcd_remap ? (int)(grid.x / 8) : 0;
  P.zigzag = 0;
  hipLaunchKernelGGL((k_ring<T, FB, KIND, S, FIRST, VEC>), grid, block, 0, s, P);
  GCMF_HIP(hipGetLastError());
  note_kernel(pl, std::string("gcmf::k_ring<") + tyname<T>() + ", " + tyname<FB>() + ", " + std::to_string(KIND) + ", " +
                      std::to_string(S) + ", " + (FIRST ? "true" : "false") + (VEC == 16 / (int)sizeof(T) ? "" : ", " + std::to_string(VEC)) + ">", S,
              launch_geom(P.H, P.nstrips, P.nwx, P.xcd_per > 0, grid.x, grid.y, nrows));
  return GCMF_OK;
}

template <typename T, typename FB, int KIND, int S, int VEC> static int launch_ring_s(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  return a.first ? launch_ring_sf<T, FB, KIND, S, true, VEC>(pl, a, s) : launch_ring_sf<T, FB, KIND, S, false, VEC>(pl, a, s);
}

template <typename T, typename FB, int KIND, int VEC = 16 / (int)sizeof(T)> static int launch_ring_k(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 5: return launch_ring_s<T, FB, KIND, 5, VEC>(pl, a, s);
    case 6: return launch_ring_s<T, FB, KIND, 6, VEC>(pl, a, s);
    case 7: return launch_ring_s<T, FB, KIND, 7, VEC>(pl, a, s);
    case 8: return launch_ring_s<T, FB, KIND, 8, VEC>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

// one translation unit per stencil kind (compile time): the dtype dispatch
template <int KIND> static int launch_ring_kind(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_ring_k<double, double, KIND>(pl, a, s);
  if constexpr (KIND == K_FLUX) {
    // f32 state of the flux kinds: its own translation unit (gcmf_ring_flux_f32.hip, two cells per lane)
    return launch_ring_flux_f32(pl, a, s);
  } else {
    if (a.fb_is_f32) return launch_ring_k<float, float, KIND>(pl, a, s);
    return launch_ring_k<float, double, KIND>(pl, a, s);
  }
}

}  // namespace gcmf
