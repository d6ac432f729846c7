// k_cgrid_ringf: the static-ring, LDS-direct form of the C-grid kernel (gcmf_cgrid_ring.hip) for the REFERENCE'S OWN SCHEME -- the forward
// three-term recurrence of reference filter.py:242-289 with f32 T_k and an f64 running sum (what NumPy >= 2 promotion makes of f32 fields,
// SURVEY 8a A2) -- i.e. Filter(evaluation="reference") on batched f32 levels (BASELINE config 5's opt-in figure).  Until round 6 that
// path ran round 2's k_cgrid_stream2<float, double, 2, 5> (281 G, plain loads, per-level nan_to_num, rolled rings).
//
// Same structure as k_cgrid_ring: a workgroup = 4 waves = 4 levels of the batch in lock-step on one (128-column window, strip) group, the
// 14 coefficient rows fetched once per workgroup into an LDS ring, one barrier per row; static rings over a period of 12 rows; packed
// f32; NaN masks per delivered row; +-inf / non-finite values born inside a launch send the workgroup to a redo pass with the full
// nan_to_num.  What differs:
//   * the level: Laplacian = CgLevel::feed<false> of gcmf_cgrid_stream2.hip operation for operation (no fused multiply-adds), then
//     T = 2 (-x - c L) - T_{k-2}, fbar += p_k T in f64 -- the same bits as k_cgrid_stream2<float, double, ...> and as single steps;
//   * per delivered row a wave takes T_{k-1} (u | v), T_{k-2} (u | v) and the running sum's row (f64: u, v) -- 4 KB per row and wave
//     instead of 3, so the coefficient rows are fetched ONE iteration ahead (they mostly come from the L2) while the state rows stay
//     two ahead: ring of S + 1 slots, 74 KB of LDS for five levels, two workgroups per CU;
//   * two results per level pair: T_{k+S-1} and T_{k+S-2} (the same row of levels S and S - 1, stored together), and the running sum;
//   * the running sum is updated IN PLACE by gcmf_apply (fbar_out = fbar_in), so a strip cannot simply be redone from its inputs: the
//     fast pass stops storing at the first iteration that sees a non-finite value, and the redo pass stores from exactly that iteration
//     on (a row of the sum depends on its own cell's history only, and the T planes are never written in place).
#include "gcmf_cgrid_ring_common.hpp"

namespace gcmf {

template <typename T> struct CRingFP {
  const T *u0, *v0;        // T_{k-1} (first launch: the input field)
  const T *up, *vp;        // T_{k-2} (first launch: unused)
  const double *fu, *fv;   // the running sum (first launch: unused)
  T *u1o, *v1o;            // T_{k+S-2} (unused by the last launch)
  T *u2o, *v2o;            // T_{k+S-1} (unused by the last launch)
  double *fuo, *fvo;       // the running sum; last launch: the result
  const T *coef[MAX_COEF];
  unsigned *redo;
  int nx, rows, out_lo, out_hi;
  int H, nwx, ngroups, nlev, nlevp, wrap, last;
  long long bstride;
  double p0, pk[8], c;
};

template <int S> struct CRingFGeom {
  static constexpr int M = S <= 4 ? 4 : 8;
  static constexpr int W = 128, WI = W - 2 * M;
  static constexpr int D = 2;                                          // state rows in flight per wave
  static constexpr int NS = S + 1;                                     // coefficient slots: S being read + the one being fetched
  static constexpr unsigned SLOTB = 14u * 512u;
  static constexpr unsigned STG_OFF = NS * SLOTB;
  static constexpr unsigned STGB = 4096u;                              // u0|v0, up|vp, fu (f64), fv (f64)
  static constexpr size_t lds_bytes() { return (size_t)STG_OFF + (unsigned)CR_WPB * D * STGB; }
  static constexpr int NDC = 8 / CR_WPB;
};

typedef double cr_d2 __attribute__((ext_vector_type(2)));

// a helper wave: its share of the coefficient rows, the barriers, nothing else
template <typename T, int S>
__device__ __forceinline__ void cgringf_helper(const CRingFP<T> &P, unsigned char *s_raw, const int lane, const int wv, const int pos0,
                                               const int r_begin, const int r_end, const int n_pad) {
  typedef CRingFGeom<S> G;
  constexpr int NS = G::NS, NDC = G::NDC, PPW = 2 * NDC;
  CRingCursor cur(P.nx, P.rows, P.wrap, r_begin, r_end, (unsigned)sizeof(T));
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)s_raw);
  const int half = lane >> 5;
  int c4 = (pos0 + 4 * (lane & 31)) % P.nx;
  if (c4 < 0) c4 += P.nx;
  const char *q_c[NDC];
  bool isa[NDC];
  unsigned pair_off[NDC];
#pragma unroll
  for (int h = 0; h < NDC; ++h) {
    const int pr = PPW * wv + 2 * h >= 14 ? 12 : PPW * wv + 2 * h;
    const int pa = pr + half;
    q_c[h] = reinterpret_cast<const char *>(P.coef[pa]) + (unsigned)c4 * 4u;
    isa[h] = pa < 7;
    pair_off[h] = (unsigned)pr * 512u;
  }
  unsigned nxt = 0;
  auto issue = [&]() {
    cur.advance();
#pragma unroll
    for (int h = 0; h < NDC; ++h) cr_dma16(q_c[h] + (isa[h] ? cur.ro : cur.rc), lds0 + nxt + pair_off[h]);
    nxt = (nxt + G::SLOTB == NS * G::SLOTB) ? 0u : nxt + G::SLOTB;
  };
  issue();
  for (int r = r_begin; r < r_begin + n_pad; ++r) {
    cr_wait_vm<0>();
    __syncthreads();
    issue();
  }
  cr_wait_vm<0>();
}

// One march of a strip by one wave.  SAN = false: the fast pass; stores stop at the first iteration that sees a non-finite value, which is
// returned (INT_MAX: none).  SAN = true: the redo pass; stores only from iteration r_from on.
template <typename T, int S, bool FIRST, bool SAN>
__device__ __forceinline__ int cgringf_march(const CRingFP<T> &P, unsigned char *s_raw, const int lane, const int wv, const long long boff,
                                             const unsigned colB, const int pos0, const bool keep, const int a, const int b, const int n_pad,
                                             const int r_from) {
  typedef typename CgV2<T>::type v2;
  typedef CRingFGeom<S> G;
  constexpr int NS = G::NS, D = G::D;
  constexpr int U = CR_U, RU = 6;
  constexpr int NDC = G::NDC, PPW = 2 * NDC;
  constexpr int NST = FIRST ? 1 : 4;   // state loads per iteration
  static_assert(S >= 2 && S + 1 <= U && U % D == 0 && U % RU == 0, "ring periods");
  static_assert(sizeof(T) == 4, "f32 state");
  const int nx = P.nx, rows = P.rows;
  const bool wrap = P.wrap, last = P.last;
  const T c = (T)P.c;
  const int r_begin = a - S, r_end = b + S;
  CRingCursor cur(nx, rows, wrap, r_begin, r_end, (unsigned)sizeof(T));

  const v2 Z = {T(0), T(0)};
  const cr_d2 ZD = {0.0, 0.0};
  v2 G0u[RU], G0v[RU];         // delivered rows of T_{k-1}; slot = (row - r_begin) mod RU
  v2 Vu, Vv;                   // the delivered row of T_{k-2}
  cr_d2 A0u, A0v;              // the delivered row of the running sum
  v2 Xu[S][3], Xv[S][3];       // X[m], m = 1 .. S - 1: rows of level m; slot = (iteration) mod 3 (all three are live: "-x" and T_{k-2} of the levels above)
  cr_d2 Au[S][2], Av[S][2];    // A[m], m = 1 .. S - 1: the running sum after level m, waiting one iteration for level m + 1
  v2 Lvt[S + 1][2], Lvh[S + 1][2], Luh[S + 1][2], LP[S + 1][2], LQ[S + 1][2], LR[S + 1][2];
  bool Ku0[U], Ku1[U], Kv0[U], Kv1[U];
#pragma unroll
  for (int l = 0; l < RU; ++l) G0u[l] = G0v[l] = Z;
  Vu = Vv = Z;
  A0u = A0v = ZD;
#pragma unroll
  for (int m = 0; m < S; ++m) {
#pragma unroll
    for (int l = 0; l < 3; ++l) Xu[m][l] = Xv[m][l] = Z;
    Au[m][0] = Au[m][1] = Av[m][0] = Av[m][1] = ZD;
  }
#pragma unroll
  for (int m = 0; m <= S; ++m) {
#pragma unroll
    for (int l = 0; l < 2; ++l) Lvt[m][l] = Lvh[m][l] = Luh[m][l] = LP[m][l] = LQ[m][l] = LR[m][l] = Z;
  }
#pragma unroll
  for (int l = 0; l < U; ++l) Ku0[l] = Ku1[l] = Kv0[l] = Kv1[l] = true;
  bool seen_inf = false;
  int r_stop = 0x7fffffff;

  unsigned slot_of[U];
#pragma unroll
  for (int l = 0; l < U; ++l) slot_of[l] = 0u;
  unsigned nxt_slot = 0u;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)s_raw);
  const int half = lane >> 5;
  int c4 = (pos0 + 4 * (lane & 31)) % nx;   // f32 rows: lanes 0..31 the u plane, four cells each, lanes 32..63 the v plane
  if (c4 < 0) c4 += nx;
  const unsigned c4B = (unsigned)c4 * 4u;
  int c2 = (pos0 + 2 * lane) % nx;          // f64 rows: all 64 lanes one plane, two cells each
  if (c2 < 0) c2 += nx;
  const unsigned c2B = (unsigned)c2 * 8u;
  const char *q_g0 = reinterpret_cast<const char *>((half ? P.v0 : P.u0) + boff) + c4B;
  const char *q_vp = reinterpret_cast<const char *>((FIRST ? (half ? P.v0 : P.u0) : (half ? P.vp : P.up)) + boff) + c4B;
  const char *q_fu = reinterpret_cast<const char *>((FIRST ? reinterpret_cast<const double *>(P.u0) : P.fu) + boff) + c2B;   // (first launch: never loaded)
  const char *q_fv = reinterpret_cast<const char *>((FIRST ? reinterpret_cast<const double *>(P.v0) : P.fv) + boff) + c2B;
  const char *q_c[NDC];
  bool isa[NDC];
  unsigned pair_off[NDC];
#pragma unroll
  for (int h = 0; h < NDC; ++h) {
    const int pr = PPW * wv + 2 * h >= 14 ? 12 : PPW * wv + 2 * h;
    const int pa = pr + half;
    q_c[h] = reinterpret_cast<const char *>(P.coef[pa]) + c4B;
    isa[h] = pa < 7;
    pair_off[h] = (unsigned)pr * 512u;
  }
  const unsigned stg0 = G::STG_OFF + (unsigned)(wv * D) * G::STGB;
  unsigned ro_c = 0u, rc_c = 0u;   // row offsets of the newest row whose STATE has been asked for: its coefficient rows go out next
  auto issue_state = [&](auto ph_c) {
    constexpr int ph = decltype(ph_c)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staging slot about to be refilled has been read out)
    cur.advance();
    ro_c = cur.ro;
    rc_c = cur.rc;
    const unsigned st = lds0 + stg0 + (unsigned)(ph % D) * G::STGB;
    cr_dma16(q_g0 + cur.ro, st);
    if constexpr (!FIRST) {
      cr_dma16(q_vp + cur.rc, st + 1024u);
      cr_dma16(q_fu + 2u * cur.rc, st + 2048u);   // (the f64 plane's rows are twice as long)
      cr_dma16(q_fv + 2u * cur.rc, st + 3072u);
    }
  };
  auto issue_coef = [&](auto ph_c) {   // the coefficient rows of the row issue_state asked for last
    constexpr int ph = decltype(ph_c)::value;
    slot_of[ph] = nxt_slot;
#pragma unroll
    for (int h = 0; h < NDC; ++h) cr_dma16(q_c[h] + (isa[h] ? ro_c : rc_c), lds0 + nxt_slot + pair_off[h]);
    nxt_slot = (nxt_slot + G::SLOTB == NS * G::SLOTB) ? 0u : nxt_slot + G::SLOTB;
  };

  v2 out_u = Z, out_v = Z;
  cr_d2 out_fu = ZD, out_fv = ZD;

  // level j of iteration r (phase ph): fed with row rho = r - j + 1 of level j - 1, produces row rho - 1 of level j
  auto level = [&](auto jj, auto ph_c) {
    constexpr int j = decltype(jj)::value;
    constexpr int ph = decltype(ph_c)::value;
    constexpr int kn = cmod(ph - (j - 1), U);
    constexpr int n3 = ph % 3, o3 = cmod(ph - 1, 3), t3 = cmod(ph - 2, 3);
    const v2 inu = (j == 1) ? G0u[ph % RU] : Xu[j >= 2 ? j - 1 : 1][n3];
    const v2 inv = (j == 1) ? G0v[ph % RU] : Xv[j >= 2 ? j - 1 : 1][n3];
    const v2 xu = (j == 1) ? G0u[cmod(ph - 1, RU)] : Xu[j >= 2 ? j - 1 : 1][o3];   // row rho - 1 of level j - 1: the "-x" term
    const v2 xv = (j == 1) ? G0v[cmod(ph - 1, RU)] : Xv[j >= 2 ? j - 1 : 1][o3];
    // T_{k-2}, row rho - 1: level j - 2's row of two iterations ago (level 1: the delivered row of T_{k-2}; level 2: the delivered rows)
    const v2 x2u = (j == 1) ? Vu : (j == 2) ? G0u[cmod(ph - 2, RU)] : Xu[j >= 3 ? j - 2 : 1][t3];
    const v2 x2v = (j == 1) ? Vv : (j == 2) ? G0v[cmod(ph - 2, RU)] : Xv[j >= 3 ? j - 2 : 1][t3];
    const cr_d2 fiu = (j == 1) ? A0u : Au[j >= 2 ? j - 1 : 1][cmod(ph - 1, 2)];
    const cr_d2 fiv = (j == 1) ? A0v : Av[j >= 2 ? j - 1 : 1][cmod(ph - 1, 2)];
    v2 su, sv;
    if constexpr (SAN) {
      su.x = cr_san(inu.x);  su.y = cr_san(inu.y);
      sv.x = cr_san(inv.x);  sv.y = cr_san(inv.y);
    } else {
      su.x = Ku0[kn] ? inu.x : T(0);  su.y = Ku1[kn] ? inu.y : T(0);
      sv.x = Kv0[kn] ? inv.x : T(0);  sv.y = Kv1[kn] ? inv.y : T(0);
    }
    v2 A[7], B[7];
    const v2 *cs = reinterpret_cast<const v2 *>(s_raw + slot_of[cmod(ph - (j - 1), U)]);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      A[q] = cs[q * 64 + lane];
      B[q] = cs[(7 + q) * 64 + lane];
    }
    constexpr int lo = cmod(ph - 1, 2), ln = ph % 2;
    const v2 vt_p = Lvt[j][lo], vh_p = Lvh[j][lo], uh_p = Luh[j][lo];
    // ---- CgLevel::feed<false> on the pair (gcmf_cgrid_stream2.hip): no fused multiply-adds ----
    const v2 ut = su * A[0], uh = su * A[1], vt = sv * A[2], vh = sv * A[3];
    v2 dut;  dut.x = ut.x - from_lower_lane0(ut.y);  dut.y = ut.y - ut.x;
    const v2 Pr = A[4] * dut - A[5] * (vt - vt_p);
    const v2 Qr = A[6] * Pr;
    v2 dvh;  dvh.x = vh_p.y - vh_p.x;  dvh.y = from_upper_lane0(vh_p.x) - vh_p.y;
    const v2 Rm = B[0] * dvh + B[1] * (uh - uh_p);
    const v2 Sm = B[2] * Rm;
    const v2 Pp = LP[j][lo];
    v2 dpp;  dpp.x = Pp.x - Pp.y;  dpp.y = Pp.y - from_upper_lane0(Pp.x);
    v2 dsm;  dsm.x = from_lower_lane0(Sm.y) - Sm.x;  dsm.y = Sm.x - Sm.y;
    const v2 lu = B[3] * dpp + B[4] * (LR[j][lo] - Rm);
    const v2 lv = B[5] * dsm - B[6] * (LQ[j][lo] - Qr);
    Lvt[j][ln] = vt;  Lvh[j][ln] = vh;  Luh[j][ln] = uh;
    LP[j][ln] = Pr;  LQ[j][ln] = Qr;  LR[j][ln] = Rm;
    // ---- the recurrence of filter.py:262-283 as k_cgrid_stream2 states it ----
    const v2 c2 = {c, c};
    const v2 avu = -xu - c2 * lu, avv = -xv - c2 * lv;
    const double pkj = P.pk[j - 1];
    v2 cu, cv;
    cr_d2 nu, nv;
    if constexpr (FIRST && j == 1) {
      cu = avu;
      cv = avv;
      nu.x = P.p0 * (double)xu.x + pkj * (double)avu.x;  nu.y = P.p0 * (double)xu.y + pkj * (double)avu.y;
      nv.x = P.p0 * (double)xv.x + pkj * (double)avv.x;  nv.y = P.p0 * (double)xv.y + pkj * (double)avv.y;
    } else {
      const v2 two = {T(2), T(2)};
      cu = two * avu - x2u;
      cv = two * avv - x2v;
      nu.x = fiu.x + pkj * (double)cu.x;  nu.y = fiu.y + pkj * (double)cu.y;
      nv.x = fiv.x + pkj * (double)cv.x;  nv.y = fiv.y + pkj * (double)cv.y;
    }
    if constexpr (j < S) {
      Xu[j][n3] = cu;
      Xv[j][n3] = cv;
      Au[j][ph % 2] = nu;
      Av[j][ph % 2] = nv;
    }
    if constexpr (j == S) { out_u = cu;  out_v = cv;  out_fu = nu;  out_fv = nv; }
  };

  auto phase = [&](auto ph_c, int r) {
    constexpr int ph = decltype(ph_c)::value;
    {
      // outstanding, oldest first: this iteration's state rows, this iteration's coefficient rows, the next iteration's state rows
      cr_wait_vm<NST>();
      __syncthreads();
      const unsigned char *st = s_raw + stg0 + (unsigned)(ph % D) * G::STGB;
      G0u[ph % RU] = reinterpret_cast<const v2 *>(st)[lane];
      G0v[ph % RU] = reinterpret_cast<const v2 *>(st + 512)[lane];
      if constexpr (!FIRST) {
        Vu = reinterpret_cast<const v2 *>(st + 1024)[lane];
        Vv = reinterpret_cast<const v2 *>(st + 1536)[lane];
        A0u = reinterpret_cast<const cr_d2 *>(st + 2048)[lane];
        A0v = reinterpret_cast<const cr_d2 *>(st + 3072)[lane];
      }
      issue_coef(cic<(ph + 1) % U>{});
      issue_state(cic<(ph + D) % U>{});
    }
    {
      const v2 gu = G0u[ph % RU], gv = G0v[ph % RU];
      if constexpr (!SAN) {
        Ku0[ph % U] = (gu.x == gu.x);  Ku1[ph % U] = (gu.y == gu.y);
        Kv0[ph % U] = (gv.x == gv.x);  Kv1[ph % U] = (gv.y == gv.y);
        seen_inf = seen_inf | (mabs(gu.x) > MLim<T>::big()) | (mabs(gu.y) > MLim<T>::big()) | (mabs(gv.x) > MLim<T>::big()) |
                   (mabs(gv.y) > MLim<T>::big());
      }
    }
    level(cic<1>{}, ph_c);
    if constexpr (S >= 2) level(cic<2>{}, ph_c);
    if constexpr (S >= 3) level(cic<3>{}, ph_c);
    if constexpr (S >= 4) level(cic<4>{}, ph_c);
    if constexpr (S >= 5) level(cic<5>{}, ph_c);
    if constexpr (S >= 6) level(cic<6>{}, ph_c);
    // ---- stores: row r - S of levels S (T_{k+S-1}, the running sum) and S - 1 (T_{k+S-2}: its row of one iteration ago) ----
    const int ju = r - S;
    if (ju >= a && ju < b) {   // wave-uniform
      const v2 xo_u = Xu[S - 1][cmod(ph - 1, 3)], xo_v = Xv[S - 1][cmod(ph - 1, 3)];   // T_{k+S-2}, row ju (also level S's "-x")
      if constexpr (!SAN) {
        // non-finite values born inside the launch show in level S's row as +-inf, or as a NaN in a cell whose T_{k+S-2} is not NaN
        const v2 tu = out_u - out_u, tv = out_v - out_v;
        seen_inf = seen_inf | ((tu.x != tu.x) & (xo_u.x == xo_u.x)) | ((tu.y != tu.y) & (xo_u.y == xo_u.y)) |
                   ((tv.x != tv.x) & (xo_v.x == xo_v.x)) | ((tv.y != tv.y) & (xo_v.y == xo_v.y));
        if (r_stop == 0x7fffffff && __any(seen_inf)) r_stop = r;   // nothing that saw it has been stored: stop here, the redo pass goes on from here
      }
      const bool store = SAN ? (r >= r_from) : (r < r_stop);
      if (keep && store) {
        const unsigned vo = colB + (unsigned)(ju * nx) * (unsigned)sizeof(T);
        *reinterpret_cast<cr_d2 *>(reinterpret_cast<char *>(P.fuo + boff) + 2 * (size_t)vo) = out_fu;
        *reinterpret_cast<cr_d2 *>(reinterpret_cast<char *>(P.fvo + boff) + 2 * (size_t)vo) = out_fv;
        if (!last) {
          *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.u2o + boff) + vo) = out_u;
          *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.v2o + boff) + vo) = out_v;
          *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.u1o + boff) + vo) = xo_u;
          *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.v1o + boff) + vo) = xo_v;
        }
      }
    }
  };

  issue_state(cic<0>{});
  issue_coef(cic<0>{});
  issue_state(cic<1>{});
  for (int r = r_begin; r < r_begin + n_pad; r += U) {
    phase(cic<0>{}, r);  phase(cic<1>{}, r + 1);  phase(cic<2>{}, r + 2);  phase(cic<3>{}, r + 3);
    phase(cic<4>{}, r + 4);  phase(cic<5>{}, r + 5);  phase(cic<6>{}, r + 6);  phase(cic<7>{}, r + 7);
    phase(cic<8>{}, r + 8);  phase(cic<9>{}, r + 9);  phase(cic<10>{}, r + 10);  phase(cic<11>{}, r + 11);
  }
  cr_wait_vm<0>();
  if (!SAN && r_stop == 0x7fffffff && __any(seen_inf)) r_stop = r_begin + n_pad;   // (seen after the last stored row: nothing left to redo)
  return r_stop;
}

template <typename T, int S, bool FIRST, int WPS>
__global__ __launch_bounds__(64 * CR_WPB, WPS) void k_cgrid_ringf(const CRingFP<T> P) {
  constexpr int WPB = CR_WPB;
  typedef CRingFGeom<S> G;
  constexpr int M = G::M, W = G::W, WI = G::WI;
  extern __shared__ __align__(16) unsigned char s_raw[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = blockIdx.x;
  const int xcd = blk & 7, slot = (blk >> 3) * WPB + wv;   // (the workgroups of a group follow each other on ONE XCD, as k_cgrid_ring)
  const int group = (slot / P.nlevp) * 8 + xcd;
  int lev = slot % P.nlevp;
  if (group >= P.ngroups) return;
  const bool shadow = lev >= P.nlev;
  if (shadow) lev = P.nlev - 1;
  const int wx = group % P.nwx, st = group / P.nwx;
  const int nx = P.nx;
  const int a = P.out_lo + st * P.H;
  const int b = min(a + P.H, P.out_hi);
  const long long boff = (long long)lev * P.bstride;
  const int pos0 = wx * WI - M;
  const int pos = pos0 + lane * 2;
  int col = pos % nx;
  if (col < 0) col += nx;
  const unsigned colB = (unsigned)col * (unsigned)sizeof(T);
  const bool keep = (lane * 2 >= M) && (lane * 2 < W - M) && (pos < nx);
  const int n_pad = ((b - a) + 2 * S + CR_U - 1) / CR_U * CR_U;
  int r_stop = 0x7fffffff;
  if (shadow) cgringf_helper<T, S>(P, s_raw, lane, wv, pos0, a - S, b + S, n_pad);
  else r_stop = cgringf_march<T, S, FIRST, false>(P, s_raw, lane, wv, boff, colB, pos0, keep, a, b, n_pad, 0);
  unsigned *s_flag = reinterpret_cast<unsigned *>(s_raw + G::STG_OFF);   // (the marches are over: the first staging word is free)
  __syncthreads();
  if (threadIdx.x == 0) *s_flag = 0u;
  __syncthreads();
  if (r_stop != 0x7fffffff && lane == 0) *s_flag = 1u;
  __syncthreads();
  const bool redo = *s_flag != 0u;
  __syncthreads();
  if (redo) {   // a non-finite value somewhere in the workgroup's rows: the strip again, nan_to_num in full at every level; every wave
    //             stores from where its own fast pass stopped (a wave that saw nothing has stored everything: it only keeps the barriers)
    if (threadIdx.x == 0 && P.redo) atomicAdd(P.redo, 1u);
    if (shadow) cgringf_helper<T, S>(P, s_raw, lane, wv, pos0, a - S, b + S, n_pad);
    else cgringf_march<T, S, FIRST, true>(P, s_raw, lane, wv, boff, colB, pos0, keep, a, b, n_pad, r_stop);
  }
}

static bool crf_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// batched f32 levels, f64 running sum, 4 or 5 levels, everything the 16-byte accesses touch aligned
bool cgrid_ringf_supported(const gcmf_plan *pl, const VecMultiArgs &a) {
  if (pl->cgrid_ring <= 0 || pl->kind != K_CGRID || pl->cgrid_tile || pl->d.dtype != GCMF_F32 || a.clen || a.fb_is_f32) return false;
  if (a.nbatch < 2 || a.S < 4 || a.S > 5) return false;
  if (pl->g.nx % 4 || pl->g.nx < 4 || pl->g.rows < a.S + 2) return false;
  if ((long long)pl->g.rows * pl->g.nx * 8 >= (1LL << 32)) return false;   // 32-bit byte offsets inside a level's f64 plane
  for (int k = 0; k < MAX_COEF; ++k)
    if (!crf_al16(pl->g.coef[k])) return false;
  for (int k = 0; k < 2; ++k)
    if (!crf_al16(a.u0[k]) || !crf_al16(a.uprev[k]) || !crf_al16(a.fb_in[k]) || !crf_al16(a.u1o[k]) || !crf_al16(a.u2o[k]) || !crf_al16(a.fb_out[k]))
      return false;
  return true;
}

template <typename T, int S, int WPS> static int launch_crf(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  typedef CRingFGeom<S> G;
  constexpr int WI = G::WI, WPB = CR_WPB;
  const Geom &g = pl->g;
  CRingFP<T> P;
  P.u0 = (const T *)a.u0[0];  P.v0 = (const T *)a.u0[1];
  P.up = (const T *)a.uprev[0];  P.vp = (const T *)a.uprev[1];
  P.fu = (const double *)a.fb_in[0];  P.fv = (const double *)a.fb_in[1];
  P.u1o = (T *)a.u1o[0];  P.v1o = (T *)a.u1o[1];
  P.u2o = (T *)a.u2o[0];  P.v2o = (T *)a.u2o[1];
  P.fuo = (double *)a.fb_out[0];  P.fvo = (double *)a.fb_out[1];
  for (int k = 0; k < MAX_COEF; ++k) P.coef[k] = (const T *)g.coef[k];
  P.redo = pl->ring_nfb;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  P.nlev = (int)a.nbatch;
  P.nlevp = (P.nlev + WPB - 1) / WPB * WPB;
  int H = pl->strip_rows;
  if (H <= 0) {   // (as launch_cr of gcmf_cgrid_ring.hip: the fewest strips of <= 96 rows fix the rounds, then the strip count fills the last one)
    const long long cap = 1024LL * WPS;
    const long long per_strip = (long long)P.nwx * P.nlevp, hmax = pl->cgrid_ring_hmax > 0 ? pl->cgrid_ring_hmax : 96;
    const long long ns_min = (nrows + hmax - 1) / hmax;
    const long long rounds = (ns_min * per_strip + cap - 1) / cap;
    long long ns = rounds * cap / per_strip;
    if (ns < ns_min) ns = ns_min;
    H = (int)((nrows + ns - 1) / ns);
    if (H < 16) H = 16;
    const long long nst = (nrows + H - 1) / H;
    H = (int)((nrows + nst - 1) / nst);
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.ngroups = P.nwx * ((nrows + H - 1) / H);
  P.wrap = g.south_wrap && g.north_wrap;
  P.last = a.last;
  P.bstride = (long long)g.rows * g.nx;
  P.p0 = a.p0;
  for (int t = 0; t < 8; ++t) P.pk[t] = a.pk[t];
  P.c = a.c;
  const long long groups_per_xcd = (P.ngroups + 7) / 8;
  const long long blocks_per_xcd = (groups_per_xcd * P.nlevp + WPB - 1) / WPB;
  dim3 block(64 * WPB), grid((unsigned)(blocks_per_xcd * 8));
  const size_t lds = G::lds_bytes();
  auto go = [&](auto kern, std::atomic<unsigned long long> &attr_set) -> int {   // (the attribute belongs to the function ON A DEVICE)
    const unsigned long long bit = 1ULL << (pl->d.device & 63);
    if (lds > 48 * 1024 && (pl->d.device >= 64 || !(attr_set.load(std::memory_order_relaxed) & bit))) {
      GCMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, P);
    return GCMF_OK;
  };
  static std::atomic<unsigned long long> set_first{0}, set_next{0};
  int rc = a.first ? go(&k_cgrid_ringf<T, S, true, WPS>, set_first) : go(&k_cgrid_ringf<T, S, false, WPS>, set_next);
  if (rc) return rc;
  note_kernel(pl, std::string("gcmf::k_cgrid_ringf<") + tyname<T>() + ", " + std::to_string(S) + ", " + (a.first ? "true" : "false") + ", " +
                      std::to_string(WPS) + ">", S, launch_geom(P.H, (nrows + H - 1) / H, P.nwx, 1, grid.x, grid.y, nrows));
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

int launch_cgrid_ringf(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 4: return launch_crf<float, 4, 2>(pl, a, s);
    case 5: return launch_crf<float, 5, 2>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
