// k_cgrid_ring: the C-grid vector Laplacian (reference gcm_filters/kernels.py:630-696) inside the backward (Clenshaw) evaluation of
// the filter polynomial (reference filter.py:217-291 restated as in gcmf_ringc_impl.hpp), S levels per pass over HBM, for batched
// levels (BASELINE config 5: 50 levels of 2400 x 3600, f32).  Round 5 rebuild of k_cgrid_stream2c around what its instruction stream
// showed (440 VALU instructions per row at S = 4 of which 130 are arithmetic: 138 register moves, 117 compare / select of the
// per-level nan_to_num, 28 address computations):
//
//   * same work split: a workgroup = 4 waves = 4 levels of the batch marching one (window, strip) group in lock-step, the 14 coefficient
//     rows fetched once per workgroup (a quarter per wave) and handed round through an LDS ring, one barrier per row; a wave owns 128
//     columns (two cells per lane, 8-byte accesses) and time-skews S levels along y, level j one row behind level j - 1;
//   * STATIC RINGS: every row that outlives an iteration (level outputs: 3 slots, operand rows in flight, the rows of f, the LDS slots)
//     lives in a ring whose slot is (row) mod (ring size); the row loop is unrolled over the common period 12, so every slot is a
//     compile-time register / LDS offset and nothing is ever moved;
//   * PACKED f32: the two cells of a lane are one <2 x float>, every multiply / add / fma is a v_pk_* instruction; only the four
//     x-differences per level need the neighbour lane (DPP);
//   * nan_to_num WITHOUT per-level compares: a NaN never leaves its cell and never becomes finite (the "-x", b_{k+2} and p_k f terms
//     keep it), so the set of NaN cells of a row is the same at every level: it is taken ONCE per row from the delivered state row
//     (lane masks in scalar registers) and every level zeroes its stencil operands with one select each;
//   * +-inf (nan_to_num clamps it to +-FLT_MAX in the stencil, kernels.py:651-652) is only WATCHED on the delivered rows; a workgroup
//     that meets one redoes its strip with the full nan_to_num at every level (SAN = true; same arithmetic);
//   * scalar addressing: row pointers in SGPRs, one 32-bit lane offset (global_load ... saddr).
//   * LDS-DIRECT LOADS (global_load_lds_dwordx4, gfx950): every operand row (u|v of b_{k+1}, of d_{k+1}, of f) and the wave's share of the
//     coefficient rows go from memory straight into LDS, 16 bytes per lane, with explicit s_waitcnt vmcnt -- no registers in flight, which
//     is what lets FIVE levels run at two waves per SIMD.
//
//   * the recurrence in REINSCH'S FORM (state planes (b_{k+1}, d_{k+1}), d_k = b_k + b_{k+1}: see gcmf_cgrid_stream2.hip): more accurate in
//     f32 than the reference's own f32 path.  A level needs d of the level before (one more live row per level than b_{k+2} took), so
//     levels 1 .. S - 1 do NOT carry the three scaled copies of their previous row from iteration to iteration: they rebuild them from
//     the raw row (which the recurrence keeps anyway), its NaN masks and the coefficient slot of the row before -- the same operands, the
//     same bits -- which is what lets five levels stay within 256 registers.
//
// Arithmetic per level = CgLevel::feed<true> of gcmf_cgrid_stream2.hip operation for operation, so this kernel, k_cgrid_stream2c (which
// still runs single-level fields, f64 plans and remainders below 4 levels) and the slab drivers give the same bits.
#include "gcmf_cgrid_ring_common.hpp"

namespace gcmf {

template <typename T> struct CRingP {
  const T *u0, *v0;    // b_{k+1} (first launch: the input f, scaled by p_n as it is loaded)
  const T *up, *vp;    // d_{k+1} = b_{k+1} + b_{k+2} (first launch: unused, d_n = b_n)
  const T *fu, *fv;    // the constant input f
  T *u1o, *v1o;        // level S: d of the next launch (unused by the last launch)
  T *u2o, *v2o;        // level S:     b_{k+1} of the next launch; last launch: the result when it has the state's type
  double *du, *dv;     // last launch: the f64 result of f32 state (or null)
  const T *coef[MAX_COEF];
  unsigned *redo;      // counts the workgroups that redid their strip with the full nan_to_num (instrumentation)
  int nx, rows, out_lo, out_hi;
  int H, nwx, ngroups, nlev, nlevp, wrap, last;
  long long bstride;
  double pn, pk[8], c;
};

// The operand rows come in through LDS-direct loads (global_load_lds_dwordx4, gfx950): 16 bytes per lane -- the access width the memory
// pipeline likes (8-byte accesses of this pattern stream at ~4.4 TB/s, experiments/cgrid_probe) -- and no registers in flight.  (Round 5
// also built and measured a plain-load form and eight waves per workgroup; both lost and are gone: launch_cgrid_ring has the numbers.)

template <int S, int D> struct CRingGeom {
  static constexpr int M = S <= 4 ? 4 : 8;   // level j is stale j cells per side; windows start on a multiple of 4 cells (16-byte loads)
  static constexpr int W = 128, WI = W - 2 * M;
  // LDS slots of the coefficient ring: the rows in flight occupy slots too: S + D (the slot of a row is a scalar that travels with the row)
  static constexpr int NS = S + D;
  static constexpr unsigned SLOTB = 14u * 512u;                        // bytes per slot (f32: 128 cells x 4 bytes per plane)
  static constexpr unsigned STG_OFF = NS * SLOTB;                      // the waves' staging slots: D per wave x (u0|v0, up|vp, fu|fv)
  // (16 plane loads are dealt to the four waves, 14 planes exist: the fourth wave's last pair fetches planes 12 | 13 a second time, into
  // their own place -- the same bytes to the same address -- instead of into a dummy kilobyte the six-level ring has no room for)
  static constexpr unsigned STGB = 3072u;
  static constexpr size_t lds_bytes() { return (size_t)STG_OFF + (unsigned)CR_WPB * D * STGB; }
  static constexpr int NDC = 8 / CR_WPB;   // coefficient loads per wave and row (two planes each; 16 plane slots over the waves)
};

// A helper wave (a level that pads the last workgroup of a tile): fetches and publishes its share of the coefficient rows, keeps the
// barriers, computes nothing.
template <typename T, int S, int D>
__device__ __forceinline__ void cgring_helper(const CRingP<T> &P, unsigned char *s_raw, const int lane, const int wv, const int pos0,
                                              const int r_begin, const int r_end, const int n_pad) {
  typedef CRingGeom<S, D> G;
  constexpr int NS = G::NS, NDC = G::NDC, PPW = 2 * NDC;   // PPW: planes per wave
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
    const int pr = PPW * wv + 2 * h >= 14 ? 12 : PPW * wv + 2 * h;   // (the padding pair: planes 12 | 13 again)
    const int pa = pr + half;
    q_c[h] = reinterpret_cast<const char *>(P.coef[pa]) + (unsigned)c4 * 4u;
    isa[h] = pa < 7;
    pair_off[h] = (unsigned)pr * 512u;
  }
  unsigned nxt = 0;
  auto issue = [&]() {
    cur.advance();
#pragma unroll
    for (int h = 0; h < NDC; ++h)
      cr_dma16(q_c[h] + (isa[h] ? cur.ro : cur.rc), lds0 + nxt + pair_off[h]);
    nxt = (nxt + G::SLOTB == NS * G::SLOTB) ? 0u : nxt + G::SLOTB;
  };
#pragma unroll
  for (int q = 0; q < D; ++q) issue();
  for (int r = r_begin; r < r_begin + n_pad; ++r) {
    cr_wait_vm<NDC * (D - 1)>();
    __syncthreads();
    issue();
  }
  cr_wait_vm<0>();
}

// One march of a strip by one wave (one level of the batch).  Returns whether a +-inf was delivered (wave-uniform); SAN = the redo pass.
template <typename T, int S, int D, bool FIRST, bool SAN, int NCARRY>
__device__ __forceinline__ bool cgring_march(const CRingP<T> &P, unsigned char *s_raw, const int lane, const int wv, const long long boff,
                                             const unsigned colB, const int pos0, const bool keep, const int a, const int b, const int n_pad) {
  typedef typename CgV2<T>::type v2;
  typedef CRingGeom<S, D> G;
  constexpr int NS = G::NS;
  constexpr int U = CR_U, RU = 6, RF = 12;
  constexpr int NDC = G::NDC, PPW = 2 * NDC;
  // NCARRY: the top NCARRY levels keep their previous row's three scaled copies in registers, the others rebuild them (three more
  // coefficient reads from LDS and a few packed multiplies per level)
  static_assert(NCARRY >= 1 && NCARRY <= S, "levels that carry");
  static_assert(D >= 1 && D <= 3 && S >= 2 && S + D <= RF && S < NS && U % D == 0, "ring periods");
  static_assert(sizeof(T) == 4, "LDS-direct loads of 16 bytes = four cells: f32 state");
  const int nx = P.nx, rows = P.rows;
  const bool wrap = P.wrap, last = P.last;
  const T c = (T)P.c;
  const int r_begin = a - S, r_end = b + S;  // rows delivered: [a - S, b + S - 1]
  const T *pu0 = (FIRST ? P.fu : P.u0) + boff, *pv0 = (FIRST ? P.fv : P.v0) + boff;
  const T *pup = (FIRST ? P.fu : P.up) + boff, *pvp = (FIRST ? P.fv : P.vp) + boff;  // (first launch: never loaded)
  const T *pfu = P.fu + boff, *pfv = P.fv + boff;
  CRingCursor cur(nx, rows, wrap, r_begin, r_end, (unsigned)sizeof(T));

  const v2 Z = {T(0), T(0)};
  v2 G0u[RU], G0v[RU];         // delivered rows of b_{k+1} ("level 0"); slot = (row - r_begin) mod RU
  v2 Vu, Vv;                   // the delivered row of d_{k+1} (read out of the staging slot every iteration)
  v2 Fu[RF], Fv[RF];           // rows of f; slot = (iteration that delivered them) mod RF
  v2 Xu[S][3], Xv[S][3];       // X[m], m = 1 .. S - 1: rows of b of level m; slot = (iteration that produced them) mod 3 (two are live)
  v2 Du[S][2], Dv[S][2];       // D[m], m = 1 .. S - 1: the row of d level m produced; slot = (iteration) mod 2
  v2 Lvt[S + 1][2], Lvh[S + 1][2], Luh[S + 1][2], LP[S + 1][2], LQ[S + 1][2], LR[S + 1][2];  // per level: what the previous row hands on
  bool Ku0[U], Ku1[U], Kv0[U], Kv1[U];   // "not NaN" of the delivered rows; slot = (row - r_begin) mod U
#pragma unroll
  for (int l = 0; l < RU; ++l) G0u[l] = G0v[l] = Z;
  Vu = Vv = Z;
#pragma unroll
  for (int l = 0; l < RF; ++l) Fu[l] = Fv[l] = Z;
#pragma unroll
  for (int m = 0; m < S; ++m) {
#pragma unroll
    for (int l = 0; l < 3; ++l) Xu[m][l] = Xv[m][l] = Z;
    Du[m][0] = Du[m][1] = Dv[m][0] = Dv[m][1] = Z;
  }
#pragma unroll
  for (int m = 0; m <= S; ++m) {
#pragma unroll
    for (int l = 0; l < 2; ++l) Lvt[m][l] = Lvh[m][l] = Luh[m][l] = LP[m][l] = LQ[m][l] = LR[m][l] = Z;
  }
#pragma unroll
  for (int l = 0; l < U; ++l) Ku0[l] = Ku1[l] = Kv0[l] = Kv1[l] = true;
  bool seen_inf = false;

  // ---- per-lane global pointers (lanes 0..31 fetch the first plane of a pair, four cells each, lanes 32..63 the second) ----
  unsigned slot_of[U];   // byte offset of the coefficient slot a row's planes went to; slot = (row - r_begin) mod U
#pragma unroll
  for (int l = 0; l < U; ++l) slot_of[l] = 0u;
  unsigned nxt_slot = 0u;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)s_raw);
  const int half = lane >> 5;
  int c4 = (pos0 + 4 * (lane & 31)) % nx;
  if (c4 < 0) c4 += nx;
  const unsigned c4B = (unsigned)c4 * 4u;
  const char *q_g0 = reinterpret_cast<const char *>(half ? pv0 : pu0) + c4B;
  const char *q_vp = reinterpret_cast<const char *>(half ? pvp : pup) + c4B;
  const char *q_ff = reinterpret_cast<const char *>(half ? pfv : pfu) + c4B;
  const char *q_c[NDC];
  bool isa[NDC];
  unsigned pair_off[NDC];
#pragma unroll
  for (int h = 0; h < NDC; ++h) {
    const int pr = PPW * wv + 2 * h >= 14 ? 12 : PPW * wv + 2 * h;   // (the padding pair: planes 12 | 13 again, see CRingGeom)
    const int pa = pr + half;   // this wave's share of the 14 coefficient planes (planes 0..6 travel with the delivered row, 7..13 with the row before)
    q_c[h] = reinterpret_cast<const char *>(P.coef[pa]) + c4B;
    isa[h] = pa < 7;
    pair_off[h] = (unsigned)pr * 512u;
  }
  const unsigned stg0 = G::STG_OFF + (unsigned)(wv * D) * G::STGB;   // this wave's staging slots
  auto issue_dma = [&](auto ph_c) {  // the LDS-direct loads of the next iteration: its coefficient slot, this wave's staging slot
    constexpr int ph = decltype(ph_c)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staging slot about to be refilled has been read out)
    cur.advance();
    slot_of[ph] = nxt_slot;
#pragma unroll
    for (int h = 0; h < NDC; ++h)
      cr_dma16(q_c[h] + (isa[h] ? cur.ro : cur.rc), lds0 + nxt_slot + pair_off[h]);
    nxt_slot = (nxt_slot + G::SLOTB == NS * G::SLOTB) ? 0u : nxt_slot + G::SLOTB;
    const unsigned st = lds0 + stg0 + (unsigned)(ph % D) * G::STGB;
    cr_dma16(q_g0 + cur.ro, st);
    if constexpr (!FIRST) cr_dma16(q_vp + cur.rc, st + 1024u);
    cr_dma16(q_ff + cur.rc, st + 2048u);
  };

  v2 out_u = Z, out_v = Z, out_pu = Z, out_pv = Z;

  // level j of iteration r (phase ph): fed with row rho = r - j + 1 of level j - 1, produces row rho - 1 of level j
  auto level = [&](auto jj, auto ph_c) {
    constexpr int j = decltype(jj)::value;
    constexpr int ph = decltype(ph_c)::value;
    constexpr int kn = cmod(ph - (j - 1), U);       // NaN masks of row rho
    constexpr int n3 = ph % 3, o3 = cmod(ph - 1, 3);
    const v2 inu = (j == 1) ? G0u[ph % RU] : Xu[j >= 2 ? j - 1 : 1][n3];
    const v2 inv = (j == 1) ? G0v[ph % RU] : Xv[j >= 2 ? j - 1 : 1][n3];
    const v2 xu = (j == 1) ? G0u[cmod(ph - 1, RU)] : Xu[j >= 2 ? j - 1 : 1][o3];   // row rho - 1 of level j - 1: the "-x" term
    const v2 xv = (j == 1) ? G0v[cmod(ph - 1, RU)] : Xv[j >= 2 ? j - 1 : 1][o3];
    // d_{k+1}, row rho - 1: what level j - 1 produced one iteration ago (level 1: the delivered row of d; first launch: d_n = b_n)
    const v2 dpu = (j == 1) ? (FIRST ? xu : Vu) : Du[j >= 2 ? j - 1 : 1][cmod(ph - 1, 2)];
    const v2 dpv = (j == 1) ? (FIRST ? xv : Vv) : Dv[j >= 2 ? j - 1 : 1][cmod(ph - 1, 2)];
    const v2 fu = Fu[cmod(ph - j + 1, RF)], fv = Fv[cmod(ph - j + 1, RF)];        // row rho - 1 of f
    v2 su, sv;
    if constexpr (SAN) {
      su.x = cr_san(inu.x);  su.y = cr_san(inu.y);
      sv.x = cr_san(inv.x);  sv.y = cr_san(inv.y);
    } else {
      su.x = Ku0[kn] ? inu.x : T(0);  su.y = Ku1[kn] ? inu.y : T(0);
      sv.x = Kv0[kn] ? inv.x : T(0);  sv.y = Kv1[kn] ? inv.y : T(0);
    }
    v2 A[7], B[7];   // the coefficient rows of iteration r - j + 1
    const v2 *cs = reinterpret_cast<const v2 *>(s_raw + slot_of[cmod(ph - (j - 1), U)]);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      A[q] = cs[q * 64 + lane];
      B[q] = cs[(7 + q) * 64 + lane];
    }
    constexpr int lo = cmod(ph - 1, 2), ln = ph % 2;
    // ---- what the previous row of this level hands on: the last level carries it, the others rebuild it (same operands, same bits) ----
    v2 vt_p, vh_p, uh_p;
    if constexpr (j <= S - NCARRY) {
      v2 pu, pv;   // the previous row as the stencil saw it
      if constexpr (SAN) {
        pu.x = cr_san(xu.x);  pu.y = cr_san(xu.y);
        pv.x = cr_san(xv.x);  pv.y = cr_san(xv.y);
      } else {
        constexpr int kp = cmod(ph - j, U);
        pu.x = Ku0[kp] ? xu.x : T(0);  pu.y = Ku1[kp] ? xu.y : T(0);
        pv.x = Kv0[kp] ? xv.x : T(0);  pv.y = Kv1[kp] ? xv.y : T(0);
      }
      // coefficient rows of the row before: the slot of iteration r - j
      const v2 *cp_ = reinterpret_cast<const v2 *>(s_raw + slot_of[cmod(ph - j, U)]);
      const v2 A1p = cp_[1 * 64 + lane], A2p = cp_[2 * 64 + lane], A3p = cp_[3 * 64 + lane];
      uh_p = pu * A1p;  vt_p = pv * A2p;  vh_p = pv * A3p;
    } else {
      vt_p = Lvt[j][lo];  vh_p = Lvh[j][lo];  uh_p = Luh[j][lo];
    }
    // ---- CgLevel::feed<true> on the pair (gcmf_cgrid_stream2.hip) ----
    const v2 ut = su * A[0], uh = su * A[1], vt = sv * A[2], vh = sv * A[3];
    // (the four x-differences as scalar operations: the neighbour lane's value rides on the subtraction as a DPP operand)
    v2 dut;  dut.x = ut.x - from_lower_lane0(ut.y);  dut.y = ut.y - ut.x;                 // ut - W ut
    const v2 Pr = __builtin_elementwise_fma(A[4], dut, -(A[5] * (vt - vt_p)));
    const v2 Qr = A[6] * Pr;
    const v2 vhp = vh_p;
    v2 dvh;  dvh.x = vhp.y - vhp.x;  dvh.y = from_upper_lane0(vhp.x) - vhp.y;             // E vh_p - vh_p
    const v2 Rm = __builtin_elementwise_fma(B[0], dvh, B[1] * (uh - uh_p));
    const v2 Sm = B[2] * Rm;
    const v2 Pp = LP[j][lo];
    v2 dpp;  dpp.x = Pp.x - Pp.y;  dpp.y = Pp.y - from_upper_lane0(Pp.x);                 // P_p - E P_p
    v2 dsm;  dsm.x = from_lower_lane0(Sm.y) - Sm.x;  dsm.y = Sm.x - Sm.y;                 // W Sm - Sm
    const v2 lu = __builtin_elementwise_fma(B[3], dpp, B[4] * (LR[j][lo] - Rm));
    const v2 lv = __builtin_elementwise_fma(B[5], dsm, -(B[6] * (LQ[j][lo] - Qr)));
    if constexpr (j > S - NCARRY) { Lvt[j][ln] = vt;  Lvh[j][ln] = vh;  Luh[j][ln] = uh; }
    LP[j][ln] = Pr;  LQ[j][ln] = Qr;  LR[j][ln] = Rm;
    // ---- Reinsch's form: d_k = p_k f - 2 c L(b_{k+1}) - d_{k+1},  b_k = d_k - b_{k+1};  the last level of the last launch is the result:
    //      p_0 f - c L(b_1) - d_1 ----
    const bool fin = last && j == S;
    const T mtc_s = fin ? -c : T(-2) * c;
    const v2 mtc = {mtc_s, mtc_s};
    const T pk_s = (T)P.pk[j - 1];
    const v2 pk = {pk_s, pk_s};
    const v2 dku = __builtin_elementwise_fma(pk, fu, __builtin_elementwise_fma(mtc, lu, -dpu));
    const v2 dkv = __builtin_elementwise_fma(pk, fv, __builtin_elementwise_fma(mtc, lv, -dpv));
    v2 cu = dku - xu, cv = dkv - xv;
    if constexpr (j == S) {
      cu.x = fin ? dku.x : cu.x;  cu.y = fin ? dku.y : cu.y;
      cv.x = fin ? dkv.x : cv.x;  cv.y = fin ? dkv.y : cv.y;
    }
    if constexpr (j < S) {
      Xu[j][n3] = cu;
      Xv[j][n3] = cv;
      Du[j][ph % 2] = dku;
      Dv[j][ph % 2] = dkv;
    }
    if constexpr (j == S) { out_u = cu;  out_v = cv;  out_pu = dku;  out_pv = dkv; }
  };

  auto phase = [&](auto ph_c, int r) {
    constexpr int ph = decltype(ph_c)::value;
    {
      // this iteration's rows were asked for D iterations ago; the loads of the D - 1 iterations in between may still be in flight
      cr_wait_vm<((FIRST ? 2 : 3) + NDC) * (D - 1)>();
      __syncthreads();   // ... and the other waves' quarters of the coefficient slot are there too
      const unsigned char *st = s_raw + stg0 + (unsigned)(ph % D) * G::STGB;
      G0u[ph % RU] = reinterpret_cast<const v2 *>(st)[lane];
      G0v[ph % RU] = reinterpret_cast<const v2 *>(st + 512)[lane];
      if constexpr (!FIRST) {
        Vu = reinterpret_cast<const v2 *>(st + 1024)[lane];
        Vv = reinterpret_cast<const v2 *>(st + 1536)[lane];
      }
      Fu[ph % RF] = reinterpret_cast<const v2 *>(st + 2048)[lane];
      Fv[ph % RF] = reinterpret_cast<const v2 *>(st + 2560)[lane];
      issue_dma(cic<(ph + D) % U>{});
    }
    // ---- the delivered row of b_{k+1}: b_n = p_n f in a first launch; its NaN cells; +-inf watched ----
    {
      v2 gu = G0u[ph % RU], gv = G0v[ph % RU];
      if constexpr (FIRST) {
        const T pn = (T)P.pn;
        const v2 pn2 = {pn, pn};
        gu = pn2 * gu;
        gv = pn2 * gv;
        G0u[ph % RU] = gu;
        G0v[ph % RU] = gv;
      }
      if constexpr (!SAN) {
        Ku0[ph % U] = (gu.x == gu.x);  Ku1[ph % U] = (gu.y == gu.y);
        Kv0[ph % U] = (gv.x == gv.x);  Kv1[ph % U] = (gv.y == gv.y);
        seen_inf = seen_inf || (mabs(gu.x) > MLim<T>::big()) || (mabs(gu.y) > MLim<T>::big()) || (mabs(gv.x) > MLim<T>::big()) ||
                   (mabs(gv.y) > MLim<T>::big());
      }
    }
    level(cic<1>{}, ph_c);
    if constexpr (S >= 2) level(cic<2>{}, ph_c);
    if constexpr (S >= 3) level(cic<3>{}, ph_c);
    if constexpr (S >= 4) level(cic<4>{}, ph_c);
    if constexpr (S >= 5) level(cic<5>{}, ph_c);
    if constexpr (S >= 6) level(cic<6>{}, ph_c);
    if constexpr (S >= 7) level(cic<7>{}, ph_c);
    if constexpr (S >= 8) level(cic<8>{}, ph_c);
    // ---- stores: row r - S of level S: b (or the result) ... ----
    const int ju = r - S;
    if (ju >= a && ju < b) {   // wave-uniform
      if constexpr (!SAN) {   // (the watch: inside this block, next to the stores that need the row in registers anyway)
        // Non-finite values that first appear INSIDE a launch (an f32 overflow in a stress, inf - inf; p_k f with f = +-inf where p_n f
        // was NaN): whatever a non-finite stencil operand touches is non-finite, so they show in the last level's row of d as +-inf or as
        // a NaN in a cell whose delivered row was not NaN -- the strip then takes the redo pass, where every level clamps like
        // k_cgrid_stream2c's nan_to_num (advisor, round 5)
        // (a NaN that was delivered sits in b's row too -- the level's own "-x" operand, still in its ring slot; d - d is NaN for +-inf as well)
        const v2 xo_u = (S >= 2) ? Xu[S >= 2 ? S - 1 : 1][cmod(ph - 1, 3)] : G0u[cmod(ph - 1, RU)];
        const v2 xo_v = (S >= 2) ? Xv[S >= 2 ? S - 1 : 1][cmod(ph - 1, 3)] : G0v[cmod(ph - 1, RU)];
        const v2 tu = out_pu - out_pu, tv = out_pv - out_pv;
        seen_inf = seen_inf | ((tu.x != tu.x) & (xo_u.x == xo_u.x)) | ((tu.y != tu.y) & (xo_u.y == xo_u.y)) |
                   ((tv.x != tv.x) & (xo_v.x == xo_v.x)) | ((tv.y != tv.y) & (xo_v.y == xo_v.y));
      }
      const unsigned vo = colB + (unsigned)(ju * nx) * (unsigned)sizeof(T);
      if (keep) {
        if (P.du) {   // (wave-uniform; last launch) f32 state, f64 result: NumPy >= 2 promotes p[k] * T
          typedef double d2 __attribute__((ext_vector_type(2)));
          d2 du, dv;
          du.x = (double)out_u.x;  du.y = (double)out_u.y;
          dv.x = (double)out_v.x;  dv.y = (double)out_v.y;
          *reinterpret_cast<d2 *>(reinterpret_cast<char *>(P.du + boff) + 2 * (size_t)vo) = du;
          *reinterpret_cast<d2 *>(reinterpret_cast<char *>(P.dv + boff) + 2 * (size_t)vo) = dv;
        } else {      // the next launch's b_{k+1}; last launch: the result in the state's type
          *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.u2o + boff) + vo) = out_u;
          *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.v2o + boff) + vo) = out_v;
        }
      }
    }
    if (!last && ju >= a && ju < b) {   // ... and its row of d
      const unsigned vo = colB + (unsigned)(ju * nx) * (unsigned)sizeof(T);
      if (keep) {
        *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.u1o + boff) + vo) = out_pu;
        *reinterpret_cast<v2 *>(reinterpret_cast<char *>(P.v1o + boff) + vo) = out_pv;
      }
    }
  };

  issue_dma(cic<0>{});
  if constexpr (D >= 2) issue_dma(cic<1>{});
  if constexpr (D >= 3) issue_dma(cic<2>{});
  for (int r = r_begin; r < r_begin + n_pad; r += U) {
    phase(cic<0>{}, r);  phase(cic<1>{}, r + 1);  phase(cic<2>{}, r + 2);  phase(cic<3>{}, r + 3);
    phase(cic<4>{}, r + 4);  phase(cic<5>{}, r + 5);  phase(cic<6>{}, r + 6);  phase(cic<7>{}, r + 7);
    phase(cic<8>{}, r + 8);  phase(cic<9>{}, r + 9);  phase(cic<10>{}, r + 10);  phase(cic<11>{}, r + 11);
  }
  cr_wait_vm<0>();   // (nothing of this march may land in LDS after a redo pass -- or the next workgroup -- has taken it over)
  return __any(seen_inf);
}

template <typename T, int S, int D, bool FIRST, int WPS, int NCARRY>
__global__ __launch_bounds__(64 * CR_WPB, WPS) void k_cgrid_ring(const CRingP<T> P) {
  constexpr int WPB = CR_WPB;
  typedef CRingGeom<S, D> G;
  constexpr int M = G::M, W = G::W, WI = G::WI;
  extern __shared__ __align__(16) unsigned char s_raw[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups are dealt to the 8 XCDs round-robin: the 13 workgroups (50 levels) of a group follow each other on ONE XCD, so that the
  // coefficient rows the first of them fetched are found in that XCD's L2 by the others (as k_cgrid_stream2)
  const int blk = blockIdx.x;
  const int xcd = blk & 7, slot = (blk >> 3) * WPB + wv;
  // (round 5 measured the other order too -- every XCD a CONTIGUOUS range of groups, so that neighbouring windows share an L2: 4.97 ms per
  // five-level launch against 4.49 ms; with the groups dealt round-robin all eight XCDs work in the same rows of every plane at a time)
  // (dealing the groups to the XCDs 2 / 4 / 8 at a time measured 457 / 450 / 444 G against 481 G, config 5)
  const int group = (slot / P.nlevp) * 8 + xcd;
  int lev = slot % P.nlevp;
  if (group >= P.ngroups) return;  // whole workgroups leave together
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
  const unsigned colB = (unsigned)col * (unsigned)sizeof(T);   // byte offset of the lane's two cells in a row (the stores)
  const bool keep = (lane * 2 >= M) && (lane * 2 < W - M) && (pos < nx);
  const int n_pad = ((b - a) + 2 * S + CR_U - 1) / CR_U * CR_U;   // the march is padded to whole ring periods
  bool bad = false;
  if (shadow) cgring_helper<T, S, D>(P, s_raw, lane, wv, pos0, a - S, b + S, n_pad);
  else bad = cgring_march<T, S, D, FIRST, false, NCARRY>(P, s_raw, lane, wv, boff, colB, pos0, keep, a, b, n_pad);
  // "did any wave meet a +-inf" without __syncthreads_or: that reserves 256 bytes of static LDS, and six levels need all 80 KB a
  // workgroup can have at two per CU.  The marches are over (every LDS-direct load has landed): the first staging word is free.
  unsigned *s_flag = reinterpret_cast<unsigned *>(s_raw + G::STG_OFF);
  __syncthreads();
  if (threadIdx.x == 0) *s_flag = 0u;
  __syncthreads();
  if (bad && lane == 0) *s_flag = 1u;
  __syncthreads();
  const bool redo = *s_flag != 0u;
  __syncthreads();   // (before the redo pass's loads overwrite the word)
  if (redo) {   // a +-inf somewhere in the workgroup's rows: the strip again, nan_to_num in full at every level
    if (threadIdx.x == 0 && P.redo) atomicAdd(P.redo, 1u);
    if (shadow) cgring_helper<T, S, D>(P, s_raw, lane, wv, pos0, a - S, b + S, n_pad);
    else cgring_march<T, S, D, FIRST, true, NCARRY>(P, s_raw, lane, wv, boff, colB, pos0, keep, a, b, n_pad);
  }
}

static bool cr_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// the state, input and result planes come straight from the caller (GCMF_DEVICE_PTRS, the slab drivers' pools): the 16-byte LDS-direct
// loads and the 16-byte f64 stores need them aligned -- unaligned views take k_cgrid_stream2c (advisor, round 5)
bool cgrid_ring_args_aligned(const VecMultiArgs &a) {
  for (int k = 0; k < 2; ++k) {
    if (!cr_al16(a.u0[k]) || !cr_al16(a.uprev[k]) || !cr_al16(a.fb_in[k]) || !cr_al16(a.u1o[k]) || !cr_al16(a.u2o[k]) || !cr_al16(a.fb_out[k]))
      return false;
  }
  return true;
}

bool cgrid_ring_supported(const gcmf_plan *pl, int64_t nbatch, int S) {
  if (pl->cgrid_ring <= 0 || pl->kind != K_CGRID || pl->cgrid_tile || pl->d.dtype != GCMF_F32) return false;
  if (nbatch < 2) return false;   // single-level fields: k_cgrid_stream2c's private-ring form
  if (S < 4 || S > 6 || S > pl->cgrid_ring_smax) return false;
  if (pl->g.nx % 4 || pl->g.nx < 4 || pl->g.rows < S + 2) return false;
  if ((long long)pl->g.rows * pl->g.nx * 4 >= (1LL << 32)) return false;   // 32-bit byte offsets inside a level's plane
  for (int k = 0; k < MAX_COEF; ++k)
    if (!cr_al16(pl->g.coef[k])) return false;
  return true;
}

template <typename T, int S, int D, int WPS, int NCARRY> static int launch_cr(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  constexpr int WI = CRingGeom<S, D>::WI, WPB = CR_WPB;
  const Geom &g = pl->g;
  CRingP<T> P;
  P.u0 = (const T *)a.u0[0];  P.v0 = (const T *)a.u0[1];
  P.up = (const T *)a.uprev[0];  P.vp = (const T *)a.uprev[1];
  P.fu = (const T *)a.fb_in[0];  P.fv = (const T *)a.fb_in[1];
  P.u1o = (T *)a.u1o[0];  P.v1o = (T *)a.u1o[1];
  P.u2o = (T *)a.u2o[0];  P.v2o = (T *)a.u2o[1];
  P.du = P.dv = nullptr;
  if (a.last) {
    if (sizeof(T) == 4 && !a.fb_is_f32) {  // f32 state, f64 result (NumPy >= 2 promotion of the reference)
      P.du = (double *)a.fb_out[0];
      P.dv = (double *)a.fb_out[1];
    } else {
      P.u2o = (T *)a.fb_out[0];
      P.v2o = (T *)a.fb_out[1];
    }
  }
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
  if (H <= 0) {
    // strips as tall as possible (a strip marches H + 2 S rows, padded to whole periods of 12) while the launch still fills whole
    // rounds of the resident waves: the fewest strips of <= 96 rows fix the number of rounds, then the strip count grows to fill the
    // last round; then H + 2 S is brought up to a whole number of periods
    const long long cap = 1024LL * WPS;
    const long long per_strip = (long long)P.nwx * P.nlevp, hmax = pl->cgrid_ring_hmax > 0 ? pl->cgrid_ring_hmax : 96;
    const long long ns_min = (nrows + hmax - 1) / hmax;
    const long long rounds = (ns_min * per_strip + cap - 1) / cap;
    long long ns = rounds * cap / per_strip;
    if (ns < ns_min) ns = ns_min;
    H = (int)((nrows + ns - 1) / ns);
    if (H < 16) H = 16;
    const long long nst = (nrows + H - 1) / H;
    H = (int)((nrows + nst - 1) / nst);   // (same strip count, evened out)
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.ngroups = P.nwx * ((nrows + H - 1) / H);
  P.wrap = g.south_wrap && g.north_wrap;
  P.last = a.last;
  P.bstride = (long long)g.rows * g.nx;
  P.pn = a.p0;
  for (int t = 0; t < 8; ++t) P.pk[t] = a.pk[t];
  P.c = a.c;
  const long long groups_per_xcd = (P.ngroups + 7) / 8;
  const long long blocks_per_xcd = (groups_per_xcd * P.nlevp + WPB - 1) / WPB;
  dim3 block(64 * WPB), grid((unsigned)(blocks_per_xcd * 8));
  const size_t lds = CRingGeom<S, D>::lds_bytes();
  // (the attribute belongs to the function ON A DEVICE: one bit per device ordinal and instantiation -- a process that drives plans on
  // several GPUs must set it on each; advisor, round 5)
  auto go = [&](auto kern, std::atomic<unsigned long long> &attr_set) -> int {
    const unsigned long long bit = 1ULL << (pl->d.device & 63);
    if (lds > 48 * 1024 && (pl->d.device >= 64 || !(attr_set.load(std::memory_order_relaxed) & bit))) {
      GCMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, P);
    return GCMF_OK;
  };
  static std::atomic<unsigned long long> set_first{0}, set_next{0};  // per instantiation
  int rc = a.first ? go(&k_cgrid_ring<T, S, D, true, WPS, NCARRY>, set_first) : go(&k_cgrid_ring<T, S, D, false, WPS, NCARRY>, set_next);
  if (rc) return rc;
  note_kernel(pl, std::string("gcmf::k_cgrid_ring<") + tyname<T>() + ", " + std::to_string(S) + ", " + std::to_string(D) + ", " +
                      (a.first ? "true" : "false") + ", " + std::to_string(WPS) + ", " + std::to_string(NCARRY) + ">", S,
              launch_geom(P.H, (nrows + H - 1) / H, P.nwx, 1, grid.x, grid.y, nrows));
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

int cgrid_ring_smax(const gcmf_plan *pl, int64_t nbatch) {
  for (int S = 6; S >= 4; --S)
    if (cgrid_ring_supported(pl, nbatch, S)) return S;
  return 0;
}

int launch_cgrid_ring(gcmf_plan *pl, const VecMultiArgs &a, hipStream_t s) {
  // What round 5 measured on config 5 (tools/measure_cgrid_ring.py, same box, alternating; G cells.steps/s): k_cgrid_stream2c 367; plain
  // loads S = 4: 388 (S = 5 spills: 280); LDS-direct loads S = 4: 407, S = 5: 490 (three rows in flight: one workgroup per CU, 412); one
  // wave per SIMD with S = 6 / 7 / 8: 378 / 264 / 256; eight levels per workgroup (coefficient rows fetched half as often, one barrier
  // for eight waves): 460; groups dealt to the XCDs 2 / 4 / 8 at a time or in contiguous ranges: 457 / 450 / 444 / 442; strips of
  // 64 / 80 / 120 / 144 rows instead of 96: 448 / 461 / 468 / 459.  Two attempts at the coefficient re-reads (the counters say the kernel
  // moves 1.5 x its algorithmic bytes and that only half of the re-reads of a group's 13 workgroups hit in the L2 -- they start up to a
  // fifth of a strip apart): the non-temporal hint on the state rows' loads, to leave the L2 to the coefficient rows: 388; a persistent
  // launch of teams (416 workgroups, the 13 of a group walking through their strips together, no waiting between them): 365 -- both
  // the same bits, both slower (profiles/r05/cfg5_*.txt).  Only the LDS-direct form with four waves per workgroup is left in this file
  // (profiles of round 5 taken before the clean-up name the kernel k_cgrid_ring<float, 5, 2, false, 2, true, 4>: the same code).
  // Round 6 (same box, alternating, config 5; G cells.steps/s): every level carrying its previous row's scaled copies in registers instead
  // of rebuilding them from LDS (the compiler, nudged by where the +-inf watch sits, now needs 225 instead of 255 registers for five
  // levels): 479 against 475 (three coefficient reads and ~8 VALU instructions fewer per level: neither the LDS nor the VALU paces this
  // kernel); SIX levels per launch at two workgroups per CU -- 80 KB of LDS each: the dummy kilobyte of the coefficient ring and the 256
  // static bytes of __syncthreads_or had to go -- 44 levels = 6 6 6 6 5 5 5 5: 508-513 (5.0 ms per six-level launch against 4.5 per
  // five); with one workgroup per CU (81 KB) the same kernel ran 6.5 ms.  Six levels with all of them carrying spill; two carry.
  const bool r5 = pl->cgrid_ring_ncarry == 1;   // (option "cgrid_ring_ncarry" = 1: round 5's form, only the last level carries)
  switch (a.S) {   // (LDS-direct loads: windows start on multiples of four cells -- cgrid_ring_supported asked for nx % 4 == 0)
    case 4: return r5 ? launch_cr<float, 4, 2, 2, 1>(pl, a, s) : launch_cr<float, 4, 2, 2, 4>(pl, a, s);
    case 5: return r5 ? launch_cr<float, 5, 2, 2, 1>(pl, a, s) : launch_cr<float, 5, 2, 2, 5>(pl, a, s);
    case 6: return r5 ? launch_cr<float, 6, 2, 2, 1>(pl, a, s) : launch_cr<float, 6, 2, 2, 2>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
