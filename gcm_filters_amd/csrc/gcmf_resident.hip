// k_resident: the filter polynomial on a field that STAYS ON THE CHIP -- "one persistent field per GPU with the whole n_steps
// polynomial fused into a single launch, 2-D blocking with LDS-staged halo tiles" (north star), for fields small enough to live in
// the register files and LDS of the 256 CUs: the 300-row slab one rank owns when a 2400 x 3600 grid is cut 8 ways (1.3 M cells with
// its ghost rows), BASELINE config 1 (512 x 512), and anything else up to ~1.5 M cells (f64).
//
// Why.  The strip-marching kernels (k_ring / k_ringc) re-read every plane from HBM / the memory-side cache once per 5-8 levels and
// need >= 1024 independent wave strips to fill the chip.  On a 300-row slab a strip is 11 rows tall and marches 11 + 2 S rows: 2.45 x
// redundant, 28 us per 8-level launch, 0.256 ms per application -- the 8-GPU strong-scaling bound was 4.0 x.  Here
// every workgroup (one per CU) owns ONE 2-D tile for the whole launch (up to 64 levels): coefficients, the constant input and both
// Clenshaw states of its cells sit in registers (a thread owns RC consecutive cells of a tile row), the newest state is mirrored in
// LDS so that neighbours can read it (structure-of-arrays: every LDS access is unit-stride across lanes, conflict-free), one barrier
// per level.  Tiles carry a halo of K = 4 cells that goes stale by one cell per level; every K levels the tiles trade their edge
// bands THROUGH the memory-side cache: band cells are staged as flat lists in LDS and stored by consecutive lanes to an exchange plane in
// UNCACHED device memory, every store acknowledged, barrier, a per-tile epoch flag is raised, the eight neighbours' flags are polled,
// halo cells come back the same way.  Neighbour-to-neighbour, never grid-wide; no L2-wide fence.  Nothing but those bands (22 % of a
// tile) and the final result touches memory between the first load and the last store.
//
// Arithmetic = k_ringc's backward (Clenshaw) level, operand for operand (gcmf_ringc_impl.hpp `level`): results are bit-identical to
// the strip-marching path however the levels are cut into launches (tests/test_gpu_resident.py).  NaN semantics: a tile runs without
// nan_to_num until a non-finite value shows up in one of its cells (checked on everything loaded and everything produced; the flag
// rides on the level barrier), from then on its stencil operands go through nan_to_num (kernels.py:175, 300) -- what k_ringc's redo
// pass computes.  REGULAR has no nan_to_num in the reference (NaN spreads, kernels.py:113-121) and none here.
//
// WHERE IT STANDS (round 4, MI355X, tools/measure_resident.py / experiments/scripts/probe_resident.py): correct and bit-identical, NOT
// faster -- therefore opt-in (GCMF_RESIDENT=1).  364 x 3600 IRREGULAR f64: load + store 33 us, ~1.0 us per level (the f64 issue
// rate of 13 cells x 8 waves per CU), ~9.5 us per tile exchange; 32 levels 132 us, 64 levels 239 us -- the strip-marching launches do
// 32 levels of the same slab in ~115 us.  The exchange is what costs: 15 of them per 63-level filter, because registers + LDS of a CU
// hold no more than a K = 4 halo around a 91 x 57 tile of a flux-form grid (7 doubles per cell).  The protocol alone (same volumes, no
// arithmetic: experiments/tile_exchange_probe) takes 4.4 us; LL-style tagged items without flags are no faster (4.9 us).  History of
// the exchange, each step measured: agent-scope fences by every wave 130 us (a whole-L2 write-back / invalidate each); per-access sc1
// atomics on ordinary memory: stale halos; a failure word in mapped host memory POLLED by 2048 lanes: 30 us of PCIe reads; plain loads
// of uncached memory: stale lines in the CU's vector L1 from two exchanges ago (`buffer_inv sc0` is a no-op outside threadgroup-split
// mode, `buffer_inv sc1` per wave costs 15 us); per-thread scattered band / halo accesses 14 us; flat lists by all lanes 9.5 us.
//
// Deadlock freedom: at most one workgroup per CU and no more workgroups than CUs, so every workgroup becomes resident once whatever ran
// before has drained; every flag wait is bounded (s_memrealtime) and a wait that runs out poisons the result with NaN and raises the
// plan's sticky failure word (mapped host memory; the next call on the plan returns GCMF_ERR_HIP) -- never a hung GPU.
// Two PROCESSES sharing one GPU must not run resident launches at the same time (each would hold CUs the other waits for): SlabFilter
// falls back to the strip-marching launches when its ranks share a device (this repo's one-GPU test set-up).
#include "gcmf_multi_common.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gcmf {

constexpr int K_MASKZ = 5;     // (as in gcmf_scalar_multi_impl.hpp: the land-mask stencil on states whose land cells are zero)
constexpr int RES_MAXL = 64;   // levels per launch
constexpr int RES_NT = 512;    // threads per workgroup: two waves per SIMD with up to 256 registers each

struct ResP {
  const double *u0, *v0;   // b_{k+1}, b_{k+2} (ignored by a first launch: b_n = p_n f, b_{n+1} = 0)
  double *uo, *vo;         // the states after L levels (not written by a last launch)
  const double *f;         // the constant input (prepare()d and land-masked as it is loaded)
  double *out;             // last launch: the result
  const double *cE, *cN, *ra;   // K_FLUX
  const uint8_t *mbits;         // K_MASKZ
  const uint8_t *lbits;         // land bits (bit 0: the cell exchanges with a neighbour) or NULL
  const double *area;           // area-weighted types or NULL
  double *ex[2][2];        // exchange planes [parity][state]
  unsigned *flags;         // one epoch word per tile
  unsigned *fail;          // failure word (device address of mapped host memory): written on a time-out, read by the host only
  unsigned *dfail;         // failure word in the uncached arena: the serial number of the launch that timed out (tiles read it)
  unsigned serial;         // this launch's serial number (never 0)
  unsigned epoch0;
  int nx, rows;            // the plan's slab allocation
  int r_lo, r_hi;          // rows this launch keeps alive (the dependency cone of out_lo .. out_hi)
  int out_lo, out_hi;      // rows stored at the end
  int nty, ntx, nruns, K, L;
  int wrap;                // y-periodic and the region is the whole domain
  int first, last;
  int xcd_per;             // tiles per XCD when the tile count is a multiple of 8 (neighbouring tiles share an L2), else 0
  double pk[RES_MAXL];
  double p0, c;
  long long spin_limit;
};

__device__ __forceinline__ bool res_finite(double x) { return __builtin_fabs(x) <= DBL_MAX; }   // false for NaN and +-inf

// The exchange planes and the tile flags are written by one XCD and read by another INSIDE one kernel; the XCDs' L2s are not coherent
// with each other for ordinary (coarse-grained) memory.  Agent-scope fences would make them so -- at a whole-L2 write-back / invalidate
// per wave: the first version of this kernel spent 130 us per exchange in them (8 waves x 32 workgroups per XCD, serialised at the L2).
// Per-access device-scope (sc1) atomics on ordinary memory turned out NOT to be enough (sporadic stale halos) and serialise (26 dependent
// round trips per edge thread).  So the exchange planes and flags live in UNCACHED device memory (hipDeviceMallocUncached: MTYPE UC,
// every access goes to the memory side, where the 256 MB memory-side cache is shared by all XCDs): plain loads and stores, issued in
// batches, ordered by "all my stores acknowledged (vmcnt 0) -> workgroup barrier -> flag store" on the writer's side and "flag seen
// -> workgroup barrier -> loads" on the reader's.
__device__ __forceinline__ void dev_store(double *p, double v) { *p = v; }
// loads go past the CU's vector L1 (device scope: sc1), which may still hold the line from two exchanges ago (same parity, same address):
// measured -- plain loads returned stale halos from the third exchange on; a `buffer_inv sc0` does nothing outside threadgroup-split
// mode and a `buffer_inv sc1` per wave costs 15 us per exchange
__device__ __forceinline__ double dev_load(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

// NT = threads per workgroup: 512 (two waves per SIMD, up to 256 registers: the deep instantiations) or 1024 (four waves per SIMD, 128
// registers: RC = 4 only -- small tiles, where a level is bound by latency and more waves hide more of it)
template <int KIND, int RC, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_resident(const ResP P) {
  constexpr int RES_NT = NT;
  constexpr bool FLUX = (KIND == K_FLUX), MASK = (KIND == K_MASKZ);
  constexpr bool WATCH = (KIND != K_REG);
  // the deepest flux instantiation keeps the constant input in LDS (a third plane) instead of 2 * RC registers: 256 registers spilled
  constexpr bool FLDS = FLUX && RC >= 13;
  constexpr int NB = RC * RES_NT;                // doubles per LDS buffer
  extern __shared__ double xs[];                 // [2][RC][RES_NT]: the newest state b_{k+1} of every cell of the padded tile (+ [RC][RES_NT]: f)
  __shared__ unsigned s_bad[2];                  // a non-finite value was seen in this tile (double-buffered with the level parity)
  __shared__ int s_fail;
  int bx = blockIdx.x;
  if (P.xcd_per > 0) bx = (bx & 7) * P.xcd_per + (bx >> 3);
  const int ty = bx / P.ntx, tx = bx - ty * P.ntx;
  const int tid = threadIdx.x;
  const int nx = P.nx, rows = P.rows, K = P.K, nruns = P.nruns;
  const int Rr = P.r_hi - P.r_lo;
  const int r0 = P.r_lo + (int)((long long)ty * Rr / P.nty), r1 = P.r_lo + (int)((long long)(ty + 1) * Rr / P.nty);
  const int c0 = (int)((long long)tx * nx / P.ntx), c1 = (int)((long long)(tx + 1) * nx / P.ntx);
  const int h = r1 - r0, w = c1 - c0;            // owned rows / columns (both >= 2 K: res_geometry)
  const int PH = h + 2 * K, PW = nruns * RC, PWl = w + 2 * K;   // padded rows, padded columns (allocated / live)
  const int prow = tid / nruns, run = tid - prow * nruns;
  const bool active = prow < PH;
  const int pc0 = run * RC;

  // ---- where cells are ---------------------------------------------------------------------------------------------------------
  // padded-tile row pr / column pc -> row / column of the plan's planes; -1: dead (outside the rows this launch keeps alive, or a
  // padding column): zero coefficients, zero state, never exchanged
  auto grow = [&](int pr) {
    int g = r0 - K + pr;
    if (P.wrap) return g < 0 ? g + rows : (g >= rows ? g - rows : g);
    return (g >= P.r_lo && g < P.r_hi) ? g : -1;
  };
  auto gcolp = [&](int pc) {
    if (pc >= PWl) return -1;
    int g = c0 - K + pc;
    g = g < 0 ? g + nx : g;
    return g >= nx ? g - nx : g;                  // (one period is enough: PWl <= nx + 2 K <= 2 nx, res_geometry)
  };
  const int gr = active ? grow(prow) : -1;
  const bool own_row = active && prow >= K && prow < PH - K && gr >= 0;
  unsigned owned = 0u, band = 0u, halo = 0u;   // bit c: cell c of this thread is ...
#pragma unroll
  for (int c = 0; c < RC; ++c) {
    const int pc = pc0 + c;
    const bool lv = gr >= 0 && pc < PWl;
    const bool ow = lv && own_row && pc >= K && pc < K + w;
    const bool bd = ow && (prow < 2 * K || prow >= PH - 2 * K || pc < 2 * K || pc >= w);
    owned |= ow ? (1u << c) : 0u;
    band |= bd ? (1u << c) : 0u;
    halo |= (lv && !ow) ? (1u << c) : 0u;
  }
  // The band (owned cells within K of the tile edge: what the neighbours need) and the halo (the padded frame around the owned cells:
  // what this tile needs) as FLAT lists, so that the exchange moves them with consecutive lanes on consecutive cells:
  //   band  i: [0, K w) top rows, [K w, 2 K w) bottom rows, then (h - 2K) x K left columns, then (h - 2K) x K right columns
  //   halo  i: [0, K PWl) top rows, [K PWl, 2 K PWl) bottom rows, then h x K left columns, then h x K right columns
  const int nband = 2 * K * w + 2 * (h - 2 * K) * K, nhalo = 2 * K * PWl + 2 * h * K;
  auto band_cell = [&](int i, int &pr, int &pc) {         // flat band index -> padded-tile coordinates
    if (i < 2 * K * w) {
      const int r = i / w;
      pc = K + (i - r * w);
      pr = r < K ? K + r : PH - 2 * K + (r - K);
    } else {
      int j = i - 2 * K * w;
      const bool right = j >= (h - 2 * K) * K;
      j -= right ? (h - 2 * K) * K : 0;
      const int r = j / K;
      pr = 2 * K + r;
      pc = (right ? w : K) + (j - r * K);
    }
  };
  auto halo_cell = [&](int i, int &pr, int &pc) {
    if (i < 2 * K * PWl) {
      const int r = i / PWl;
      pc = i - r * PWl;
      pr = r < K ? r : PH - 2 * K + r;
    } else {
      int j = i - 2 * K * PWl;
      const bool right = j >= h * K;
      j -= right ? h * K : 0;
      const int r = j / K;
      pr = K + r;
      pc = (right ? K + w : 0) + (j - r * K);
    }
  };
  // my own cells' flat indices: a thread's row is of one kind (top / bottom / middle), so cell c's index is base + c (clamped users)
  const int my_band0 = !own_row ? 0 : (prow < 2 * K ? (prow - K) * w : (prow >= PH - 2 * K ? K * w + (prow - (PH - 2 * K)) * w : -1));
  const int my_halo0 = !active ? 0 : (prow < K ? prow * PWl : (prow >= PH - K ? K * PWl + (prow - (PH - K)) * PWl : -1));
  auto band_index = [&](int c) {                          // (only for cells whose band bit is set)
    const int pc = pc0 + c;
    if (my_band0 >= 0) return my_band0 + (pc - K);
    const int r = prow - 2 * K;
    return 2 * K * w + (pc < 2 * K ? r * K + (pc - K) : (h - 2 * K) * K + r * K + (pc - w));
  };
  auto halo_index = [&](int c) {                          // (only for cells whose halo bit is set)
    const int pc = pc0 + c;
    if (my_halo0 >= 0) return my_halo0 + pc;
    const int r = prow - K;
    return 2 * K * PWl + (pc < K ? r * K + pc : h * K + r * K + (pc - K - w));
  };
  // LDS neighbours (threads): the padded edge reads itself (garbage that never reaches a valid cell: the halo is K deep)
  const int tN = (active && prow + 1 < PH && !(!P.wrap && gr == rows - 1)) ? tid + nruns : tid;
  const int tS = (active && prow >= 1 && !(!P.wrap && gr == 0)) ? tid - nruns : tid;
  const int tW = run > 0 ? tid - 1 : tid;
  const int tE = (run < nruns - 1) ? tid + 1 : tid;

  // ---- load: planes come in COALESCED (consecutive lanes = consecutive cells of a padded-tile row), two at a time, are staged
  // row-major in the two LDS buffers and picked up by their owners (a thread's RC cells are consecutive there) ---------------------------
  // flat element e = k * RES_NT + tid of the padded tile -> offset in the plan's planes, -1 = dead
  auto offs = [&](int k) {
    const int e = k * RES_NT + tid;
    int o = -1;
    if (e < PH * PW) {
      const int pr = e / PW, pc = e - pr * PW;
      const int g = grow(pr), gc_ = gcolp(pc);
      if (g >= 0 && gc_ >= 0) o = g * nx + gc_;
    }
    return o;
  };
  int goff[RC];        // (the load phase only: the store phase recomputes them instead of keeping RC registers through the levels)
#pragma unroll
  for (int k = 0; k < RC; ++k) goff[k] = offs(k);
  double *stgA = xs, *stgB = xs + NB;             // staging: PH * PW <= RC * RES_NT doubles each
  auto stage2 = [&](const double *pa, const double *pb) {   // (call between barriers) two planes in flight
    double va[RC], vb[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) {
      va[k] = goff[k] >= 0 ? pa[goff[k]] : 0.0;
      vb[k] = goff[k] >= 0 ? pb[goff[k]] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < RC; ++k) {
      const int e = k * RES_NT + tid;
      if (e < PH * PW) {
        stgA[e] = va[k];
        stgB[e] = vb[k];
      }
    }
  };
  const int mine = prow * PW + pc0;               // my first cell in a staged tile
  double b1[RC], b2[RC], ff[FLDS ? 1 : RC];
  double *fl = xs + 2 * NB;
  double cEr[FLUX ? RC + 1 : 1], cNr[FLUX ? RC : 1], cSr[FLUX ? RC : 1], rar[FLUX ? RC : 1];
  unsigned mb[MASK ? RC : 1];
  bool bad = false;
  if (tid == 0) { s_bad[0] = 0u; s_bad[1] = 0u; s_fail = 0; }
  {
    // round 1: the constant input -- prepare()d (x area, kernels.py:100-101) and land-masked as it is staged -- and the first
    // coefficient plane (flux kinds) / the mask bytes (land-mask kinds)
    double va[RC], vb[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) {
      double fv = 0.0, q = 0.0;
      if (goff[k] >= 0) {
        fv = P.f[goff[k]];
        if (P.area) fv = fv * P.area[goff[k]];
        if (P.lbits && !(P.lbits[goff[k]] & 1u)) fv = 0.0;
        if constexpr (FLUX) q = P.cE[goff[k]];
        if constexpr (MASK) q = (double)P.mbits[goff[k]];
      }
      va[k] = fv;
      vb[k] = q;
    }
#pragma unroll
    for (int k = 0; k < RC; ++k) {
      const int e = k * RES_NT + tid;
      if (e < PH * PW) {
        stgA[e] = va[k];
        stgB[e] = vb[k];
      }
    }
  }
  __syncthreads();
  double fown[RC];
#pragma unroll
  for (int c = 0; c < RC; ++c) {
    fown[c] = active ? stgA[mine + c] : 0.0;
    if constexpr (FLUX) cEr[c + 1] = active ? stgB[mine + c] : 0.0;
    if constexpr (MASK) mb[c] = active ? (unsigned)stgB[mine + c] : 0u;
    if (WATCH) bad = bad || !res_finite(fown[c]);
  }
  if constexpr (FLUX) cEr[0] = (active && pc0 > 0) ? stgB[mine - 1] : 0.0;   // the east face of the cell west of my run (run 0: a padded edge, garbage anyway)
  __syncthreads();
  if constexpr (FLUX) {
    stage2(P.cN, P.ra);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      cNr[c] = active ? stgA[mine + c] : 0.0;
      cSr[c] = (active && prow > 0) ? stgA[mine - PW + c] : 0.0;   // the north face of the row below (dead rows: zero, no flux)
      rar[c] = active ? stgB[mine + c] : 0.0;
    }
    __syncthreads();
  }
  if (P.first) {
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      b1[c] = P.p0 * fown[c];                      // b_n = p_n f (f is zero on land and on dead cells), b_{n+1} = 0
      b2[c] = 0.0;
    }
  } else {
    stage2(P.u0, P.v0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      b1[c] = active ? stgA[mine + c] : 0.0;
      b2[c] = active ? stgB[mine + c] : 0.0;
      if (WATCH) bad = bad || !(res_finite(b1[c]) && res_finite(b2[c]));
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < RC; ++c) {
    if constexpr (FLDS) fl[c * RES_NT + tid] = fown[c];
    else ff[c] = fown[c];
    xs[c * RES_NT + tid] = b1[c];
  }
  if (WATCH && bad) s_bad[0] = 1u;
  __syncthreads();

  bool sani = false;
  unsigned epoch = P.epoch0;
  const double cc = P.c;

  auto level = [&](auto sani_c, int l) {
    constexpr bool SANI = decltype(sani_c)::value;
    const double *cur = xs + (l & 1) * NB;
    double *nxt = xs + ((l + 1) & 1) * NB;
    const double pk = P.pk[l];
    const double two = (P.last && l == P.L - 1) ? 1.0 : 2.0;   // the last level of a filter is p_0 f + A(b_1) - b_2: A, not 2 A
    auto sn = [&](double x) { return SANI ? msan(x) : x; };
    const double xW = sn(cur[(RC - 1) * RES_NT + tW]);
    const double xE = sn(cur[tE]);
    double xc[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) xc[c] = sn(b1[c]);
    double fprev = 0.0;
    if constexpr (FLUX) fprev = (xc[0] - xW) * cEr[0];
    bool bd = false;
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      const double xn = sn(cur[c * RES_NT + tN]);
      const double xsv = sn(cur[c * RES_NT + tS]);
      const double xe = (c == RC - 1) ? xE : xc[c < RC - 1 ? c + 1 : c];
      double Lp;
      if constexpr (FLUX) {
        const double fe = (xe - xc[c]) * cEr[c + 1];
        const double fw = fprev;
        fprev = fe;
        const double fn = (xn - xc[c]) * cNr[c];
        const double fs = (xc[c] - xsv) * cSr[c];
        Lp = ((fe - fw) + (fn - fs)) * rar[c];
      } else {
        const double xw = (c == 0) ? xW : xc[c > 0 ? c - 1 : 0];
        if constexpr (MASK) {
          const unsigned bb = mb[c];
          const double wf = (double)(bb >> 5);
          Lp = rfma(-wf, xc[c], xe);
          Lp = Lp + xw;
          Lp = Lp + xn;
          Lp = Lp + xsv;
          Lp = (bb & 1u) ? Lp : 0.0;
        } else {
          Lp = rfma(-4.0, xc[c], xe);
          Lp = Lp + xw;
          Lp = Lp + xn;
          Lp = Lp + xsv;
        }
      }
      const double av = cheb_a<true>(b1[c], cc, Lp);        // "-x" takes the raw value: a NaN stays in its cell (filter.py:166-175)
      double tk = rfma(two, av, -b2[c]);
      tk = rfma(pk, FLDS ? fl[c * RES_NT + tid] : ff[FLDS ? 0 : c], tk);
      b2[c] = b1[c];
      b1[c] = tk;
      nxt[c * RES_NT + tid] = tk;
      if (WATCH && !SANI) bd = bd || !res_finite(tk);
    }
    if (WATCH && !SANI && bd) s_bad[(l + 1) & 1] = 1u;
  };

  for (int l = 0; l < P.L; ++l) {
    if (WATCH && !sani && s_bad[l & 1]) sani = true;       // (uniform: written before the barrier that ended the previous level)
    if (active) {
      if (sani) level(std::true_type{}, l);
      else level(std::false_type{}, l);
    }
    __syncthreads();
    if ((l + 1) % K == 0 && l + 1 < P.L) {
      // ---- trade the edge bands with the eight neighbour tiles through the memory-side cache.  The buffer of the state before this
      // level is free now: owners put their band cells there as flat lists (both states), all lanes copy the lists out with consecutive
      // lanes on consecutive cells; every store acknowledged (vmcnt 0), barrier, ONE flag store; eight lanes poll the neighbours'
      // flags, barrier; the halo comes in the same way (device-scope loads) and its owners pick it up.  No L2-wide fence anywhere. ------
      ++epoch;
      double *e1 = P.ex[epoch & 1][0], *e2 = P.ex[epoch & 1][1];
      double *sg = xs + (l & 1) * NB;                      // (nband + nhalo <= 2 K (PH + PW) << NB / 2)
      double *sg2 = sg + NB / 2;
      if (band) {
#pragma unroll
        for (int c = 0; c < RC; ++c)
          if ((band >> c) & 1u) {
            const int i = band_index(c);
            sg[i] = b1[c];
            sg2[i] = b2[c];
          }
      }
      __syncthreads();
      for (int i = tid; i < nband; i += RES_NT) {
        int pr, pc;
        band_cell(i, pr, pc);
        const long long o = (long long)grow(pr) * nx + gcolp(pc);   // (band cells are owned: alive)
        e1[o] = sg[i];
        e2[o] = sg2[i];
      }
      // a tile of this launch has already timed out (its bands are garbage from then on): no more waiting, every tile poisons its result
      if (tid == 9 && __hip_atomic_load(P.dfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.serial) s_fail = 1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);                             // every store of this wave has been acknowledged by the memory side
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&P.flags[bx], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid >= 1 && tid <= 8) {
        const int q = tid - 1 + (tid - 1 >= 4 ? 1 : 0);          // 0..8 without the centre
        int ny_ = ty + q / 3 - 1, nx_ = tx + q % 3 - 1;
        nx_ = nx_ < 0 ? nx_ + P.ntx : (nx_ >= P.ntx ? nx_ - P.ntx : nx_);
        bool there = true;
        if (ny_ < 0 || ny_ >= P.nty) {
          if (P.wrap) ny_ = ny_ < 0 ? ny_ + P.nty : ny_ - P.nty;
          else there = false;                                      // the region ends here: those halo rows are dead
        }
        if (there && !s_fail) {
          const unsigned *fl_ = &P.flags[ny_ * P.ntx + nx_];
          const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
          for (;;) {
            const unsigned v = __hip_atomic_load(fl_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - epoch) >= 0) break;
            // (the failure word lives in mapped HOST memory: it is written on a time-out, never polled -- a read of it is a PCIe
            // round trip, and 2048 lanes polling it cost 30 us per exchange in the first version)
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > P.spin_limit) {
              __hip_atomic_store(P.dfail, P.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (before any later band of this tile)
              __hip_atomic_store(P.fail, P.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (which launch: the host tells the plan that issued it)
              s_fail = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      __syncthreads();
      for (int i = tid; i < nhalo; i += RES_NT) {
        int pr, pc;
        halo_cell(i, pr, pc);
        const int g = grow(pr), gc_ = gcolp(pc);
        double v1 = 0.0, v2 = 0.0;
        if (g >= 0 && gc_ >= 0) {
          const long long o = (long long)g * nx + gc_;
          v1 = dev_load(&e1[o]);
          v2 = dev_load(&e2[o]);
        }
        sg[i] = v1;
        sg2[i] = v2;
      }
      __syncthreads();
      if (halo) {
        double *nxt = xs + ((l + 1) & 1) * NB;
        bool bd = false;
#pragma unroll
        for (int c = 0; c < RC; ++c)
          if ((halo >> c) & 1u) {
            const int i = halo_index(c);
            b1[c] = sg[i];
            b2[c] = sg2[i];
            nxt[c * RES_NT + tid] = b1[c];
            if (WATCH) bd = bd || !(res_finite(b1[c]) && res_finite(b2[c]));
          }
        if (WATCH && bd) s_bad[(l + 1) & 1] = 1u;
      }
      __syncthreads();
    }
  }

  // ---- store: owners put their cells into the staged tiles, consecutive lanes write consecutive cells -------------------------------
  // (a neighbour that timed out wrote the launch's failure word BEFORE it posted the garbage bands this tile may have consumed since)
  if (tid == 0 && __hip_atomic_load(P.dfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.serial) s_fail = 1;
  const double poison = __longlong_as_double(-1LL);
  if (active) {
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      stgA[mine + c] = b1[c];
      stgB[mine + c] = b2[c];
    }
  }
  __syncthreads();
  const bool failed = s_fail != 0;
#pragma unroll
  for (int k = 0; k < RC; ++k) {
    const int e = k * RES_NT + tid;
    const int o = offs(k);
    if (o >= 0) {
      const int pr = e / PW, pc = e - pr * PW;
      const int g = o / nx;
      if (pr >= K && pr < PH - K && pc >= K && pc < K + w && g >= P.out_lo && g < P.out_hi) {
        if (P.last) {
          double r = stgA[e];
          if (P.area) r = r / P.area[o];                     // finalize(): / area (kernels.py:103-104)
          P.out[o] = failed ? poison : r;
        } else {
          P.uo[o] = failed ? poison : stgA[e];
          P.vo[o] = failed ? poison : stgB[e];
        }
      }
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------

struct ResGeom {
  int rc = 0, nt = 512, nty = 0, ntx = 0, nruns = 0, K = 4;
  long long cost = 0;
};

// tiles for a region of Rr rows x nx columns on at most max_wg workgroups: a thread owns RC cells of a padded tile row, a workgroup
// has RES_NT threads and 2 * RC * RES_NT * 8 bytes of LDS.  The cheapest geometry = the fewest padded cells per workgroup.
static ResGeom res_geometry(int kind, int Rr, int nx, int K, int max_wg) {
  ResGeom best;
  // (RC, NT) pairs on offer: 1024 threads only with four cells per thread (128 registers per lane)
  const int rcs_flux[][2] = {{4, 1024}, {4, 512}, {8, 512}, {13, 512}}, rcs_other[][2] = {{4, 1024}, {4, 512}, {8, 512}, {13, 512}, {16, 512}};
  const int(*rcs)[2] = kind == K_FLUX ? rcs_flux : rcs_other;
  const int nrc = kind == K_FLUX ? 4 : 5;
  static const int nt_force = getenv("GCMF_RESIDENT_NT") ? atoi(getenv("GCMF_RESIDENT_NT")) : 0;
  for (int q = 0; q < nrc; ++q) {
    const int RC = rcs[q][0], NT = rcs[q][1];
    if (nt_force && NT != nt_force) continue;
    for (int ntx = 1; ntx <= max_wg && ntx <= nx; ++ntx) {
      const int w = (nx + ntx - 1) / ntx;
      if (nx / ntx < 2 * K) continue;                        // tiles at least 2 K wide: the flat band lists do not overlap
      const int nruns = (w + 2 * K + RC - 1) / RC;
      const int ph_max = NT / nruns;
      if (ph_max <= 2 * K) continue;
      const int nty_min = (Rr + (ph_max - 2 * K) - 1) / (ph_max - 2 * K);
      int nty = std::min(max_wg / ntx, Rr / (2 * K) > 0 ? Rr / (2 * K) : 1);   // (tiles at least 2 K rows tall)
      if (nty < nty_min || nty < 1) continue;
      const int h = (Rr + nty - 1) / nty;
      // the exchange stages the band and the halo of both states as flat lists in ONE free LDS buffer (RC * NT doubles, half per
      // state): the halo list, 2 K (w + h + 2 K) cells, must fit a half (tall narrow tiles of narrow grids would not)
      if ((long long)2 * K * (w + h + 2 * K) > (long long)RC * NT / 2) continue;
      // cost = padded cells per workgroup (what every level works through); within 3 % the geometry with FEWER cells per thread wins
      // (more waves to hide the LDS / issue latency of a level: the pairs are tried in that order), then the one with fewer tiles
      const long long cost = (long long)RC * nruns * (h + 2 * K);
      if (!best.rc || cost * 103 < best.cost * 100 || (cost == best.cost && RC == best.rc && nty * ntx < best.nty * best.ntx)) {
        best.rc = RC; best.nt = NT; best.nty = nty; best.ntx = ntx; best.nruns = nruns; best.cost = cost;
      }
    }
  }
  return best;
}

// Exchange planes, tile flags and the failure word: ONE set per device and process, shared by every plan (resident kernels of a process
// run one at a time, see below), grow-only and NEVER handed back to the allocator.  Round 4's fuzzing showed why: with per-plan
// uncached allocations that were freed with their plan, one later filter in a few thousand -- of any kind, never one that ran on the
// chip itself -- came out wrong (tools/fuzz_gpu.py under GCMF_RESIDENT=1): memory that has been mapped uncached and written past the L2
// must not come back as an ordinary cached allocation while the L2 may still hold lines of its earlier life.
struct ResArena {
  char *ex = nullptr;
  size_t ex_bytes = 0;
  unsigned *flags = nullptr;   // 1024 words: one epoch word per tile (<= 256 tiles); word 1000 = the serial number of a launch that timed out
  unsigned *fail_host = nullptr, *fail_dev = nullptr;
  unsigned epoch = 0;
  unsigned serial = 0;
  unsigned bar_count = 0;          // arrivals the persistent strip-marching launches (k_ringc_one) have added to flags[1001] so far
  hipEvent_t chain_ev = nullptr;   // end of the last resident launch of this process on the device
  bool chain_set = false;
  // Two PROCESSES must not run resident kernels on one GPU at the same time (each would hold CUs the other's missing workgroups wait
  // for).  A process that wants to run them takes an advisory lock on a file named after the GPU (flock, kept until the process ends:
  // no cost per call); a process that does not get it runs the strip-marching launches instead -- the same bits -- and asks again later.
  int lock_fd = -1;
  bool have_lock = false;
  std::chrono::steady_clock::time_point last_try{};
  bool tried = false;
  bool disabled = false;           // a launch of this process timed out all the same (a process outside the lock's reach): strips from now on
  unsigned failed_serial = 0;      // serial number of that launch (0: none): handed ONCE to the plan that issued it (resident_take_failure)
  unsigned long long failures = 0; // time-outs seen on this device by this process (gcmf_resident_status)
  bool lock_busy = false;          // the last attempt at the lock found another process holding it
  std::chrono::steady_clock::time_point last_use{};   // last resident launch (the idle watchdog releases the lock some seconds later)
};

}  // namespace gcmf

using namespace gcmf;

namespace gcmf {
void resident_free(gcmf_plan *pl) { pl->resident = nullptr; }   // (nothing per plan any more)
}  // namespace gcmf

// Resident kernels of ONE process run one at a time, whatever streams they are launched on (two plans filtered from two threads): two
// of them interleaved on the chip would each hold CUs the other's missing workgroups need.  A process-wide chain of events does it
// without touching the host: every resident launch waits for the previous one's end.  (Two PROCESSES on one GPU cannot be chained; see
// the header of this file.)  The same lock guards the arena.
// (both live for ever: the idle watchdog below may still look at them while the process's static destructors run)
static std::mutex &g_chain_mu = *new std::mutex;
static std::map<int, ResArena> &g_arena = *new std::map<int, ResArena>;   // by device ordinal (a node in CPX mode shows 64 devices)

// ---- the lock file ------------------------------------------------------------------------------------------------------------
// /dev/shm is shared by every user of the machine -- it has to be: the lock keeps OTHER users' processes on this GPU off the on-chip
// kernel too -- so the name is predictable and anybody may have put something there first (advisor, round 5).  The file is therefore
// never followed through a symbolic link (O_NOFOLLOW), has its mode set only by the process that CREATED it (O_CREAT | O_EXCL), is opened
// read-only otherwise (flock needs no write access), and is refused unless it is a regular file with a single link.
static int res_open_lock(const std::string &path) {
  for (int attempt = 0; attempt < 3; ++attempt) {
    int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0 && errno == ENOENT) {
      fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0666);
      if (fd < 0 && errno == EEXIST) continue;          // somebody created it between the two calls: open theirs
      if (fd >= 0) (void)fchmod(fd, 0666);              // ours: other users' processes share the GPU too (umask)
    }
    if (fd < 0) return -1;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_nlink != 1) { close(fd); return -1; }
    return fd;
  }
  return -1;
}

// ---- the idle watchdog: a process that ran an on-chip kernel once must not keep every other process of the GPU off that path until
// it exits (an idle notebook).  GCMF_RESIDENT_LOCK_IDLE_S seconds (default 5; 0 = keep it until exit) after the last
// resident launch of this process has FINISHED the lock is given back; the next small grid asks for it again.
static std::mutex &g_wd_mu = *new std::mutex;
static std::condition_variable &g_wd_cv = *new std::condition_variable;
static bool g_wd_stop = false, g_wd_started = false;
static std::thread *g_wd_thread = nullptr;

static double res_idle_seconds() {
  static const double v = getenv("GCMF_RESIDENT_LOCK_IDLE_S") ? atof(getenv("GCMF_RESIDENT_LOCK_IDLE_S")) : 5.0;
  return v;
}

static void res_watchdog() {
  const auto idle = std::chrono::duration<double>(res_idle_seconds());
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(g_wd_mu);
      if (g_wd_cv.wait_for(lk, std::chrono::milliseconds(500), [] { return g_wd_stop; })) return;
    }
    std::lock_guard<std::mutex> chain(g_chain_mu);
    const auto now = std::chrono::steady_clock::now();
    for (auto &kv : g_arena) {
      ResArena &st = kv.second;
      if (!st.have_lock || st.lock_fd < 0 || now - st.last_use < idle) continue;
      if (st.chain_set && hipEventQuery(st.chain_ev) != hipSuccess) { (void)hipGetLastError(); continue; }   // (still running / queued)
      if (flock(st.lock_fd, LOCK_UN) == 0) {
        st.have_lock = false;
        st.tried = false;
      }
    }
  }
}

static void res_watchdog_stop() {   // atexit: registered after HIP's own handlers, so it runs before the runtime is torn down
  {
    std::lock_guard<std::mutex> lk(g_wd_mu);
    g_wd_stop = true;
  }
  g_wd_cv.notify_all();
  if (g_wd_thread && g_wd_thread->joinable()) g_wd_thread->join();
}

static void res_watchdog_start() {   // (g_chain_mu held)
  if (g_wd_started || res_idle_seconds() <= 0.0) return;
  g_wd_started = true;
  g_wd_thread = new std::thread(res_watchdog);
  atexit(res_watchdog_stop);
}

// (g_chain_mu held)  a time-out since the last look?  -> strips from now on, the serial remembered for the plan that issued it
static void res_note_failure(ResArena &st) {
  if (!st.fail_host) return;
  const unsigned f = __atomic_load_n(st.fail_host, __ATOMIC_ACQUIRE);
  if (!f) return;
  __atomic_store_n(st.fail_host, 0u, __ATOMIC_RELEASE);
  st.disabled = true;
  st.failed_serial = f;
  ++st.failures;
}

// (g_chain_mu held)  May this process run resident kernels on `dev` now?  why (optional): GCMF_RESIDENT_* reason when not.
static bool res_process_allowed(int dev, int *why = nullptr) {
  ResArena &st = g_arena[dev];
  res_note_failure(st);
  if (st.disabled) { if (why) *why = GCMF_RESIDENT_DISABLED; return false; }
  // A CU mask (HSA_CU_MASK, ROC_GLOBAL_CU_MASK) takes compute units away while hipDeviceAttributeMultiprocessorCount still reports all
  // of them: "one workgroup per CU, all resident" no longer holds (advisor, round 4).  Such a process takes the strip-marching launches.
  static const bool cu_masked = (getenv("HSA_CU_MASK") && *getenv("HSA_CU_MASK")) || (getenv("ROC_GLOBAL_CU_MASK") && *getenv("ROC_GLOBAL_CU_MASK"));
  if (cu_masked) { if (why) *why = GCMF_RESIDENT_OFF; return false; }
  if (st.have_lock) return true;
  static const bool lock_on = !(getenv("GCMF_RESIDENT_LOCK") && atoi(getenv("GCMF_RESIDENT_LOCK")) == 0);
  if (!lock_on) { st.have_lock = true; return true; }
  const auto now = std::chrono::steady_clock::now();
  if (st.tried && now - st.last_try < std::chrono::seconds(1)) {   // (asked a moment ago)
    if (why) *why = st.lock_busy ? GCMF_RESIDENT_LOCK_BUSY : GCMF_RESIDENT_OFF;
    return false;
  }
  st.tried = true;
  st.last_try = now;
  if (st.lock_fd < 0) {
    char bus[64] = "unknown";
    if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "dev%d", dev); }
    for (char *c = bus; *c; ++c) if (*c == ':' || *c == '/') *c = '_';
    const char *own = getenv("GCMF_RESIDENT_LOCK_DIR");   // (tests: a lock namespace of their own)
    const char *dirs[2] = {own && *own ? own : "/dev/shm", "/tmp"};
    for (int q = 0; q < 2 && st.lock_fd < 0; ++q) st.lock_fd = res_open_lock(std::string(dirs[q]) + "/gcmf_resident_" + bus + ".lock");
    if (st.lock_fd < 0) { st.have_lock = true; return true; }   // nowhere to put a lock file: as before (bounded waits, loud failure)
  }
  if (flock(st.lock_fd, LOCK_EX | LOCK_NB) == 0) {
    st.have_lock = true;
    st.lock_busy = false;
    st.last_use = now;
    res_watchdog_start();
  } else {
    st.lock_busy = true;
    if (why) *why = GCMF_RESIDENT_LOCK_BUSY;
  }
  return st.have_lock;
}

template <int KIND, int RC, int NT = 512> static int res_launch(const ResP &P, int nwg, hipStream_t s) {
  const size_t lds = (size_t)((KIND == K_FLUX && RC >= 13) ? 3 : 2) * RC * NT * sizeof(double);
  static bool attr_done = false;
  if (!attr_done) {
    GCMF_HIP(hipFuncSetAttribute((const void *)k_resident<KIND, RC, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  // A plain launch: the grid has at most one workgroup per CU, so every workgroup becomes resident as soon as whatever ran before on
  // the chip has drained (ordinary kernels always finish).  GCMF_RESIDENT_COOP=1 asks the runtime to guarantee it instead
  // (hipLaunchCooperativeKernel: its cooperative queue costs a cross-queue dependency per launch).
  static const bool coop = getenv("GCMF_RESIDENT_COOP") && atoi(getenv("GCMF_RESIDENT_COOP")) != 0;
  if (coop) {
    ResP Pc = P;
    void *args[] = {(void *)&Pc};
    GCMF_HIP(hipLaunchCooperativeKernel((const void *)k_resident<KIND, RC, NT>, dim3(nwg), dim3(NT), args, (unsigned)lds, s));
  } else {
    hipLaunchKernelGGL((k_resident<KIND, RC, NT>), dim3(nwg), dim3(NT), lds, s, P);
    GCMF_HIP(hipGetLastError());
  }
  return GCMF_OK;
}

template <int KIND> static int res_launch_kind(int rc, int nt, const ResP &P, int nwg, hipStream_t s) {
  if (nt == 1024 && rc == 4) return res_launch<KIND, 4, 1024>(P, nwg, s);
  switch (rc) {
    case 4: return res_launch<KIND, 4>(P, nwg, s);
    case 8: return res_launch<KIND, 8>(P, nwg, s);
    case 13: return res_launch<KIND, 13>(P, nwg, s);
    case 16:
      if constexpr (KIND != K_FLUX) return res_launch<KIND, 16>(P, nwg, s);
  }
  set_error("k_resident: no instantiation for %d cells per thread", rc);
  return GCMF_ERR_INVALID_ARG;
}

static int res_kind(const gcmf_plan *pl) {
  // K_MASK plans run the land-zeroed form (K_MASKZ) of the stencil, as k_ringc does (land is masked out of f as it is loaded)
  return pl->kind == K_MASK ? K_MASKZ : pl->kind;
}

// Can L levels with output rows [row_lo, row_hi) of this plan run resident?  (f64 scalar plans without a tripole seam in the region.)
static bool res_supported(const gcmf_plan *pl, int row_lo, int row_hi, int L, ResGeom *g_out, int *r_lo_out, int *r_hi_out, bool *wrap_out) {
  if (!pl || pl->ncomp != 1 || pl->d.dtype != GCMF_F64 || L < 1 || L > RES_MAXL) return false;
  if (!(pl->kind == K_FLUX || pl->kind == K_MASK || pl->kind == K_REG)) return false;
  if (pl->kind == K_MASK && !pl->g.mbits) return false;
  const int rows = pl->g.rows;
  if (row_lo < 0 || row_hi > rows || row_hi <= row_lo) return false;
  const bool whole = (row_lo == 0 && row_hi == rows);
  const bool wrap = whole && pl->g.south_wrap && pl->g.north_wrap;
  int r_lo = wrap ? 0 : std::max(0, row_lo - L), r_hi = wrap ? rows : std::min(rows, row_hi + L);
  if (pl->g.fold && r_hi == rows) return false;          // the tripole seam is not handled here (k_fold_band's job)
  if (whole && !wrap && !pl->g.fold) {
    // closed / clamped ends of a single slab: handled (dead rows beyond, clamped neighbours), as k_ringc does
  }
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (const char *e = getenv("GCMF_RESIDENT_MAX_WG")) ncu = std::min(ncu, std::max(1, atoi(e)));
  // The halo depth K = levels between two tile exchanges: deeper halos cost padded cells (every level works on them), shallower ones
  // cost exchanges.  Measured (experiments/scripts/measure_resident_k.py, us per application at K = 4 / 8 / 12): IRREGULAR 256 x 256 n 63: 92 / 72 / 138;
  // 512 x 512: 110 / 110 / 154; 600 x 640: 143 / 116 / 158; REGULAR 512 x 512 n 36: 55 / 54 / 71; REGULAR_WITH_LAND 720 x 1440 n 56:
  // 200 / 183 / 167.  So: K = 8 where a geometry exists for it, else K = 4 (the 8-way slab of 2400 x 3600 only fits K = 4);
  // GCMF_RESIDENT_K forces one (4, 8, 12).
  ResGeom g;
  static const int k_force = getenv("GCMF_RESIDENT_K") ? atoi(getenv("GCMF_RESIDENT_K")) : 0;
  for (int K : {8, 4, 12}) {
    if (k_force ? K != k_force : K == 12) continue;
    if (pl->g.nx < 2 * K + 16 || r_hi - r_lo < 2 * K) continue;
    const ResGeom c = res_geometry(res_kind(pl), r_hi - r_lo, pl->g.nx, K, ncu);
    if (!c.rc) continue;
    g = c;
    g.K = K;
    break;
  }
  (void)L;
  if (!g.rc) return false;
  if (g_out) *g_out = g;
  if (r_lo_out) *r_lo_out = r_lo;
  if (r_hi_out) *r_hi_out = r_hi;
  if (wrap_out) *wrap_out = wrap;
  return true;
}

namespace gcmf {

// Whether gcmf_apply / gcmf_slab_apply_backward pick the resident kernel BY THEMSELVES (it is bit-identical to the strip-marching launches,
// so this is a question of speed only; measured in round 4, tools/measure_resident.py):
//   * whole small grids (gcmf_apply, `whole`): yes while the tiles fit the 1024-thread geometry (~420 k cells) -- the polynomial runs in ONE launch and the tiles are small enough
//     for the flag exchanges to be cheap: IRREGULAR 512 x 512, n 63: 88 us against 179 us for eight strip-marching launches; at
//     720 x 1440 the two are equal or the strips win.  The REGULAR / land-mask kinds (cheaper levels, two strip launches for 16 levels)
//     only from 24 levels on: 512 x 512 n 36 51.7 against 54.5 us, but n 16 (BASELINE config 1) 25.4-28.9 against 23.6-27.1 us;
//   * row slabs of a multi-GPU run: no -- on the 8-way slab of 2400 x 3600 a tile exchange costs ~9.5 us against ~1 us per level and
//     registers + LDS only hold a K = 4 halo: 0.336 against 0.307 ms per application.
// env GCMF_RESIDENT=1 forces it wherever it fits, =0 forbids it; the building block (gcmf_resident_levels) is always available.
bool resident_supported(const gcmf_plan *pl, int row_lo, int row_hi, int L, int n_total, int *why) {
  if (why) *why = GCMF_RESIDENT_OFF;
  const char *e = getenv("GCMF_RESIDENT");
  const int mode = e ? atoi(e) : -1;
  if (mode == 0) return false;
  ResGeom g;
  if (!res_supported(pl, row_lo, row_hi, L, &g, nullptr, nullptr, nullptr)) return false;
  if (mode < 0) {
    // auto: whole grids whose tiles are small enough for the 1024-thread / four-cells-per-thread geometry (up to ~420 k cells on 256
    // CUs) -- tools/measure_resident_sizes.py: on chip / strips = 0.46-0.56 (IRREGULAR n 63), 0.55-0.69 (REGULAR_WITH_LAND n 56),
    // 0.67-0.86 (REGULAR n 56) at 147-410 k cells, and 1.0-1.4 from 640 k cells on, where the tiles need 13-16 cells per thread
    static const long long max_cells = getenv("GCMF_RESIDENT_MAX_CELLS") ? atoll(getenv("GCMF_RESIDENT_MAX_CELLS")) : 0;
    const bool whole = pl && row_lo == 0 && row_hi == pl->g.rows && pl->full;
    if (!whole) return false;
    if (max_cells > 0 ? (long long)pl->g.rows * pl->g.nx > max_cells : !(g.rc == 4 && g.nt == 1024)) return false;
    if (pl->kind != K_FLUX && n_total < 24) return false;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> chain(g_chain_mu);
  return res_process_allowed(dev, why);   // (another process of this GPU runs resident kernels: the strip-marching launches, same bits)
}

// Did one of the resident launches with serial numbers [lo, hi] (a plan's own, wrap-around safe) time out?  Reported ONCE, to the plan
// that issued the launch -- whose output it poisoned with NaN -- never to an unrelated call (advisor, round 5): a time-out nobody claims
// stays visible in gcmf_resident_status and switches the process to the strip-marching launches all the same.
bool resident_take_failure(int dev, unsigned lo, unsigned hi) {
  std::lock_guard<std::mutex> chain(g_chain_mu);
  auto it = g_arena.find(dev);
  if (it == g_arena.end()) return false;
  ResArena &st = it->second;
  res_note_failure(st);
  if (!st.failed_serial || !lo) return false;
  if ((unsigned)(st.failed_serial - lo) > (unsigned)(hi - lo)) return false;   // somebody else's launch
  st.failed_serial = 0;
  return true;
}

void resident_status(int dev, int *state, unsigned long long *failures) {
  std::lock_guard<std::mutex> chain(g_chain_mu);
  auto it = g_arena.find(dev);
  int st_ = GCMF_RESIDENT_OFF;
  unsigned long long nf = 0;
  if (it != g_arena.end()) {
    ResArena &st = it->second;
    res_note_failure(st);
    nf = st.failures;
    st_ = st.disabled ? GCMF_RESIDENT_DISABLED : st.have_lock ? GCMF_RESIDENT_OK : st.lock_busy ? GCMF_RESIDENT_LOCK_BUSY : GCMF_RESIDENT_OFF;
  }
  if (state) *state = st_;
  if (failures) *failures = nf;
}

bool resident_fits(const gcmf_plan *pl, int row_lo, int row_hi, int L) { return res_supported(pl, row_lo, row_hi, L, nullptr, nullptr, nullptr, nullptr); }

// (g_chain_mu held)  the chain event, the flag words and the failure word of a device's arena
static int res_arena_ready(ResArena *st) {
  if (!st->chain_ev) GCMF_HIP(hipEventCreateWithFlags(&st->chain_ev, hipEventDisableTiming));
  if (!st->flags) {
    GCMF_HIP(hipExtMallocWithFlags((void **)&st->flags, 1024 * sizeof(unsigned), hipDeviceMallocUncached));
    GCMF_HIP(hipMemset(st->flags, 0, 1024 * sizeof(unsigned)));
    GCMF_HIP(hipHostMalloc((void **)&st->fail_host, 64, hipHostMallocMapped));
    *st->fail_host = 0u;
    GCMF_HIP(hipHostGetDevicePointer((void **)&st->fail_dev, st->fail_host, 0));
  }
  return GCMF_OK;
}

int resident_persistent_launch(int dev, hipStream_t s, unsigned nbar, const std::function<int(unsigned *, unsigned *, unsigned, unsigned)> &launch) {
  std::lock_guard<std::mutex> chain(g_chain_mu);
  if (!res_process_allowed(dev)) {
    set_error("persistent launch: another process runs persistent kernels on this GPU (or an earlier one of this process timed out)");
    return GCMF_ERR_UNSUPPORTED;
  }
  ResArena *st = &g_arena[dev];
  { const int rc_ = res_arena_ready(st); if (rc_) return rc_; }
  if (st->chain_set) GCMF_HIP(hipStreamWaitEvent(s, st->chain_ev, 0));
  struct Mark {
    hipEvent_t e; hipStream_t s; bool *set;
    ~Mark() { if (hipEventRecord(e, s) == hipSuccess) *set = true; }
  } mark{st->chain_ev, s, &st->chain_set};
  st->last_use = std::chrono::steady_clock::now();
  if (++st->serial == 0) st->serial = 1;
  const unsigned bar0 = st->bar_count;
  st->bar_count += nbar;
  return launch(st->flags, st->fail_dev, st->serial, bar0);
}

// L levels (a.S is ignored: pk = the L coefficients) of the backward evaluation on rows [a.row_lo, a.row_hi), one launch.
// The plan's mutex is held and its device is current.
int launch_resident(gcmf_plan *pl, const MultiArgs &a, const double *pk, int L, hipStream_t s) {
  ResGeom g;
  int r_lo = 0, r_hi = 0;
  bool wrap = false;
  if (!res_supported(pl, a.row_lo, a.row_hi, L, &g, &r_lo, &r_hi, &wrap)) {
    set_error("k_resident: rows [%d, %d) x %d levels of this plan do not fit on the chip", a.row_lo, a.row_hi, L);
    return GCMF_ERR_UNSUPPORTED;
  }
  if (a.nbatch != 1) {
    set_error("k_resident: one field per launch");
    return GCMF_ERR_UNSUPPORTED;
  }
  int dev = 0;
  GCMF_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> chain(g_chain_mu);
  if (!res_process_allowed(dev)) {
    set_error("k_resident: another process runs resident kernels on this GPU (or an earlier resident launch of this process timed out): "
              "use the strip-marching launches (gcmf_apply does by itself)");
    return GCMF_ERR_UNSUPPORTED;
  }
  ResArena *st = &g_arena[dev];
  const size_t plane = (size_t)pl->g.rows * pl->g.nx * sizeof(double);
  { const int rc_ = res_arena_ready(st); if (rc_) return rc_; }
  if (st->ex_bytes < 4 * plane) {
    // grow: a new, larger block; the old one stays where it is (an earlier launch may still be using it, and uncached memory is never
    // handed back -- see ResArena)
    const size_t want = std::max(4 * plane, 2 * st->ex_bytes);
    char *blk = nullptr;
    GCMF_HIP(hipExtMallocWithFlags((void **)&blk, want, hipDeviceMallocUncached));
    st->ex = blk;
    st->ex_bytes = want;
  }
  if (st->chain_set) GCMF_HIP(hipStreamWaitEvent(s, st->chain_ev, 0));
  struct Mark {
    hipEvent_t e; hipStream_t s; bool *set;
    ~Mark() { if (hipEventRecord(e, s) == hipSuccess) *set = true; }
  } mark{st->chain_ev, s, &st->chain_set};
  const Geom &gm = pl->g;
  ResP P{};
  P.u0 = (const double *)a.u0; P.v0 = (const double *)a.v0; P.uo = (double *)a.uo; P.vo = (double *)a.vo;
  P.f = (const double *)a.fb_in; P.out = (double *)a.fb_out;
  P.cE = (const double *)gm.coef[0]; P.cN = (const double *)gm.coef[1]; P.ra = (const double *)gm.coef[2];
  P.mbits = gm.mbits;
  P.lbits = (pl->n_land > 0) ? pl->lbits : nullptr;
  P.area = (pl->kind != K_FLUX && gm.area_weighted) ? (const double *)gm.area : nullptr;
  for (int par = 0; par < 2; ++par)
    for (int q = 0; q < 2; ++q) P.ex[par][q] = (double *)(st->ex + (size_t)(par * 2 + q) * plane);
  P.flags = st->flags;
  P.fail = st->fail_dev;
  P.dfail = st->flags + 1000;
  st->last_use = std::chrono::steady_clock::now();
  if (++st->serial == 0) st->serial = 1;
  P.serial = st->serial;
  if (!pl->res_lo) pl->res_lo = st->serial;   // (the plan's mutex is held: the launches of ITS application, should one of them time out)
  pl->res_hi = st->serial;
  P.epoch0 = st->epoch;
  P.nx = gm.nx; P.rows = gm.rows;
  P.r_lo = r_lo; P.r_hi = r_hi; P.out_lo = a.row_lo; P.out_hi = a.row_hi;
  P.nty = g.nty; P.ntx = g.ntx; P.nruns = g.nruns; P.K = g.K; P.L = L;
  P.wrap = wrap ? 1 : 0;
  P.first = a.first; P.last = a.last;
  const int nwg = g.nty * g.ntx;
  P.xcd_per = (pl->xcd_remap && nwg % 8 == 0) ? nwg / 8 : 0;
  for (int t = 0; t < RES_MAXL; ++t) P.pk[t] = t < L ? pk[t] : 0.0;
  P.p0 = a.p0; P.c = a.c;
  long long ms = 10000;   // a tile waits this long for a neighbour (another kernel may be holding CUs for a while); a real clash ends here, loudly
  if (const char *e = getenv("GCMF_RESIDENT_TIMEOUT_MS")) ms = std::max(1LL, atoll(e));
  P.spin_limit = ms * 100000LL;
  st->epoch += (unsigned)((L - 1) / P.K);
  int rc;
  switch (res_kind(pl)) {
    case K_FLUX: rc = res_launch_kind<K_FLUX>(g.rc, g.nt, P, nwg, s); break;
    case K_MASKZ: rc = res_launch_kind<K_MASKZ>(g.rc, g.nt, P, nwg, s); break;
    default: rc = res_launch_kind<K_REG>(g.rc, g.nt, P, nwg, s); break;
  }
  if (rc) return rc;
  char geom[160];
  snprintf(geom, sizeof geom, "tiles=%dx%d RC=%d NT=%d nruns=%d K=%d L=%d rowlo=%d rowhi=%d", g.nty, g.ntx, g.rc, g.nt, g.nruns, P.K, L, r_lo, r_hi);
  note_kernel(pl, std::string("gcmf::k_resident<") + std::to_string(res_kind(pl)) + ", " + std::to_string(g.rc) + ", " + std::to_string(g.nt) + ">", L, geom);
  return GCMF_OK;
}

}  // namespace gcmf
