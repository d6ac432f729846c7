// k_ringc_one: the north star's literal form -- "the whole n_steps polynomial fused into a SINGLE launch" -- at BASELINE size, for whole f64
// flux-form grids without a tripole seam (BASELINE config 3; reference filter.py:162-212 / kernels.py:297-315 as restated in
// gcmf_ringc_impl.hpp).  OPT-IN (gcmf_set_option(plan, "single_launch", 1) / env GCMF_SINGLE_LAUNCH=1), measured, and NOT the default:
//
//   the same strip marches as k_ringc<double, K_FLUX, S> -- a 2400 x 3600 field does not fit the chip, so every S levels are still one
//   pass over HBM -- but the n / S passes run inside ONE persistent launch: between two passes every workgroup writes its L2's dirty
//   lines back, arrives at a counter in uncached memory, waits for the others, and invalidates its caches (the eight XCDs' L2s are
//   not coherent with each other inside a kernel).  That barrier costs more than the launch boundary it replaces (1.7 us), which is
//   why the default stays with back-to-back launches (DESIGN.md 3.1 has the numbers).
//
// Deadlock freedom as k_resident's (gcmf_resident.hip): one wave per SIMD, at most one workgroup per CU and no more workgroups than CUs,
// so all of them are resident once whatever ran before has drained; launched under the process's on-chip lock and chained behind its
// other persistent launches; every wait is bounded -- a wait that runs out raises the launch's failure word (reported ONCE, to the plan
// that issued it), every workgroup that sees it stops marching and fills its rows of the result with NaN.
// Same bits as the back-to-back launches (the same instruction stream per pass).
#include "gcmf_ringc_impl.hpp"
#include "gcmf_api_internal.hpp"

#include <algorithm>
#include <cstdlib>

namespace gcmf {

constexpr int ONE_MAXP = 16;   // passes per launch

template <int S> struct OneP {
  MultiP<double, double> base;   // everything the passes share (coefficients, geometry, the constant input, the result plane)
  const double *u0[ONE_MAXP], *v0[ONE_MAXP];
  double *uo[ONE_MAXP], *vo[ONE_MAXP];
  double pk[ONE_MAXP][S];
  int npass, nwg;
  unsigned *bar;         // the arrival counter (uncached device memory, never reset: bar0 = its value when this launch was enqueued)
  unsigned bar0;
  unsigned *fail_host;   // mapped host word: the serial number of a launch that timed out
  unsigned *dfail;       // the same for the workgroups (uncached device memory)
  unsigned serial;
  long long spin_limit;  // in s_memrealtime ticks (100 MHz)
  int debug_skip;        // tests: workgroup 0 does not arrive at the second barrier (option "single_launch" = 2): everybody's wait runs out
};

// all workgroups of the launch meet here; false = the wait ran out (or another workgroup's did)
template <int S> __device__ __forceinline__ bool one_barrier(const OneP<S> &P, const int q) {
  __builtin_amdgcn_s_waitcnt(0);   // this wave's stores have reached the L2
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the L2's dirty lines go to the memory side (the other XCDs read them from there)
    if (!(P.debug_skip && blockIdx.x == 0 && q == 1)) __hip_atomic_fetch_add(P.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = P.bar0 + (unsigned)P.nwg * (unsigned)(q + 1);
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    int ok = 1;
    while ((int)(__hip_atomic_load(P.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      if (__hip_atomic_load(P.dfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.serial) { ok = 0; break; }
      if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > P.spin_limit) {
        __hip_atomic_store(P.dfail, P.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.fail_host, P.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 and this XCD's L2 forget what they held of the other XCDs' rows
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

template <int S>
__global__ __launch_bounds__(256, 1) void k_ringc_one(const OneP<S> P) {
  int bx = blockIdx.x;
  if (P.base.xcd_per > 0 && bx < 8 * P.base.xcd_per) bx = (bx & 7) * P.base.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mine = wid < P.base.nwaves;   // (the last workgroup may have idle waves: they keep the barriers)
  bool ok = true;
  for (int q = 0; q < P.npass; ++q) {
    if (mine && ok) {
      MultiP<double, double> Q = P.base;
      Q.u0 = P.u0[q];  Q.v0 = P.v0[q];  Q.uo = P.uo[q];  Q.vo = P.vo[q];
#pragma unroll
      for (int t = 0; t < S; ++t) Q.pk[t] = P.pk[q][t];
      Q.first = (q == 0) ? 1 : 0;
      Q.last = (q == P.npass - 1) ? 1 : 0;
      if (q == 0) ringc_walk<double, K_FLUX, S, true, false, false, false>(Q, wid);
      else ringc_walk<double, K_FLUX, S, false, false, false, false>(Q, wid);
    }
    if (q + 1 < P.npass) ok = one_barrier<S>(P, q) && ok;
  }
  if (!ok && mine) {   // a wait ran out somewhere: never a plausible-but-wrong result -- this wave's rows of the result are NaN
    constexpr int M = (S + 1) / 2 * 2, WI = 128 - 2 * M;
    const int wx = wid % P.base.nwx, st = wid / P.base.nwx, lane = threadIdx.x & 63;
    const int a = P.base.out_lo + st * P.base.H, b = min(a + P.base.H, P.base.out_hi);
    const double poison = __longlong_as_double(0x7ff8000000000000LL);
    for (int k = 0; k < 2; ++k) {
      const int col = wx * WI + 2 * lane + k;
      if (2 * lane + k < WI && col < P.base.nx)
        for (int r = a; r < b; ++r) P.base.fb_out[(long long)r * P.base.nx + col] = poison;
    }
  }
}

// can this application run as ONE launch?  whole f64 flux grid without a seam, one field, equal passes of 9 or 8 levels
int ringc_one_depth(const gcmf_plan *pl, int n_steps, int64_t nbatch) {
  if (!pl->single_launch || !ringc9_ok(pl) || nbatch != 1 || pl->strip_rows > 0) return 0;
  for (int S : {9, 8})
    if (n_steps % S == 0 && n_steps / S >= 1 && n_steps / S <= ONE_MAXP) return S;
  return 0;
}

template <int S> static int launch_one(gcmf_plan *pl, const double *p, int n_steps, double c, const void *f, void *out, void *const *pool, hipStream_t s) {
  constexpr int M = (S + 1) / 2 * 2, WI = 128 - 2 * M, R = RingGeom::R;
  const Geom &g = pl->g;
  OneP<S> P{};
  MultiP<double, double> &B = P.base;
  B.fb_in = (const double *)f;
  B.fb_out = (double *)out;
  B.d_out = nullptr;
  B.cE = (const double *)g.coef[0];
  B.cN = (const double *)g.coef[1];
  B.ra = (const double *)g.coef[2];
  B.zrow = (const double *)pl->zero_row;
  B.nfb = pl->ring_nfb;
  B.mbits = g.mbits;
  B.lbits = (pl->n_land > 0) ? pl->lbits : nullptr;
  B.area = (const double *)g.area;
  B.nx = g.nx;
  B.rows = g.rows;
  B.out_lo = 0;
  B.out_hi = g.rows;
  B.nwx = (g.nx + WI - 1) / WI;
  const long long want = strips_per_column(B.nwx, g.rows, S, R);   // (as launch_ringc_sf: whole ring periods)
  int H = (int)((g.rows + want - 1) / want);
  if (H < 4) H = 4;
  H += (R - (H + 2 * S) % R) % R;
  if (H > g.rows) H = g.rows;
  B.H = H;
  B.nstrips = (g.rows + H - 1) / H;
  B.nwaves = B.nwx * B.nstrips;
  B.npack = 0;
  B.wrap = g.south_wrap && g.north_wrap;
  B.area_weighted = 0;
  B.bstride = (long long)g.rows * g.nx;
  B.p0 = p[n_steps];
  B.c = c;
  B.zigzag = pl->zigzag;
  P.npass = n_steps / S;
  P.nwg = (B.nwaves + 3) / 4;
  B.xcd_per = pl->xcd_remap ? P.nwg / 8 : 0;
  int dev = 0, ncu = 0;
  GCMF_HIP(hipGetDevice(&dev));
  GCMF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  if (P.nwg > ncu) {
    set_error("k_ringc_one: %d workgroups do not fit %d compute units at one each", P.nwg, ncu);
    return GCMF_ERR_UNSUPPORTED;
  }
  // the state rotates through the pool of four planes exactly as the back-to-back launches rotate it
  const void *u = nullptr, *v = nullptr;
  for (int q = 0; q < P.npass; ++q) {
    void *fr[2] = {nullptr, nullptr};
    int nf = 0;
    for (int k = 0; k < 4 && nf < 2; ++k)
      if (pool[k] != u && pool[k] != v) fr[nf++] = pool[k];
    P.u0[q] = (const double *)u;  P.v0[q] = (const double *)v;
    P.uo[q] = (double *)fr[0];  P.vo[q] = (double *)fr[1];
    for (int t = 0; t < S; ++t) P.pk[q][t] = p[n_steps - (q * S + 1 + t)];
    u = fr[0];  v = fr[1];
  }
  long long ms = 2000;
  if (const char *e = getenv("GCMF_RESIDENT_TIMEOUT_MS")) ms = std::max(1LL, atoll(e));
  P.spin_limit = ms * 100000LL;
  P.debug_skip = pl->single_launch == 2 ? 1 : 0;
  const int rc = resident_persistent_launch(dev, s, (unsigned)P.nwg * (unsigned)(P.npass - 1), [&](unsigned *flags, unsigned *fail_dev, unsigned serial, unsigned bar0) -> int {
    P.bar = flags + 1001;
    P.dfail = flags + 1000;
    P.fail_host = fail_dev;
    P.serial = serial;
    P.bar0 = bar0;
    if (!pl->res_lo) pl->res_lo = serial;
    pl->res_hi = serial;
    hipLaunchKernelGGL((k_ringc_one<S>), dim3(P.nwg), dim3(256), 0, s, P);
    GCMF_HIP(hipGetLastError());
    return GCMF_OK;
  });
  if (rc) return rc;
  note_kernel(pl, std::string("gcmf::k_ringc_one<") + std::to_string(S) + ">", n_steps,
              launch_geom(B.H, B.nstrips, B.nwx, B.xcd_per > 0, (unsigned)P.nwg, 1, g.rows));
  return GCMF_OK;
}

int launch_ringc_one(gcmf_plan *pl, int S, const double *p, int n_steps, double c, const void *f, void *out, void *const *pool, hipStream_t s) {
  switch (S) {
    case 9: return launch_one<9>(pl, p, n_steps, c, f, out, pool, s);
    case 8: return launch_one<8>(pl, p, n_steps, c, f, out, pool, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

}  // namespace gcmf
