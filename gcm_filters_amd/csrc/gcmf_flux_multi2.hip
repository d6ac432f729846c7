// Flux-form scalar kinds, deep temporal blocking, TWO rows per loop iteration.
//
// Same algorithm, wave mapping, operands and arithmetic as k_scalar_multi<T, FB, K_FLUX, S, 1> (gcmf_scalar_multi.hip;
// reference gcm_filters/kernels.py:259-315, 345-372, 402-429, 517-585 inside the recurrence of filter.py:162-212) -- the
// results are bit-identical -- but the row loop is unrolled by two: the register windows (4 row slots per level instead
// of 3) and the lag lines of the coefficient rows and of fbar (one entry more each) are shifted by two rows once per
// pair instead of by one row per row.  That shifting is a third of the one-row kernel's VALU instructions (SQ counters,
// profiles/r01/cfg3_sq_counters.txt; cost probe in experiments/README.md), and the kernel is VALU-issue bound at its one
// wave per SIMD.
#include "gcmf_flux_multi2_body.hpp"

namespace gcmf {

template <typename T, typename FB, int S>
__global__ __launch_bounds__(256, 1) void k_flux_multi2(const MultiP<T, FB> P) {
  flux_multi2_march<T, FB, S>(P, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

template <typename T, typename FB, int S> static int launch_f2(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
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
  P.mbits = nullptr;
  P.area = nullptr;
  P.nx = g.nx;
  P.rows = g.rows;
  P.out_lo = a.row_lo;
  P.out_hi = a.row_hi;
  const int nrows = a.row_hi - a.row_lo;
  if (nrows <= 0 || a.nbatch <= 0) return GCMF_OK;
  P.nwx = (g.nx + WI - 1) / WI;
  int H = pl->strip_rows;
  if (H <= 0) {  // one resident round of waves at one wave per SIMD (see launch_multi_s)
    long long want = 1024 / ((long long)P.nwx * a.nbatch);
    if (want < 1) want = 1;
    H = (int)((nrows + want - 1) / want);
    if (H < 2 * S) H = 2 * S;
  }
  if (H > nrows) H = nrows;
  P.H = H;
  P.nstrips = (nrows + H - 1) / H;
  P.nwaves = P.nwx * P.nstrips;
  P.wrap = g.south_wrap && g.north_wrap;
  P.first = a.first;
  P.last = a.last;
  P.area_weighted = 0;
  P.bstride = (long long)g.rows * g.nx;
  for (int t = 0; t < MAX_PK; ++t) P.pk[t] = t < S ? a.pk[t] : 0.0;
  P.p0 = a.p0;
  P.c = a.c;
  dim3 block(256), grid((P.nwaves + 3) / 4, (unsigned)a.nbatch);
  hipLaunchKernelGGL((k_flux_multi2<T, FB, S>), grid, block, 0, s, P);
  note_kernel(pl, std::string("gcmf::k_flux_multi2<") + tyname<T>() + ", " + tyname<FB>() + ", " + std::to_string(S) + ">", S);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

// S in 5..8, flux kinds without prepare / finalize (none of them has one)
bool flux_multi2_supported(const gcmf_plan *pl, int S) {
  return pl->kind == K_FLUX && !pl->g.area_weighted && S >= 5 && S <= 8;
}

template <typename T, typename FB> static int launch_f2_s(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  switch (a.S) {
    case 5: return launch_f2<T, FB, 5>(pl, a, s);
    case 6: return launch_f2<T, FB, 6>(pl, a, s);
    case 7: return launch_f2<T, FB, 7>(pl, a, s);
    case 8: return launch_f2<T, FB, 8>(pl, a, s);
  }
  return GCMF_ERR_INVALID_ARG;
}

int launch_flux_multi2(gcmf_plan *pl, const MultiArgs &a, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_f2_s<double, double>(pl, a, s);
  if (a.fb_is_f32) return launch_f2_s<float, float>(pl, a, s);
  return launch_f2_s<float, double>(pl, a, s);
}

}  // namespace gcmf
