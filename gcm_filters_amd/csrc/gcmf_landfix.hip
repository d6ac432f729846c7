// Land kept out of the state (see gcmf_plan::lbits): the result of a cell that exchanges nothing with its neighbours.
//
// Such a cell has L = 0 at every step (reference: the land mask / the closed faces zero every flux, kernels.py:163-187,
// 286-315), so its recurrence T_k = 2(-T_{k-1} - c 0) - T_{k-2}, fbar += p_k T_k (filter.py:162-212) involves nothing but
// its own input value.  gcmf_apply runs the blocked kernels with these cells zeroed and calls this kernel at the end: it
// replays the recurrence with the same operations, in the same precision and order as the stencil kernels (the helpers of
// gcmf_recurrence.hpp), prepare / finalize included, and writes the result over `out` on those cells only.
#include "gcmf_multi_common.hpp"

namespace gcmf {

template <typename T, typename FB, bool FUSED>
__global__ __launch_bounds__(256) void k_land_fix(const T *in, FB *out, const uint8_t *lbits, const T *area,
                                                  const double *p, int n_steps, double c_, long long ncell,
                                                  long long ntotal) {
  extern __shared__ double sp[];  // p[0..n_steps], read by every step of every cell
  for (int k = threadIdx.x; k <= n_steps; k += blockDim.x) sp[k] = p[k];
  __syncthreads();
  const T c = (T)c_;
  // four consecutive cells per thread (ncell is a multiple of 4: one field, one word of the byte plane): four independent
  // dependency chains per lane; cells that do exchange with neighbours are computed along and not stored
  for (long long q4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; q4 < ntotal; q4 += (long long)gridDim.x * blockDim.x * 4) {
    const long long cell = q4 % ncell;
    const unsigned m = *reinterpret_cast<const unsigned *>(lbits + cell);
    if ((m & 0x01010101u) == 0x01010101u) continue;
    T xm2[4], xm1[4];
    FB fb[4];
    bool finite = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      T x = in[q4 + j];
      if (area) x = x * area[cell + j];                 // prepare()
      const T a = cheb_a<FUSED>(x, c, T(0));
      fb[j] = cheb_acc_first<FUSED, T, FB>(sp[0], sp[1], x, a);
      xm2[j] = x;
      xm1[j] = a;
      // |x| <= max / 2 of the STATE type: the unfused forward kinds form 2 * a, which overflows beyond that and the general loop's bits
      // would differ (advisor, round 3)
      finite = finite && (mabs(x) <= MLim<T>::big() * T(0.5));
    }
    if (finite) {
      // With L = 0 and a finite x every step is exact up to the running sum: A(T) = -T, T_k = 2 A(T_{k-1}) - T_{k-2} = (-1)^k x, so only
      // fbar's chain is left (one dependent operation per step instead of three) -- the SAME bits: p_k * (-x) = -(p_k * x), fused or not.
      // (an infinite x makes T_2 = inf - inf = NaN in the recurrence: those cells take the general loop below)
      for (int k = 2; k <= n_steps; ++k) {
        const double pk = sp[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = cheb_acc<FUSED, T, FB>(fb[j], pk, (k & 1) ? xm1[j] : xm2[j]);   // xm1 = -x, xm2 = x
      }
    } else {
      for (int k = 2; k <= n_steps; ++k) {
        const double pk = sp[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T a = cheb_a<FUSED>(xm1[j], c, T(0));
          const T tk = cheb_t<FUSED>(a, xm2[j]);
          fb[j] = cheb_acc<FUSED, T, FB>(fb[j], pk, tk);
          xm2[j] = xm1[j];
          xm1[j] = tk;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((m >> (8 * j)) & 1u) continue;
      FB r = fb[j];
      if (area) r = r / (FB)area[cell + j];             // finalize()
      out[q4 + j] = r;
    }
  }
}

// zero the isolated cells of two state planes in place (the outputs of the first blocked launch): from then on NaN on land
// cannot reach the NaN / inf bookkeeping of the blocked kernels.  Reads the byte plane, writes land cells only.
// Cells [cell0, cell0 + nsub) of every field (both multiples of 4: the caller checks nx % 4 == 0).
template <typename T>
__global__ __launch_bounds__(256) void k_zero_land(T *a, T *b, const uint8_t *lbits, long long ncell, long long cell0, long long nsub,
                                                   long long ntotal) {
  for (long long q4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; q4 < ntotal; q4 += (long long)gridDim.x * blockDim.x * 4) {
    const long long field = q4 / nsub, cell = cell0 + (q4 - field * nsub);  // 4 cells of one field
    const unsigned m = *reinterpret_cast<const unsigned *>(lbits + cell);
    if ((m & 0x01010101u) == 0x01010101u) continue;
    const long long o = field * ncell + cell;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (!((m >> (8 * k)) & 1u)) { a[o + k] = T(0); b[o + k] = T(0); }
  }
}

// rows [row_lo, row_hi) of both planes (row_hi <= 0: all rows)
int launch_zero_land(gcmf_plan *pl, void *a, void *b, int64_t nbatch, hipStream_t s, int row_lo, int row_hi) {
  if (row_hi <= 0) { row_lo = 0; row_hi = (int)pl->rows_alloc; }
  const long long ncell = (long long)pl->rows_alloc * pl->d.nx, cell0 = (long long)row_lo * pl->d.nx;
  const long long nsub = (long long)(row_hi - row_lo) * pl->d.nx, ntotal = nsub * nbatch;
  if (ntotal <= 0) return GCMF_OK;
  long long nb = (ntotal / 4 + 255) / 256;
  if (nb > 65536) nb = 65536;
  dim3 block(256), grid((unsigned)nb);
  if (pl->d.dtype == GCMF_F64)
    hipLaunchKernelGGL(k_zero_land<double>, grid, block, 0, s, (double *)a, (double *)b, pl->lbits, ncell, cell0, nsub, ntotal);
  else
    hipLaunchKernelGGL(k_zero_land<float>, grid, block, 0, s, (float *)a, (float *)b, pl->lbits, ncell, cell0, nsub, ntotal);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

template <typename T, typename FB>
static int launch_lf(gcmf_plan *pl, const void *in, void *out, const double *dp, int n_steps, double c, int64_t nbatch,
                     hipStream_t s) {
  const long long ncell = (long long)pl->rows_alloc * pl->d.nx, ntotal = ncell * nbatch;
  const T *area = pl->area_weighted ? (const T *)pl->g.area : nullptr;
  long long nb = (ntotal / 4 + 255) / 256;
  if (nb > 32768) nb = 32768;
  dim3 block(256), grid((unsigned)nb);
  const size_t lds = ((size_t)n_steps + 1) * sizeof(double);
  if (pl->kind == K_FLUX)
    hipLaunchKernelGGL((k_land_fix<T, FB, true>), grid, block, lds, s, (const T *)in, (FB *)out, pl->lbits, area, dp, n_steps, c,
                       ncell, ntotal);
  else
    hipLaunchKernelGGL((k_land_fix<T, FB, false>), grid, block, lds, s, (const T *)in, (FB *)out, pl->lbits, area, dp, n_steps,
                       c, ncell, ntotal);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

int launch_land_fix(gcmf_plan *pl, const void *in, void *out, const double *dp, int n_steps, double c, int fb_is_f32,
                    int64_t nbatch, hipStream_t s) {
  if (pl->d.dtype == GCMF_F64) return launch_lf<double, double>(pl, in, out, dp, n_steps, c, nbatch, s);
  if (fb_is_f32) return launch_lf<float, float>(pl, in, out, dp, n_steps, c, nbatch, s);
  return launch_lf<float, double>(pl, in, out, dp, n_steps, c, nbatch, s);
}

}  // namespace gcmf
