// The Chebyshev recurrence arithmetic (reference gcm_filters/filter.py:162-212) shared by the scalar kernels, single-step
// (gcmf_scalar.hip) and temporally blocked (gcmf_scalar_multi.hip), so that they agree bit for bit with each other.
//   FUSED = false (REGULAR / land-mask kinds): numpy's operation order and roundings, one IEEE operation per numpy
//           operation -- these kinds are bit-exact against the reference.
//   FUSED = true  (flux-form kinds, whose plan-time folded coefficients already differ from the reference in the last
//           bit): -x - c L, 2 A - T_{k-2} and fbar += p_k T_k are one fused multiply-add each (one rounding and one
//           instruction less per line).
// The vector kernels keep their own (unfused) copies of these lines next to their stencils.
// The library is compiled with -ffp-contract=off: nothing is fused unless it is written here.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace gcmf {

__device__ __forceinline__ double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// A(x) = -x - c L(x)
#ifdef GCMF_NO_FUSE  // A/B switch for experiments
#define GCMF_FUSE_OK false
#else
#define GCMF_FUSE_OK true
#endif

template <bool FUSED, typename T> __device__ __forceinline__ T cheb_a(T x, T c, T L) {
  if (FUSED && GCMF_FUSE_OK) return rfma(-c, L, -x);
  return -x - c * L;
}
// T_k = 2 A(T_{k-1}) - T_{k-2}
template <bool FUSED, typename T> __device__ __forceinline__ T cheb_t(T a, T x2) {
  if (FUSED && GCMF_FUSE_OK) return rfma(T(2), a, -x2);
  return T(2) * a - x2;
}
// fbar += p_k T_k   (fbar is f64 for f32 state unless the caller asked for f32 output)
template <bool FUSED, typename T, typename FB> __device__ __forceinline__ FB cheb_acc(FB fb, double pk, T tk) {
  if (std::is_same<FB, T>::value) {
    if (FUSED && GCMF_FUSE_OK) return (FB)rfma((T)pk, tk, (T)fb);
    return fb + (FB)((T)pk * tk);
  }
  if (FUSED && GCMF_FUSE_OK) return (FB)rfma(pk, (double)tk, (double)fb);
  return fb + (FB)(pk * (double)tk);
}
// fbar = p_0 T_0 + p_1 T_1
template <bool FUSED, typename T, typename FB> __device__ __forceinline__ FB cheb_acc_first(double p0, double p1, T x, T a) {
  if (std::is_same<FB, T>::value) {
    if (FUSED && GCMF_FUSE_OK) return (FB)rfma((T)p1, a, (T)p0 * x);
    return (FB)((T)p0 * x + (T)p1 * a);
  }
  if (FUSED && GCMF_FUSE_OK) return (FB)rfma(p1, (double)a, p0 * (double)x);
  return (FB)(p0 * (double)x + p1 * (double)a);
}

}  // namespace gcmf
