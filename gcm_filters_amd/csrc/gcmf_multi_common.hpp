// Helpers shared by the temporally blocked scalar kernels (gcmf_scalar_multi.hip, gcmf_scalar_skew.hip).
#pragma once
#include "gcmf_internal.hpp"
#include "gcmf_recurrence.hpp"

#include <cfloat>
#include <type_traits>

namespace gcmf {

template <typename T> struct MLim;
template <> struct MLim<float> { static __device__ __forceinline__ float big() { return FLT_MAX; } };
template <> struct MLim<double> { static __device__ __forceinline__ double big() { return DBL_MAX; } };

template <typename T> __device__ __forceinline__ T msan(T x) {  // numpy.nan_to_num
  if (x != x) return T(0);
  if (x > MLim<T>::big()) return MLim<T>::big();
  if (x < -MLim<T>::big()) return -MLim<T>::big();
  return x;
}

template <typename T, int VEC> struct alignas((sizeof(T) * VEC) > 16 ? 16 : (sizeof(T) * VEC)) MPack { T s[VEC]; };
template <typename T, int VEC> __device__ __forceinline__ void mload(T (&d)[VEC], const T *p) {
  const MPack<T, VEC> v = *reinterpret_cast<const MPack<T, VEC> *>(p);
#pragma unroll
  for (int k = 0; k < VEC; ++k) d[k] = v.s[k];
}
template <typename T, int VEC> __device__ __forceinline__ void mstore(T *p, const T (&d)[VEC]) {
  MPack<T, VEC> v;
#pragma unroll
  for (int k = 0; k < VEC; ++k) v.s[k] = d[k];
  *reinterpret_cast<MPack<T, VEC> *>(p) = v;
}

// lane i <- lane i-1 / lane i+1 of the same wave by DPP (one VALU move per dword; no LDS crossbar round trip).
// The outermost lanes keep their own value -- they are margin lanes whose results are never stored.
__device__ __forceinline__ int dpp_up_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }    // wave_shr:1
__device__ __forceinline__ int dpp_down_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }  // wave_shl:1
__device__ __forceinline__ double from_lower_lane(double v) {
  return __hiloint2double(dpp_up_i(__double2hiint(v)), dpp_up_i(__double2loint(v)));
}
__device__ __forceinline__ double from_upper_lane(double v) {
  return __hiloint2double(dpp_down_i(__double2hiint(v)), dpp_down_i(__double2loint(v)));
}
__device__ __forceinline__ float from_lower_lane(float v) { return __int_as_float(dpp_up_i(__float_as_int(v))); }
__device__ __forceinline__ float from_upper_lane(float v) { return __int_as_float(dpp_down_i(__float_as_int(v))); }

// nan_to_num that also reports what it removed (bit0: was NaN, bit1: was +-inf), so the raw value can be
// rebuilt later from the sanitised one without keeping a second copy in registers.  Written as selects
// (v_cmp + v_cndmask), not branches: this runs for every value of every level.
template <typename T> __device__ __forceinline__ T msan_flag(T x, unsigned &f) {
  const bool isn = (x != x);
  const bool big = (__builtin_fabs(x) > MLim<T>::big());  // false for NaN
  f = (isn ? 1u : 0u) | (big ? 2u : 0u);
  const T clamped = big ? __builtin_copysign(MLim<T>::big(), x) : x;
  return isn ? T(0) : clamped;
}
template <> __device__ __forceinline__ float msan_flag<float>(float x, unsigned &f) {
  const bool isn = (x != x);
  const bool big = (__builtin_fabsf(x) > FLT_MAX);
  f = (isn ? 1u : 0u) | (big ? 2u : 0u);
  const float clamped = big ? __builtin_copysignf(FLT_MAX, x) : x;
  return isn ? 0.f : clamped;
}
template <typename T> __device__ __forceinline__ T unsan(T g, unsigned f) {
  const T inf = __builtin_copysign((T)__builtin_inf(), g);
  const T r = (f & 2u) ? inf : g;
  return (f & 1u) ? (T)__builtin_nan("") : r;
}
template <> __device__ __forceinline__ float unsan<float>(float g, unsigned f) {
  const float inf = __builtin_copysignf(__builtin_inff(), g);
  const float r = (f & 2u) ? inf : g;
  return (f & 1u) ? __builtin_nanf("") : r;
}

// The same hops with zero fill: the outermost lane receives 0 instead of keeping its own value.  `old` is dead then, so the
// move needs no preceding copy of the source (one VALU instruction less per dword); margin lanes only.
__device__ __forceinline__ int dpp_up0_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_down0_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ double from_lower_lane0(double v) {
  return __hiloint2double(dpp_up0_i(__double2hiint(v)), dpp_up0_i(__double2loint(v)));
}
__device__ __forceinline__ double from_upper_lane0(double v) {
  return __hiloint2double(dpp_down0_i(__double2hiint(v)), dpp_down0_i(__double2loint(v)));
}
__device__ __forceinline__ float from_lower_lane0(float v) { return __int_as_float(dpp_up0_i(__float_as_int(v))); }
__device__ __forceinline__ float from_upper_lane0(float v) { return __int_as_float(dpp_down0_i(__float_as_int(v))); }

__device__ __forceinline__ double mabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float mabs(float x) { return __builtin_fabsf(x); }

constexpr int MAX_S = 8;
constexpr int MAX_PK = 9;   // coefficients a launch carries: k_ringc<double, K_FLUX> also runs nine levels per launch (round 5)

template <typename T, typename FB> struct MultiP {
  const T *u0;      // T_{k-1}
  const T *v0;      // T_{k-2}            (unused when first)
  T *uo;            // T_{k-1+S}          (unused when last)
  T *vo;            // T_{k-2+S}          (unused when last)
  const FB *fb_in;  // running sum in     (unused when first)
  FB *fb_out;       // running sum out / finalised result when last
  double *d_out;    // k_ringc, f32 state: the f64 result of the last launch (instead of fb_out) or NULL
  const T *cE, *cN, *ra;
  const T *zrow;    // nx zeros (k_ring: coefficient rows beyond a closed boundary)
  const uint8_t *mbits;
  unsigned *nfb;         // k_ring: counts the wave strips that met a NaN / inf and were redone by the general march (or NULL)
  const uint8_t *lbits;  // k_ring, first launch: bit 0 of a cell's byte = it exchanges with a neighbour (gcmf_plan::lbits) or NULL
  const T *area;
  int nx, rows, out_lo, out_hi;
  int H, nwx, nstrips, nwaves;
  int wrap, first, last, area_weighted;
  int xcd_per;       // k_ring: workgroups per XCD for the XCD-contiguous order (0 = launch order)
  int zigzag;        // k_ringc, flux kinds: odd strips march upwards (gcmf_ringc_impl.hpp)
  int npack;         // k_ringc / k_ringcs, batches (round 6): > 0 = the npack fields of the batch are ONE column of npack * (out_hi - out_lo) rows
                     // per window, cut into runs of H rows -- a wave walks its run, at most two (field, row range) segments (0: gridDim.y = batch)
  int fold_rows;     // k_ringcz on the plan that owns the tripole seam (round 6): the top fold_rows rows are strips that START at the seam, each
  int nfw;           // zipped with the strip of its MIRROR window (nfw such window pairs cover the two halves of a row); 0 = none
  long long bstride;
  double pk[MAX_PK];  // coefficient of level t (1-based) at pk[t-1]
  double p0;         // first only
  double c;
};


}  // namespace gcmf
