// Shared pieces of the static-ring C-grid kernels (gcmf_cgrid_ring.hip: backward / Clenshaw evaluation; gcmf_cgrid_ringf.hip: the
// reference's forward recurrence): packed pairs, ring arithmetic, the LDS-direct load, the scalar row cursor.
#pragma once
#include "gcmf_multi_common.hpp"
#include "gcmf_recurrence.hpp"
#include <atomic>
#include <cstdlib>

namespace gcmf {

template <typename T> struct CgV2;
template <> struct CgV2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct CgV2<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <int N> using cic = std::integral_constant<int, N>;
constexpr int cmod(int a, int m) { return ((a % m) + m) % m; }

template <typename T> __device__ __forceinline__ T cr_san(T x) {  // numpy.nan_to_num, as c2san of gcmf_cgrid_stream2.hip
  const bool isn = (x != x);
  const bool big = (mabs(x) > MLim<T>::big());
  const T clamped = big ? (x > T(0) ? MLim<T>::big() : -MLim<T>::big()) : x;
  return isn ? T(0) : clamped;
}

constexpr int CR_U = 12;  // unroll factor of the row loop = common period of all rings

constexpr int CR_WPB = 4;   // waves (= levels of the batch) per workgroup

#pragma clang diagnostic ignored "-Winline-asm"   // (M0 on the clobber list: the compiler has no use of its own for it in these kernels)
// one LDS-direct load: every lane fetches 16 bytes from its global address, lane l's land at (LDS address in M0) + 16 l
__device__ __forceinline__ void cr_dma16(const void *gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void cr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The row cursor of a march (scalar registers): issue after issue it walks the rows r_begin, r_begin + 1, ... of the strip (periodic or
// clamped at the slab's edges), stops at the last delivered row (the padded iterations of the last ring period re-load it), and gives
// the byte offsets of the row it is on (ro) and of the row before it (rc) inside a level's plane (< 4 GB).
struct CRingCursor {
  int nx, rows, r_end, ri, cj;
  bool wrap;
  unsigned ro, rc, es;
  __device__ __forceinline__ CRingCursor(int nx_, int rows_, bool wrap_, int r_begin, int r_end_, unsigned es_)
      : nx(nx_), rows(rows_), r_end(r_end_), ri(r_begin), wrap(wrap_), es(es_) {
    int r = r_begin - 1;
    if (wrap) {
      r = r < 0 ? r + rows : (r >= rows ? r - rows : r);
      r = r < 0 ? r + rows : (r >= rows ? r - rows : r);  // |overshoot| <= S + 1 may exceed one period on tiny grids
    } else {
      r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    }
    cj = r;
    ro = rc = (unsigned)(cj * nx) * es;
  }
  __device__ __forceinline__ void advance() {
    const bool adv = ri < r_end;
    int nj;
    if (wrap) {
      nj = cj + 1;
      nj = nj >= rows ? nj - rows : nj;
    } else {
      nj = ri < 0 ? 0 : (ri >= rows ? rows - 1 : ri);
    }
    rc = adv ? ro : rc;
    cj = adv ? nj : cj;
    ro = (unsigned)(cj * nx) * es;
    ri += adv ? 1 : 0;
  }
};

}  // namespace gcmf
