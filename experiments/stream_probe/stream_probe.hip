// What HBM rate can k_ring's ACCESS PATTERN reach at all?  9 planes (6 read, 3 written), waves march 128-column windows through
// strips of rows, no arithmetic.  Variants: rows in flight per wave (U), waves resident per SIMD (1: 1024 strips-waves, 2: 2048
// with half-height strips), XCD-contiguous workgroup order, non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct P { const double *r[6]; double *w[3]; int nx, rows, H, nwx, nwaves, xcd_per; };

// the same bytes, but every wave's rows are CONTIGUOUS in every plane (a tiled layout: [strip][window][row][128 cells])
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_tiled(const P p) {
  const int lane = threadIdx.x & 63;
  int bx = blockIdx.x;
  if (p.xcd_per > 0 && bx < 8 * p.xcd_per) bx = (bx & 7) * p.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= p.nwaves) return;
  const long long base = (long long)wid * p.H * 112;   // 112 stored cells per row and window
  const bool keep = lane < 56;
  for (int j = 0; j < p.H; ++j) {
    const long long o = base + (long long)j * 112 + lane * 2;
    double2 x[6];
    if (keep) {
#pragma unroll
      for (int q = 0; q < 6; ++q) x[q] = *(const double2 *)(p.r[q] + o);
      double2 s0, s1, s2;
      s0.x = x[0].x + x[3].x; s0.y = x[0].y + x[3].y;
      s1.x = x[1].x + x[4].x; s1.y = x[1].y + x[4].y;
      s2.x = x[2].x + x[5].x; s2.y = x[2].y + x[5].y;
      *(double2 *)(p.w[0] + o) = s0; *(double2 *)(p.w[1] + o) = s1; *(double2 *)(p.w[2] + o) = s2;
    }
  }
}

template <int U, int WPS, bool NT>
__global__ __launch_bounds__(256, WPS) void k_stream(const P p) {
  const int lane = threadIdx.x & 63;
  int bx = blockIdx.x;
  if (p.xcd_per > 0 && bx < 8 * p.xcd_per) bx = (bx & 7) * p.xcd_per + (bx >> 3);
  const int wid = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= p.nwaves) return;
  const int wx = wid % p.nwx, st = wid / p.nwx;
  const int a = st * p.H, b = min(a + p.H, p.rows);
  const int col = wx * 112 + lane * 2;
  if (col + 1 >= p.nx) return;
  const bool keep = lane >= 4 && lane < 60;
  for (int j = a; j < b; j += U) {
    double2 x[U][6];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long o = (long long)min(j + u, b - 1) * p.nx + col;
#pragma unroll
      for (int q = 0; q < 6; ++q) x[u][q] = *(const double2 *)(p.r[q] + o);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j + u >= b) break;
      const long long o = (long long)(j + u) * p.nx + col;
      double2 s0, s1, s2;
      s0.x = x[u][0].x + x[u][3].x; s0.y = x[u][0].y + x[u][3].y;
      s1.x = x[u][1].x + x[u][4].x; s1.y = x[u][1].y + x[u][4].y;
      s2.x = x[u][2].x + x[u][5].x; s2.y = x[u][2].y + x[u][5].y;
      if (keep) {
        if (NT) {
          __builtin_nontemporal_store(s0.x, p.w[0] + o); __builtin_nontemporal_store(s0.y, p.w[0] + o + 1);
          __builtin_nontemporal_store(s1.x, p.w[1] + o); __builtin_nontemporal_store(s1.y, p.w[1] + o + 1);
          __builtin_nontemporal_store(s2.x, p.w[2] + o); __builtin_nontemporal_store(s2.y, p.w[2] + o + 1);
        } else {
          *(double2 *)(p.w[0] + o) = s0; *(double2 *)(p.w[1] + o) = s1; *(double2 *)(p.w[2] + o) = s2;
        }
      }
    }
  }
}

template <int U, int WPS, bool NT> int run(P p, int xcd, const char *name, size_t bytes) {
  const int slots = 1024 * WPS;
  p.nwx = (p.nx + 111) / 112;
  const int want = slots / p.nwx;
  p.H = (p.rows + want - 1) / want;
  const int nstrips = (p.rows + p.H - 1) / p.H;
  p.nwaves = p.nwx * nstrips;
  const int nblk = (p.nwaves + 3) / 4;
  p.xcd_per = xcd ? nblk / 8 : 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int L = 30;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stream<U, WPS, NT>), dim3(nblk), dim3(256), 0, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int l = 0; l < L; ++l) hipLaunchKernelGGL((k_stream<U, WPS, NT>), dim3(nblk), dim3(256), 0, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s U=%d waves/SIMD=%d xcd=%d H=%3d waves=%4d: %6.1f us, %.2f TB/s (9 planes)\n", name, U, WPS, xcd, p.H, p.nwaves, 1e3 * ms / L,
         9.0 * bytes / (ms / L * 1e-3) / 1e12);
  return 0;
}

template <int WPS> int run_tiled(P p, int xcd, size_t bytes) {
  const int slots = 1024 * WPS;
  p.nwx = (p.nx + 111) / 112;
  const int want = slots / p.nwx;
  p.H = (p.rows + want - 1) / want;
  p.nwaves = p.nwx * ((p.rows + p.H - 1) / p.H);
  while ((long long)p.nwaves * p.H * 112 > (long long)p.nx * p.rows) --p.nwaves;   // stay inside the planes
  const int nblk = (p.nwaves + 3) / 4;
  p.xcd_per = xcd ? nblk / 8 : 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int L = 30;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_tiled<WPS>), dim3(nblk), dim3(256), 0, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int l = 0; l < L; ++l) hipLaunchKernelGGL((k_tiled<WPS>), dim3(nblk), dim3(256), 0, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double moved = 9.0 * (double)p.nwaves * p.H * 112 * 8;
  printf("%-34s waves/SIMD=%d xcd=%d H=%3d waves=%4d: %6.1f us, %.2f TB/s (%.0f MB)\n", "tiled: a wave's rows contiguous", WPS, xcd, p.H, p.nwaves,
         1e3 * ms / L, moved / (ms / L * 1e-3) / 1e12, moved / 1e6);
  return 0;
}

int main() {
  const int nx = 3600, rows = 2400;
  const size_t bytes = (size_t)nx * rows * sizeof(double);
  P p; p.nx = nx; p.rows = rows;
  for (int q = 0; q < 6; ++q) { CK(hipMalloc((void **)&p.r[q], bytes)); CK(hipMemset((void *)p.r[q], 0, bytes)); }
  for (int q = 0; q < 3; ++q) { CK(hipMalloc((void **)&p.w[q], bytes)); CK(hipMemset(p.w[q], 0, bytes)); }
  for (int rep = 0; rep < 2; ++rep) {
    for (int xcd = 0; xcd < 2; ++xcd) {
      if (run<1, 1, false>(p, xcd, "1 row in flight", bytes)) return 1;
      if (run<2, 1, false>(p, xcd, "2 rows in flight", bytes)) return 1;
      if (run<3, 1, false>(p, xcd, "3 rows in flight", bytes)) return 1;
      if (run<4, 1, false>(p, xcd, "4 rows in flight", bytes)) return 1;
      if (run<6, 1, false>(p, xcd, "6 rows in flight", bytes)) return 1;
      if (run<2, 2, false>(p, xcd, "2 waves/SIMD, 2 rows", bytes)) return 1;
      if (run<4, 2, false>(p, xcd, "2 waves/SIMD, 4 rows", bytes)) return 1;
      if (run<3, 1, true>(p, xcd, "3 rows, non-temporal stores", bytes)) return 1;
      if (run_tiled<1>(p, xcd, bytes)) return 1;
      if (run_tiled<2>(p, xcd, bytes)) return 1;
    }
  }
  return 0;
}
