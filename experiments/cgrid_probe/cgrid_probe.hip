// What HBM rate can k_cgrid_stream2's ACCESS PATTERN reach?  50 levels of 2400x3600 f32 (u, v), f64 fbar, 14 shared f32 coefficient
// planes; a workgroup = 4 waves = 4 levels of one 128-cell window marching a strip of rows; per row each wave reads u, v, u', v'
// (8 B per lane), fu, fv (16 B per lane) and a quarter of the 14 coefficient rows, writes 4 state rows and 2 fbar rows.
// Variants: with / without the per-row barrier + LDS hand-over of the coefficient rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct P { const float *t[4]; const double *f[2]; float *to[4]; double *fo[2]; const float *c[14]; int nx, rows, H, nwx, ngroups, nlev4; long long lstride; };

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_probe4(const P p) {   // 4 cells per lane: 16-byte state accesses, two 16-byte fbar accesses
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = blockIdx.x, xcd = blk & 7, slot = blk >> 3;
  const int group = (slot / p.nlev4) * 8 + xcd, lev4 = slot % p.nlev4;
  if (group >= p.ngroups) return;
  int lev = lev4 * 4 + wv;
  if (lev >= 50) lev = 49;
  const int wx = group % p.nwx, st = group / p.nwx;
  const int a = st * p.H, b = min(a + p.H, p.rows);
  int col = wx * 244 + lane * 4;
  if (col + 3 >= p.nx) col = 0;
  const bool keep = lane >= 2 && lane < 62 && lev4 * 4 + wv < 50;
  const long long lo = (long long)lev * p.lstride;
  float acc = 0.f;
  for (int j = a; j < b; ++j) {
    const long long o = (long long)j * p.nx + col;
    float4 x[4]; double2 y[2][2]; float4 cc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = *(const float4 *)(p.t[q] + lo + o);
#pragma unroll
    for (int q = 0; q < 2; ++q) { y[q][0] = *(const double2 *)(p.f[q] + lo + o); y[q][1] = *(const double2 *)(p.f[q] + lo + o + 2); }
#pragma unroll
    for (int q = 0; q < 4; ++q) cc[q] = *(const float4 *)(p.c[(wv + 4 * q) % 14] + o);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += cc[q].x + cc[q].w;
    if (keep) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { float4 r = x[q]; r.x += acc; *(float4 *)(p.to[q] + lo + o) = r; }
#pragma unroll
      for (int q = 0; q < 2; ++q) { double2 r = y[q][0]; r.x += x[q].x; *(double2 *)(p.fo[q] + lo + o) = r; *(double2 *)(p.fo[q] + lo + o + 2) = y[q][1]; }
    }
  }
}

template <bool SYNC, int WPS>
__global__ __launch_bounds__(256, WPS) void k_probe(const P p) {
  __shared__ float2 s_c[2][14][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = blockIdx.x, xcd = blk & 7, slot = blk >> 3;
  const int group = (slot / p.nlev4) * 8 + xcd, lev4 = slot % p.nlev4;
  if (group >= p.ngroups) return;
  int lev = lev4 * 4 + wv;
  if (lev >= 50) lev = 49;
  const int wx = group % p.nwx, st = group / p.nwx;
  const int a = st * p.H, b = min(a + p.H, p.rows);
  int col = wx * 116 + lane * 2;
  if (col + 1 >= p.nx) col = 0;
  const bool keep = lane >= 3 && lane < 61 && lev4 * 4 + wv < 50;
  const long long lo = (long long)lev * p.lstride;
  float acc = 0.f;
  for (int j = a; j < b; ++j) {
    const long long o = (long long)j * p.nx + col;
    float2 x[4]; double2 y[2]; float2 cc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = *(const float2 *)(p.t[q] + lo + o);
#pragma unroll
    for (int q = 0; q < 2; ++q) y[q] = *(const double2 *)(p.f[q] + lo + o);
#pragma unroll
    for (int q = 0; q < 4; ++q) cc[q] = *(const float2 *)(p.c[(wv + 4 * q) % 14] + o);
    if (SYNC) {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (wv + 4 * q < 14) s_c[j & 1][wv + 4 * q][lane] = cc[q];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 14; ++q) { const float2 v = s_c[j & 1][q][lane]; acc += v.x + v.y; }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc += cc[q].x + cc[q].y;
    }
    if (keep) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { float2 r; r.x = x[q].x + acc; r.y = x[q].y; *(float2 *)(p.to[q] + lo + o) = r; }
#pragma unroll
      for (int q = 0; q < 2; ++q) { double2 r; r.x = y[q].x + x[q].x; r.y = y[q].y; *(double2 *)(p.fo[q] + lo + o) = r; }
    }
  }
}

template <bool SYNC, int WPS, int V4 = 0> int run(P p, const char *name, double bytes) {
  const long long cap = 1024 * WPS;  // waves resident
  p.nwx = V4 ? (p.nx + 243) / 244 : (p.nx + 115) / 116;
  p.nlev4 = 13;
  const long long per_strip = (long long)p.nwx * p.nlev4 * 4, hmax = 96;
  const long long ns_min = (p.rows + hmax - 1) / hmax;
  const long long rounds = (ns_min * per_strip + cap - 1) / cap;
  long long ns = rounds * cap / per_strip; if (ns < ns_min) ns = ns_min;
  p.H = (int)((p.rows + ns - 1) / ns);
  p.ngroups = p.nwx * ((p.rows + p.H - 1) / p.H);
  const long long gpx = (p.ngroups + 7) / 8;
  dim3 grid((unsigned)(gpx * p.nlev4 * 8)), block(256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (V4) hipLaunchKernelGGL((k_probe4<WPS>), grid, block, 0, 0, p); else hipLaunchKernelGGL((k_probe<SYNC, WPS>), grid, block, 0, 0, p);
  CK(hipDeviceSynchronize());
  const int L = 4;
  CK(hipEventRecord(e0));
  for (int l = 0; l < L; ++l) { if (V4) hipLaunchKernelGGL((k_probe4<WPS>), grid, block, 0, 0, p); else hipLaunchKernelGGL((k_probe<SYNC, WPS>), grid, block, 0, 0, p); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s H=%3d groups=%5d: %.2f ms per launch, %.2f TB/s of %.1f GB compulsory\n", name, p.H, p.ngroups, ms / L, bytes / (ms / L * 1e-3) / 1e12, bytes / 1e9);
  return 0;
}

int main() {
  const int nx = 3600, rows = 2400, nlev = 50;
  const size_t cells = (size_t)nx * rows;
  P p; p.nx = nx; p.rows = rows; p.lstride = (long long)cells;
  float *T[8]; double *F[4];
  for (int q = 0; q < 8; ++q) { CK(hipMalloc(&T[q], cells * nlev * 4)); CK(hipMemset(T[q], 0, cells * nlev * 4)); }
  for (int q = 0; q < 4; ++q) { CK(hipMalloc(&F[q], cells * nlev * 8)); CK(hipMemset(F[q], 0, cells * nlev * 8)); }
  for (int q = 0; q < 14; ++q) { float *c; CK(hipMalloc(&c, cells * 4)); CK(hipMemset(c, 0, cells * 4)); p.c[q] = c; }
  for (int q = 0; q < 4; ++q) { p.t[q] = T[q]; p.to[q] = T[4 + q]; }
  for (int q = 0; q < 2; ++q) { p.f[q] = F[q]; p.fo[q] = F[2 + q]; }
  const double bytes = (double)cells * nlev * (4 * 4 + 2 * 8 + 4 * 4 + 2 * 8) + (double)cells * 13 * 14 * 4;
  for (int rep = 0; rep < 2; ++rep) {
    if (run<false, 2>(p, "no barrier, 2 waves per SIMD", bytes)) return 1;
    if (run<true, 2>(p, "barrier + LDS hand-over, 2 waves per SIMD", bytes)) return 1;
    if (run<false, 1>(p, "no barrier, 1 wave per SIMD", bytes)) return 1;
    if (run<true, 1>(p, "barrier + LDS hand-over, 1 wave per SIMD", bytes)) return 1;
    if (run<false, 2, 1>(p, "4 cells per lane, 2 waves per SIMD", bytes)) return 1;
    if (run<false, 1, 1>(p, "4 cells per lane, 1 wave per SIMD", bytes)) return 1;
  }
  return 0;
}
