// Does a working set that fits the 256 MB memory-side cache (MALL / Infinity Cache) stream faster than one that does not?
// A 6-read : 2-write plane pass (the mix of a k_ringc launch) over a working set of W MB, repeated; GB/s by working-set size.
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct P { const double2 *r[6]; double2 *w[2]; long long n; };

__global__ __launch_bounds__(256) void k_pass(P p) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
    double2 a = p.r[0][i], b = p.r[1][i], c = p.r[2][i], d = p.r[3][i], e = p.r[4][i], f = p.r[5][i];
    double2 u, v;
    u.x = a.x + b.x * c.x + d.x; u.y = a.y + b.y * c.y + d.y;
    v.x = e.x - f.x * a.x;       v.y = e.y - f.y * a.y;
    p.w[0][i] = u;
    p.w[1][i] = v;
  }
}

int main() {
  const size_t maxb = 1200ull << 20;
  char *buf;
  CK(hipMalloc(&buf, maxb));
  CK(hipMemset(buf, 0, maxb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mb : {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768, 1100}) {
    const size_t plane = ((size_t)mb << 20) / 8 / 16 * 16;   // bytes per plane, 8 planes in the working set
    P p;
    for (int q = 0; q < 6; ++q) p.r[q] = (const double2 *)(buf + q * plane);
    for (int q = 0; q < 2; ++q) p.w[q] = (double2 *)(buf + (6 + q) * plane);
    p.n = plane / 16;
    const int reps = 30;
    for (int blocks : {2048, 8192}) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_pass, dim3(blocks), dim3(256), 0, 0, p);
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_pass, dim3(blocks), dim3(256), 0, 0, p);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("working set %4d MB, %5d blocks: %7.1f us per pass, %6.2f TB/s\n", mb, blocks, ms / reps * 1e3, 8.0 * plane * reps / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
