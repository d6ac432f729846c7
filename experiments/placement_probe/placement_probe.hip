// How do the relative addresses of the 9 planes a k_ring-like launch streams change its speed?
// Planes are carved from one arena at base + k * (plane bytes rounded up to `round`) + k * pad.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct P { const double *c0, *c1, *c2, *u, *v, *f; double *uo, *vo, *fo; int nx, rows, H, nwx, nwaves; };
__global__ __launch_bounds__(256, 1) void k_march(const P p) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= p.nwaves) return;
  const int wx = wid % p.nwx, st = wid / p.nwx;
  const int a = st * p.H, b = min(a + p.H, p.rows);
  const int col = wx * 112 + lane * 2;
  if (col + 1 >= p.nx) return;
  for (int j = a; j < b; ++j) {
    const long long o = (long long)j * p.nx + col;
    const double2 x0 = *(const double2 *)(p.c0 + o), x1 = *(const double2 *)(p.c1 + o), x2 = *(const double2 *)(p.c2 + o);
    const double2 y0 = *(const double2 *)(p.u + o), y1 = *(const double2 *)(p.v + o), y2 = *(const double2 *)(p.f + o);
    double2 r0, r1, r2;
    r0.x = x0.x * y0.x + y1.x; r0.y = x0.y * y0.y + y1.y;
    r1.x = x1.x * y1.x + y2.x; r1.y = x1.y * y1.y + y2.y;
    r2.x = x2.x * y2.x + y0.x; r2.y = x2.y * y2.y + y0.y;
    if (lane >= 4 && lane < 60) { *(double2 *)(p.uo + o) = r0; *(double2 *)(p.vo + o) = r1; *(double2 *)(p.fo + o) = r2; }
  }
}
int main(int argc, char **argv) {
  const int nx = 3600, rows = 2400, H = 80;
  const size_t bytes = (size_t)nx * rows * sizeof(double);
  const size_t arena_bytes = 9 * (bytes + (64u << 20)) + (64u << 20);
  char *arena; CK(hipMalloc(&arena, arena_bytes)); CK(hipMemset(arena, 0, arena_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("arena %p, plane %zu bytes\n", arena, bytes);
  const size_t rounds[] = {256, 4096, 1u << 21};
  const size_t pads[] = {0, 256, 1024, 4096, 4352, 16384, 65536, 69632, 262144, 1u << 20, (1u << 20) + 4096, 1u << 21, 3u << 20, 5u << 20};
  for (size_t round : rounds) for (size_t pad : pads) {
    const size_t stride = (bytes + round - 1) / round * round + pad;
    double *pl[9];
    for (int k = 0; k < 9; ++k) pl[k] = (double *)(arena + k * stride);
    P p; p.c0 = pl[0]; p.c1 = pl[1]; p.c2 = pl[2]; p.u = pl[3]; p.v = pl[4]; p.f = pl[5]; p.uo = pl[6]; p.vo = pl[7]; p.fo = pl[8];
    p.nx = nx; p.rows = rows; p.H = H; p.nwx = (nx + 111) / 112; p.nwaves = p.nwx * (rows / H);
    const int L = 30;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_march, dim3((p.nwaves + 3) / 4), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_march, dim3((p.nwaves + 3) / 4), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("round %8zu pad %8zu (stride %% 2MB = %7zu): %.1f us per launch\n", round, pad, stride % (1u << 21), 1e3 * ms / L);
  }
  return 0;
}
