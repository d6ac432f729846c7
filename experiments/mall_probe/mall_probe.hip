// Does a launch that starts where the previous one ended find its operands in L2 / MALL?
// 9 planes of rows x nx doubles (6 read, 3 written per launch; the 3 written are 3 of the next launch's reads), waves march
// through strips of rows like k_ring does.  Compare: every launch marches north  vs  launches alternate north / south.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct P { const double *c0, *c1, *c2, *u, *v, *f; double *uo, *vo, *fo; int nx, rows, H, nwx, nwaves, dir; };

__global__ __launch_bounds__(256, 1) void k_march(const P p) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= p.nwaves) return;
  const int wx = wid % p.nwx, st = wid / p.nwx;
  const int a = st * p.H, b = min(a + p.H, p.rows);
  const int col = wx * 128 + lane * 2;
  for (int i = 0; i < b - a; ++i) {
    const int j = p.dir > 0 ? a + i : b - 1 - i;
    const long long o = (long long)j * p.nx + col;
    const double2 x0 = *(const double2 *)(p.c0 + o), x1 = *(const double2 *)(p.c1 + o), x2 = *(const double2 *)(p.c2 + o);
    const double2 y0 = *(const double2 *)(p.u + o), y1 = *(const double2 *)(p.v + o), y2 = *(const double2 *)(p.f + o);
    double2 r0, r1, r2;
    r0.x = x0.x * y0.x + y1.x; r0.y = x0.y * y0.y + y1.y;
    r1.x = x1.x * y1.x + y2.x; r1.y = x1.y * y1.y + y2.y;
    r2.x = x2.x * y2.x + y0.x; r2.y = x2.y * y2.y + y0.y;
    *(double2 *)(p.uo + o) = r0; *(double2 *)(p.vo + o) = r1; *(double2 *)(p.fo + o) = r2;
  }
}

int main() {
  const int nx = 4096, rows = 2480, H = 80;
  const size_t n = (size_t)nx * rows, bytes = n * sizeof(double);
  double *c[3], *s[2][3];
  for (int k = 0; k < 3; ++k) { CK(hipMalloc(&c[k], bytes)); CK(hipMemset(c[k], 0, bytes)); }
  for (int q = 0; q < 2; ++q) for (int k = 0; k < 3; ++k) { CK(hipMalloc(&s[q][k], bytes)); CK(hipMemset(s[q][k], 0, bytes)); }
  P p; p.c0 = c[0]; p.c1 = c[1]; p.c2 = c[2]; p.nx = nx; p.rows = rows; p.H = H; p.nwx = nx / 128; p.nwaves = p.nwx * (rows / H);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("planes %.1f MB each, %d waves, traffic per launch %.1f MB\n", bytes / 1e6, p.nwaves, 9 * bytes / 1e6);
  for (int rep = 0; rep < 3; ++rep) for (int mode = 0; mode < 2; ++mode) {
    const int L = 40;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int l = 0; l < L; ++l) {
      const int q = l & 1;
      p.u = s[q][0]; p.v = s[q][1]; p.f = s[q][2]; p.uo = s[q ^ 1][0]; p.vo = s[q ^ 1][1]; p.fo = s[q ^ 1][2];
      p.dir = (mode == 1 && (l & 1)) ? -1 : 1;
      hipLaunchKernelGGL(k_march, dim3((p.nwaves + 3) / 4), dim3(256), 0, 0, p);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %.1f us per launch, %.2f TB/s\n", mode ? "alternating north/south" : "always north          ", 1e3 * ms / L, 9.0 * bytes / (ms / L * 1e-3) / 1e12);
  }
  return 0;
}
