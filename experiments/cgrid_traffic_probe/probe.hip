// Known-byte-count probe for the rocprofv3 FETCH_SIZE counter on the access shapes of the C-grid kernels.
//
//   probe <mode> [levels]      mode 0: LDS-direct loads, 16 bytes per lane (k_cgrid_ring, round 5)
//                              mode 1: plain 8-byte-per-lane loads          (k_cgrid_stream2c, rounds 2-4)
//
// `levels` x 6 planes of 2400 x 3584 f32.  A wave owns a 128-cell window of a 96-row strip; windows start at 128 wx - 8 cells (NOT on a
// 128-byte line, like the kernels' windows, whose halo shifts them) and do not overlap: every byte of every plane is requested exactly
// once, so   bytes read = levels * 6 * 2400 * 3584 * 4   is known and FETCH_SIZE / that = what the counter tallies for this shape.
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./probe <mode>   (tools/traffic_probe.sh does, and prints the ratio).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NX = 3584, NY = 2400, H = 96, NWX = NX / 128, NPL = 6;

__device__ __forceinline__ void dma16(const void *gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
}

template <int MODE> __global__ __launch_bounds__(256, 2) void k_probe(const float *planes, float *out, long long pstride) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int unit = blockIdx.x * 4 + wv;
  const int wx = unit % NWX, st = unit / NWX;
  if (st * H >= NY) return;
  const float *base = planes + (long long)blockIdx.y * NPL * pstride;
  float acc = 0.f;
  if (MODE == 0) {
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)s_raw) + wv * 2 * 3072;
    const int half = lane >> 5;
    int c4 = (wx * 128 - 8 + 4 * (lane & 31)) % NX;
    if (c4 < 0) c4 += NX;
    const char *q[3];
    for (int p = 0; p < 3; ++p) q[p] = reinterpret_cast<const char *>(base + (long long)(2 * p + half) * pstride) + c4 * 4;
    auto issue = [&](int r) {
      const unsigned ro = (unsigned)(min(r, NY - 1) * NX) * 4u;
      for (int p = 0; p < 3; ++p) dma16(q[p] + ro, lds0 + (r & 1) * 3072 + p * 1024);
    };
    issue(st * H);
    issue(st * H + 1);
    for (int r = st * H; r < st * H + H; ++r) {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      const float2 *s = reinterpret_cast<const float2 *>(s_raw + wv * 2 * 3072 + (r & 1) * 3072);
      for (int p = 0; p < 6; ++p) { const float2 v = s[p * 64 + lane]; acc += v.x + v.y; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(r + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    int col = (wx * 128 - 8 + 2 * lane) % NX;
    if (col < 0) col += NX;
    float2 nxt[6], nx2[6];
    auto ld = [&](float2(&d)[6], int r) {
      const long long ro = (long long)min(r, NY - 1) * NX + col;
      for (int p = 0; p < 6; ++p) d[p] = *reinterpret_cast<const float2 *>(base + p * pstride + ro);
    };
    ld(nxt, st * H);
    ld(nx2, st * H + 1);
    for (int r = st * H; r < st * H + H; r += 2) {
      float2 now[6];
      for (int p = 0; p < 6; ++p) now[p] = nxt[p];
      ld(nxt, r + 2);
      for (int p = 0; p < 6; ++p) acc += now[p].x + now[p].y;
      for (int p = 0; p < 6; ++p) now[p] = nx2[p];
      ld(nx2, r + 3);
      for (int p = 0; p < 6; ++p) acc += now[p].x + now[p].y;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, levels = argc > 2 ? atoi(argv[2]) : 8;
  const long long pstride = (long long)NX * NY;
  float *planes, *out;
  CHECK(hipMalloc(&planes, sizeof(float) * pstride * NPL * levels));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(planes, 0, sizeof(float) * pstride * NPL * levels));
  const int units = NWX * (NY / H);
  dim3 grid((units + 3) / 4, levels), block(256);
  const size_t lds = 4 * 2 * 3072;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, grid, block, lds, 0, planes, out, pstride);
    else hipLaunchKernelGGL(k_probe<1>, grid, block, 0, 0, planes, out, pstride);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)levels * NPL * pstride * 4;
    printf("mode %d (%s): %.1f MB requested once each, %.3f ms = %.2f TB/s\n", mode, mode == 0 ? "LDS-direct 16 B/lane" : "plain 8 B/lane", bytes / 1e6, ms,
           bytes / ms / 1e9);
  }
  printf("KNOWN_BYTES %lld\n", (long long)levels * NPL * pstride * 4);
  return 0;
}
