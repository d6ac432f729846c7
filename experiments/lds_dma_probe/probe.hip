// Probe: global_load_lds_dwordx4 (gfx950 LDS-direct loads, 16 bytes per lane) issued from inline assembly
//   * does M0 address LDS beyond 64 KB?  (a workgroup of k_cgrid_ring needs ~73 KB)
//   * data layout in LDS (lane l writes 16 bytes at M0 + l * 16), unaligned global addresses (4-byte aligned only)
//   * completion is tracked by vmcnt
// build: hipcc --offload-arch=gfx950 -O2 -o probe probe.hip ; run: ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float *src, float *dst, unsigned lds_off, int shift) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  const int lane = threadIdx.x;
  const float *p = src + shift + lane * 4;   // 16 bytes per lane
  const unsigned m0v = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)(s_raw)) + lds_off;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(m0v) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float *l = reinterpret_cast<const float *>(s_raw + lds_off);
  for (int q = 0; q < 4; ++q) dst[q * 64 + lane] = l[q * 64 + lane];
}

int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *src, *dst;
  hipMalloc(&src, n * 4);
  hipMalloc(&dst, 256 * 4);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  const size_t lds = 150 * 1024;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int bad_total = 0;
  for (unsigned off : {0u, 1024u, 60u * 1024u, 70u * 1024u, 100u * 1024u, 140u * 1024u})
    for (int shift : {0, 1, 2, 4}) {
      hipMemset(dst, 0xff, 256 * 4);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, src, dst, off, shift);
      std::vector<float> o(256);
      hipError_t e = hipMemcpy(o.data(), dst, 256 * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < 256; ++i) bad += (o[i] != (float)(i + shift));
      printf("lds offset %6u  global shift %d floats: %s (%d wrong, first values %g %g %g %g) %s\n", off, shift, bad ? "MISMATCH" : "ok", bad, o[0], o[1], o[2], o[3],
             e == hipSuccess ? "" : hipGetErrorString(e));
      bad_total += bad;
    }
  printf(bad_total ? "SOME MISMATCHES\n" : "ALL OK\n");
  return 0;
}
