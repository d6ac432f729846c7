// How fast can 256 resident workgroups trade edge bands with their eight neighbours through (uncached) device memory?
// mode 0: plain data stores -> vmcnt(0) -> barrier -> per-tile epoch flag; readers poll the 8 flags, barrier, load (the protocol of
//         k_resident as shipped)
// mode 1: LL-style tagged items, no flags: a value travels as {lo, tag, hi, tag} (16 B, each 8-byte half carries its own tag), readers
//         poll the items themselves
// Volumes as in the 8-way slab of BASELINE config 3: 1120 band cells and 1248 halo cells per tile, two states each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NTY = 4, NTX = 64, NT = 512, BAND = 1120 * 2, HALO_PER_NB = 156 * 2;

__global__ __launch_bounds__(NT, 2) void k_probe(double *data, uint4 *items, unsigned *flags, int iters, int mode, unsigned ep0, double *sink,
                                                 int work) {
  const int bx = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);
  const int ty = bx / NTX, tx = bx % NTX, tid = threadIdx.x;
  double acc = 0.0;
  unsigned epoch = ep0;
  for (int it = 0; it < iters; ++it) {
    // stand-in for K levels of compute
    for (int q = 0; q < work; ++q) acc = acc * 1.0000001 + 1e-9;
    ++epoch;
    const int par = epoch & 1;
    if (mode == 0) {
      double *mine = data + ((size_t)par * 256 + bx) * BAND;
      for (int i = tid; i < BAND; i += NT) mine[i] = acc + i;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flags[bx], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid >= 1 && tid <= 8) {
        const int q = tid - 1 + (tid - 1 >= 4 ? 1 : 0);
        const int ny = (ty + q / 3 - 1 + NTY) % NTY, nx = (tx + q % 3 - 1 + NTX) % NTX;
        while ((int)(__hip_atomic_load(&flags[ny * NTX + nx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      for (int i = tid; i < 8 * HALO_PER_NB; i += NT) {
        const int nb = i / HALO_PER_NB, j = i % HALO_PER_NB, q = nb + (nb >= 4 ? 1 : 0);
        const int ny = (ty + q / 3 - 1 + NTY) % NTY, nx = (tx + q % 3 - 1 + NTX) % NTX;
        const double *theirs = data + ((size_t)par * 256 + ny * NTX + nx) * BAND;
        acc += __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)&theirs[j * 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * 1e-30;
      }
      __syncthreads();
    } else {
      uint4 *mine = items + ((size_t)par * 256 + bx) * BAND;
      for (int i = tid; i < BAND; i += NT) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(acc + i);
        mine[i] = make_uint4((unsigned)b, epoch, (unsigned)(b >> 32), epoch);
      }
      for (int i = tid; i < 8 * HALO_PER_NB; i += NT) {
        const int nb = i / HALO_PER_NB, j = i % HALO_PER_NB, q = nb + (nb >= 4 ? 1 : 0);
        const int ny = (ty + q / 3 - 1 + NTY) % NTY, nx = (tx + q % 3 - 1 + NTX) % NTX;
        const uint4 *theirs = items + ((size_t)par * 256 + ny * NTX + nx) * BAND;
        uint4 v;
        for (;;) {
          // 16-byte device-scope load (two 8-byte atomics would do as well: each half carries its own tag)
          const unsigned long long *p = (const unsigned long long *)&theirs[j * 3];
          const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
          if (v.y == epoch && v.w == epoch) break;
          __builtin_amdgcn_s_sleep(1);
        }
        acc += __longlong_as_double((long long)(((unsigned long long)v.z << 32) | v.x)) * 1e-30;
      }
      __syncthreads();
    }
  }
  if (acc == 123.456) sink[0] = acc;
}

int main(int argc, char **argv) {
  const int iters = 200;
  double *data, *sink;
  uint4 *items;
  unsigned *flags;
  CK(hipExtMallocWithFlags((void **)&data, (size_t)2 * 256 * BAND * 8, hipDeviceMallocUncached));
  CK(hipExtMallocWithFlags((void **)&items, (size_t)2 * 256 * BAND * 16, hipDeviceMallocUncached));
  CK(hipExtMallocWithFlags((void **)&flags, 1024 * 4, hipDeviceMallocUncached));
  CK(hipMalloc((void **)&sink, 8));
  CK(hipMemset(data, 0, (size_t)2 * 256 * BAND * 8));
  CK(hipMemset(items, 0, (size_t)2 * 256 * BAND * 16));
  CK(hipMemset(flags, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned ep = 0;
  for (int work : {0, 2000}) {
    for (int mode : {0, 1, 0, 1}) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(NT), 0, 0, data, items, flags, iters, mode, ep, sink, work);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        ep += iters;
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("work %d mode %d (%s): %.2f us per exchange\n", work, mode, mode ? "tagged items, no flags" : "flags", best * 1e3 / iters);
    }
  }
  return 0;
}
