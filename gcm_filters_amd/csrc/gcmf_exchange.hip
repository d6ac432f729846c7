// Halo exchange of the row-slab (multi-GPU) driver, issued from C++: RCCL send / recv pairs over xGMI on a side
// stream, overlapped with the interior launch of the compute stream.
//
// The reference has no spatial decomposition (gcm_filters/filter.py:478-486 parallelises over non-core dims only);
// this is the exchange step SURVEY 8e describes for the j-slab split the north star asks for.  One exchange
// refreshes `halo` ghost rows per slab edge of every state array (T_{k-1}, T_{k-2}); it is needed once per `halo`
// recurrence steps (s-step halos, gcm_filters_amd/distributed.py).
//
// Layout: a state array is (nblocks, rows_alloc, nx) with nblocks = ncomp * nbatch; the `halo` edge rows of one block
// are contiguous, so with few blocks the rows are sent straight out of / received straight into the state arrays
// (no packing); with many blocks (batched levels) they are packed into one message per neighbour by a copy kernel.
//
// RCCL is bound lazily with dlopen (the library that torch already loaded is reused; libgcmf itself has no link-time
// dependency on it and stays loadable on a box without RCCL -- gcmf_comm_create then fails with a message).
#include "gcmf_internal.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>

namespace gcmf {

struct Id128 { char b[128]; };  // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)

// the few RCCL entry points used, by signature (rccl.h: ncclResult_t = int, ncclDataType_t ncclInt8 = 0)
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  // optional (gcmf_comm_info): what RCCL itself says about the communicator
  int (*GetVersion)(int *) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  int (*CommUserRank)(void *, int *) = nullptr;
};
static std::mutex g_rccl_mu;
static Rccl g_rccl;

static bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void *h = nullptr;
  for (const char *n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    set_error("RCCL not found (dlopen librccl.so.1: %s)", dlerror());
    return false;
  }
  Rccl r;
  r.h = h;
#define GCMF_SYM(field, name)                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));           \
  if (!r.field) {                                                          \
    set_error("RCCL symbol %s missing", name);                             \
    return false;                                                          \
  }
  GCMF_SYM(GetUniqueId, "ncclGetUniqueId")
  GCMF_SYM(CommInitRank, "ncclCommInitRank")
  GCMF_SYM(CommDestroy, "ncclCommDestroy")
  GCMF_SYM(GroupStart, "ncclGroupStart")
  GCMF_SYM(GroupEnd, "ncclGroupEnd")
  GCMF_SYM(Send, "ncclSend")
  GCMF_SYM(Recv, "ncclRecv")
  GCMF_SYM(GetErrorString, "ncclGetErrorString")
#undef GCMF_SYM
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
  r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  g_rccl = r;
  return true;
}

#define GCMF_NCCL(call)                                                                         \
  do {                                                                                          \
    int r_ = (call);                                                                            \
    if (r_ != 0) {                                                                              \
      set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return GCMF_ERR_HIP;                                                                      \
    }                                                                                           \
  } while (0)

// Inside ncclGroupStart .. ncclGroupEnd: remember the first failure, keep going to GroupEnd (an early return would leave the
// group open and the comm unusable), then GCMF_NCCL_GROUP_END reports it and clears the exchange state.
#define GCMF_NCCL_IN_GROUP(call)                                                                    \
  do {                                                                                              \
    int r_ = grp_rc ? 0 : (call);                                                                   \
    if (r_ != 0) {                                                                                  \
      grp_rc = r_;                                                                                  \
      set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__);     \
    }                                                                                               \
  } while (0)
#define GCMF_NCCL_GROUP_END()                                                                       \
  do {                                                                                              \
    int e_ = g_rccl.GroupEnd();                                                                     \
    if (!grp_rc && e_ != 0) {                                                                       \
      grp_rc = e_;                                                                                  \
      set_error("ncclGroupEnd failed: %s (%s:%d)", g_rccl.GetErrorString(e_), __FILE__, __LINE__);  \
    }                                                                                               \
    if (grp_rc) {                                                                                   \
      c->states.clear();                                                                            \
      c->pending_unpack = false;                                                                    \
      return GCMF_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

// rows [r0, r0 + nrows) of every block of `src` (nblocks, rows_alloc, nx) <-> a packed (nblocks, nrows, nx) buffer;
// 16 bytes per lane, row_bytes is a multiple of 16 (checked by the caller)
__global__ void k_pack_rows(const uint4 *__restrict__ state, uint4 *__restrict__ packed, long long block_q, int row_q, int r0,
                            int nrows, long long nblocks, int unpack) {
  const long long per = (long long)nrows * row_q;
  const long long n = per * nblocks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / per, rem = i - b * per;
    const long long s = b * block_q + (long long)r0 * row_q + rem;
    if (unpack) const_cast<uint4 *>(state)[s] = packed[i];
    else packed[i] = state[s];
  }
}

}  // namespace gcmf

using namespace gcmf;

struct gcmf_comm {
  void *comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t side = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  void *pack = nullptr;  // [send north | send south | recv south | recv north], each pack_part bytes
  size_t pack_part = 0;
  bool pending_unpack = false;
  // what gcmf_halo_finish has to unpack
  std::vector<void *> states;
  long long nblocks = 0;
  int rows_alloc = 0, nx = 0, first_owned = 0, rows_owned = 0, halo = 0, south = -1, north = -1;
  size_t esize = 0;
  std::mutex mu;
};

extern "C" {

int gcmf_comm_unique_id(void *id128) {
  if (!id128) return GCMF_ERR_INVALID_ARG;
  if (!rccl_load()) return GCMF_ERR_UNSUPPORTED;
  GCMF_NCCL(g_rccl.GetUniqueId(id128));
  return GCMF_OK;
}

void gcmf_comm_destroy(gcmf_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->side) { (void)hipStreamSynchronize(c->side); }
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->pack) (void)hipFree(c->pack);
  delete c;
}

// What RCCL reports about the communicator libgcmf's exchanges run on: library version (ncclGetVersion), ranks (ncclCommCount) and this
// rank (ncclCommUserRank); -1 where the entry point is missing.  bench.py --gpus N prints it with the N > 1 line.
int gcmf_comm_info(gcmf_comm *c, int *version, int *nranks, int *rank) {
  if (!c || !version || !nranks || !rank) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  *version = *nranks = *rank = -1;
  if (g_rccl.GetVersion) (void)g_rccl.GetVersion(version);
  if (g_rccl.CommCount) (void)g_rccl.CommCount(c->comm, nranks);
  if (g_rccl.CommUserRank) (void)g_rccl.CommUserRank(c->comm, rank);
  return GCMF_OK;
}

int gcmf_comm_create(const void *id128, int world, int rank, int device, gcmf_comm **out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) {
    set_error("gcmf_comm_create: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (!rccl_load()) return GCMF_ERR_UNSUPPORTED;
  GCMF_HIP(hipSetDevice(device));
  gcmf_comm *c = new gcmf_comm();
  c->world = world;
  c->rank = rank;
  c->device = device;
  Id128 id;
  memcpy(id.b, id128, sizeof id.b);
  int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != 0) {
    set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    c->comm = nullptr;
    gcmf_comm_destroy(c);
    return GCMF_ERR_HIP;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("gcmf_comm_create: %s", hipGetErrorString(e));
    gcmf_comm_destroy(c);
    return GCMF_ERR_HIP;
  }
  *out = c;
  return GCMF_OK;
}

// Post the exchange: everything enqueued on `stream` so far is waited for by the side stream, which then moves
// rows [first_owned + rows_owned - halo, first_owned + rows_owned) of every block of every state to `north`'s southern ghost
// rows and rows [first_owned, first_owned + halo) to `south`'s northern ghost rows, and receives this slab's own ghost rows.
// south / north: peer ranks, -1 where the slab edge is a physical boundary.  Returns at once; work enqueued on `stream`
// after this call (the interior rows) runs concurrently with the transfers.  The state arrays must not be written until
// gcmf_halo_finish, except rows outside the 2 x halo sent and 2 x halo received ones.
int gcmf_halo_start(gcmf_comm *c, void *const *states, int nstate, int64_t nblocks, int64_t rows_alloc, int64_t nx,
                    int64_t first_owned, int64_t rows_owned, int halo, int dtype, int south, int north, void *stream) {
  if (!c || !states || nstate < 1 || nstate > 4 || nblocks < 1 || halo < 1 || rows_owned < halo ||
      first_owned + rows_owned > rows_alloc || south >= c->world || north >= c->world) {
    set_error("gcmf_halo_start: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  if ((south >= 0 && first_owned < halo) || (north >= 0 && rows_alloc - first_owned - rows_owned < halo)) {
    set_error("gcmf_halo_start: the slab has fewer ghost rows than the halo");
    return GCMF_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->pending_unpack || !c->states.empty()) {
    set_error("gcmf_halo_start: the previous exchange was not finished");
    return GCMF_ERR_INVALID_ARG;
  }
  GCMF_HIP(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t es = dtype_size(dtype);
  const size_t row_bytes = (size_t)nx * es, edge = (size_t)halo * row_bytes, block_bytes = (size_t)rows_alloc * row_bytes;
  const int r_top = (int)(first_owned + rows_owned - halo), r_bot = (int)first_owned;
  const int g_s = (int)(first_owned - halo), g_n = (int)(first_owned + rows_owned);
  GCMF_HIP(hipEventRecord(c->ev_ready, s));
  GCMF_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
  const bool packed = (nblocks * nstate > 8) && (row_bytes % 16 == 0);
  c->states.assign(states, states + nstate);
  c->nblocks = nblocks; c->rows_alloc = (int)rows_alloc; c->nx = (int)nx; c->first_owned = (int)first_owned;
  c->rows_owned = (int)rows_owned; c->halo = halo; c->south = south; c->north = north; c->esize = es;
  int grp_rc = 0;
  struct Abandon {   // any failing return below leaves no half-posted exchange behind (the next gcmf_halo_start would refuse)
    gcmf_comm *c;
    bool posted = false;
    ~Abandon() { if (!posted) { c->states.clear(); c->pending_unpack = false; } }
  } abandon{c};
  if (!packed) {
    GCMF_NCCL(g_rccl.GroupStart());
    // per peer the order of sends is [northward, southward] and of receives [into south ghosts, into north ghosts]:
    // with two ranks (or one) both neighbours are the same peer and messages are matched in posting order
    for (int q = 0; q < nstate; ++q)
      for (int64_t b = 0; b < nblocks; ++b) {
        char *base = (char *)states[q] + (size_t)b * block_bytes;
        if (north >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Send(base + (size_t)r_top * row_bytes, edge, 0, north, c->comm, c->side));
        if (south >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Send(base + (size_t)r_bot * row_bytes, edge, 0, south, c->comm, c->side));
        if (south >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Recv(base + (size_t)g_s * row_bytes, edge, 0, south, c->comm, c->side));
        if (north >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Recv(base + (size_t)g_n * row_bytes, edge, 0, north, c->comm, c->side));
      }
    GCMF_NCCL_GROUP_END();
    c->pending_unpack = false;
  } else {
    const size_t part = (size_t)nstate * nblocks * edge;
    if (c->pack_part < part) {
      if (c->pack) {
        GCMF_HIP(hipStreamSynchronize(c->side));
        GCMF_HIP(hipFree(c->pack));
        c->pack = nullptr;
        c->pack_part = 0;
      }
      GCMF_HIP(hipMalloc(&c->pack, 4 * part));
      c->pack_part = part;
    }
    char *sn = (char *)c->pack, *ss = sn + c->pack_part, *rs = ss + c->pack_part, *rn = rs + c->pack_part;
    const int row_q = (int)(row_bytes / 16);
    const long long block_q = (long long)(block_bytes / 16);
    const size_t per_state = (size_t)nblocks * edge;
    const long long n16 = (long long)(per_state / 16);
    const int grid = (int)std::min<long long>((n16 + 255) / 256, 2048);
    for (int q = 0; q < nstate; ++q) {
      if (north >= 0)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, c->side, (const uint4 *)states[q], (uint4 *)(sn + q * per_state),
                           block_q, row_q, r_top, halo, (long long)nblocks, 0);
      if (south >= 0)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, c->side, (const uint4 *)states[q], (uint4 *)(ss + q * per_state),
                           block_q, row_q, r_bot, halo, (long long)nblocks, 0);
    }
    GCMF_HIP(hipGetLastError());
    GCMF_NCCL(g_rccl.GroupStart());
    if (north >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Send(sn, part, 0, north, c->comm, c->side));
    if (south >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Send(ss, part, 0, south, c->comm, c->side));
    if (south >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Recv(rs, part, 0, south, c->comm, c->side));
    if (north >= 0) GCMF_NCCL_IN_GROUP(g_rccl.Recv(rn, part, 0, north, c->comm, c->side));
    GCMF_NCCL_GROUP_END();
    c->pending_unpack = true;
  }
  (void)g_s; (void)g_n;
  abandon.posted = true;
  return GCMF_OK;
}

// Make `stream` wait for the exchange posted by gcmf_halo_start (and unpack the ghost rows if they travelled packed).
int gcmf_halo_finish(gcmf_comm *c, void *stream) {
  if (!c) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->states.empty()) {
    set_error("gcmf_halo_finish: no exchange in flight");
    return GCMF_ERR_INVALID_ARG;
  }
  GCMF_HIP(hipSetDevice(c->device));
  if (c->pending_unpack) {
    const size_t row_bytes = (size_t)c->nx * c->esize, edge = (size_t)c->halo * row_bytes;
    const size_t per_state = (size_t)c->nblocks * edge;
    char *rs = (char *)c->pack + 2 * c->pack_part, *rn = rs + c->pack_part;
    const int row_q = (int)(row_bytes / 16);
    const long long block_q = (long long)((size_t)c->rows_alloc * row_bytes / 16);
    const long long n16 = (long long)(per_state / 16);
    const int grid = (int)std::min<long long>((n16 + 255) / 256, 2048);
    for (size_t q = 0; q < c->states.size(); ++q) {
      if (c->south >= 0)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, c->side, (const uint4 *)c->states[q], (uint4 *)(rs + q * per_state),
                           block_q, row_q, c->first_owned - c->halo, c->halo, c->nblocks, 1);
      if (c->north >= 0)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, c->side, (const uint4 *)c->states[q], (uint4 *)(rn + q * per_state),
                           block_q, row_q, c->first_owned + c->rows_owned, c->halo, c->nblocks, 1);
    }
    GCMF_HIP(hipGetLastError());
    c->pending_unpack = false;
  }
  GCMF_HIP(hipEventRecord(c->ev_done, c->side));
  GCMF_HIP(hipStreamWaitEvent((hipStream_t)stream, c->ev_done, 0));
  c->states.clear();
  return GCMF_OK;
}

}  // extern "C"
