// Peer-to-peer halo exchange for the ranks of ONE node: the "latency escape hatch" of SURVEY 8e / section 5 ("direct peer stores into
// the neighbour's halo rows via hipDeviceEnablePeerAccess / IPC + flag").
//
// gcmf_exchange.hip moves the ghost rows with an RCCL send / recv group on a side stream: robust, but an exchange costs two
// cross-queue events plus RCCL's point-to-point latency (tens of microseconds), which is first-order once an 8-way slab of a
// 2400x3600 grid computes a whole application in 0.28 ms.  Here every rank owns a MAILBOX block in its HBM that its two
// neighbours map through HIP IPC and write into directly (xGMI peer stores on a multi-GPU node); arrival and consumption are
// signalled by sequence numbers in the same block:
//
//   exchange q (1, 2, ...), parity p = q & 1, per neighbour:
//     k_p2p_post     waits until the neighbour has consumed exchange q-2 (its ack, written into MY block), copies my edge rows
//                    into ITS mailbox[p], fences, and the last workgroup releases   its.arrive[p][side] = q
//     k_p2p_collect  acquires  my.arrive[p][side] >= q, copies my mailbox[p] into my ghost rows, and the last workgroup releases
//                    the neighbour's ack[side'] = q
//
// Both kernels are enqueued on the COMPUTE stream (post after the edge launches that produced the rows, collect before the next
// launch that reads the ghost rows): no side stream, no events, no host round trip; the interior launch between them is the
// overlap.  Two mailboxes per side, so a rank may run one exchange ahead of a slow neighbour.  A neighbour that is this same
// process (ring of one rank, or both neighbours of a two-rank ring being one peer) is written through ordinary pointers.
//
// FAILURE IS LOUD (round 4; round 3 copied whatever was in the mailbox after a timed-out wait and carried on).  Every wait is bounded
// (GCMF_P2P_TIMEOUT_MS, default 30 s: much longer than a host hiccup of a neighbour -- ranks are not host-synchronised between
// applications -- and still not a hung GPU).  A wait that runs out
//   * sets this rank's sticky `failed` word, which lives in MAPPED HOST memory: gcmf_p2p_status reads it without touching the device,
//     gcmf_p2p_start / gcmf_slab_apply_backward refuse to enqueue anything more on a failed exchange (GCMF_ERR_P2P_TIMEOUT),
//   * raises `abort` in BOTH neighbours' headers; every wait polls its own header's `abort`, so the failure travels round the ring
//     in microseconds instead of one time-out per rank, and every later wait of a failed rank returns at once,
//   * fills the ghost rows it was about to deliver with NaN (all-ones bytes) instead of stale mailbox contents, and
//   * gcmf_p2p_guard (enqueued by the slab drivers after the last launch of an application) overwrites the application's RESULT with
//     NaN on a failed rank: the stencils treat a NaN neighbour as zero (nan_to_num, kernels.py:286), so poisoned ghost rows alone
//     would not reach the output.
// Nothing is ever silently wrong: the result is NaN on the device and the host raises at its next call or synchronisation point.
//
// MEMORY MODEL.  The block (header + mailboxes) is allocated FINE-GRAINED (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), the
// one kind of device memory for which HIP promises coherence between agents WHILE kernels run (coarse-grained hipMalloc memory is
// only coherent at kernel boundaries: the home GPU's L2 may keep stale flag or payload lines).  Remote memory is only ever WRITTEN
// (payload, flags, acks, abort) with system-scope release stores after a system-scope fence; the owner polls with system-scope
// acquire loads.  This is the guarantee RCCL's and rocSHMEM's own flag protocols rest on.
//
// Validated on this pool with ranks SHARING one GPU (world 2 / 3 / 8: tests/test_gpu_distributed.py, exchange="p2p") and as a ring
// of one; it has not run across GPUs (no multi-GPU box here), which is why SlabFilter's exchange="auto" stays with RCCL and
// SlabFilter warns when p2p neighbours sit on different devices.
#include "gcmf_internal.hpp"

#include <cstdlib>
#include <cstring>

namespace gcmf {

struct P2PHeader {
  // line 0: written by the NEIGHBOURS (system-scope stores over IPC / xGMI), read here with system-scope loads
  unsigned arrive[2][2];  // [parity][side]: side 0 written by my SOUTHERN neighbour (rows for my south ghosts), 1 by my northern one
  unsigned ack[2];        // [side]: ack[0] written by my southern neighbour ("I have consumed what you sent me up to ..."), 1 northern
  unsigned abort;         // raised by a neighbour whose wait failed (or that was told to abort): the failure travels round the ring
  unsigned pad0[25];
  // line 1: this rank's own bookkeeping (never touched by a neighbour)
  unsigned cnt_post[2], cnt_collect[2];   // workgroup counters of my own kernels (the last workgroup publishes)
  unsigned failed;        // device-side copy of the sticky failure word (the host reads the mapped one, P2PArgs::host_failed)
  unsigned pad1[27];
};
static_assert(sizeof(P2PHeader) == 256, "two 128-byte lines");

struct P2PArgs {
  char *my_block;          // header + data[2 parity][2 side][cap]
  char *peer_block[2];     // [0] southern neighbour's block, [1] northern neighbour's (or NULL)
  const char *state[4];    // state arrays (device), nstate of them
  int nstate;
  long long nblocks, block_q;   // blocks per state, 16-byte units per block
  int row_q;                    // 16-byte units per row
  int rows_edge;                // halo rows
  int r_send[2];                // first row sent to [south, north]: first_owned / first_owned + rows_owned - halo
  int r_ghost[2];               // first ghost row filled from [south, north]
  long long cap;                // bytes per mailbox
  unsigned seq;
  long long spin_limit;         // wall-clock ticks (s_memrealtime, 100 MHz)
  unsigned *host_failed;        // mapped host word: 1 = a wait timed out, 2 = aborted by a neighbour
  int skip_post;                // test hook (gcmf_p2p_debug_skip_post): this rank "forgets" to post -- its neighbours must fail loudly
};

__device__ __forceinline__ uint4 *mailbox(char *block, int parity, int side, long long cap) {
  return reinterpret_cast<uint4 *>(block + sizeof(P2PHeader) + ((long long)(parity * 2 + side)) * cap);
}

// wait until *flag >= want (system scope).  Returns 0 = there, 1 = timed out, 2 = aborted (by a neighbour, or this rank failed before)
__device__ __forceinline__ int p2p_wait(const unsigned *flag, unsigned want, P2PHeader *mine, long long limit) {
  if (__hip_atomic_load(&mine->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 2;
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)(v - want) >= 0) return 0;
    if (__hip_atomic_load(&mine->abort, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) return 2;
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > limit) return 1;
    __builtin_amdgcn_s_sleep(8);
  }
}

// one thread: record the failure (sticky, host-visible) and pass it on to both neighbours
__device__ __forceinline__ void p2p_fail(const P2PArgs &A, P2PHeader *mine, int why) {
  __hip_atomic_store(&mine->failed, (unsigned)why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (A.host_failed && __hip_atomic_load(A.host_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
    __hip_atomic_store(A.host_failed, (unsigned)why, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int s = 0; s < 2; ++s)
    if (A.peer_block[s] && A.peer_block[s] != A.my_block)
      __hip_atomic_store(&reinterpret_cast<P2PHeader *>(A.peer_block[s])->abort, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// blockIdx.y = side (0: towards the south, 1: towards the north)
__global__ __launch_bounds__(256) void k_p2p_post(const P2PArgs A) {
  const int side = blockIdx.y;
  char *peer = A.peer_block[side];
  if (!peer || A.skip_post) return;
  P2PHeader *mine = reinterpret_cast<P2PHeader *>(A.my_block);
  P2PHeader *theirs = reinterpret_cast<P2PHeader *>(peer);
  __shared__ int bad;
  if (threadIdx.x == 0) {
    // the neighbour must have consumed the exchange that used this parity last (q - 2); its ack lands in MY header
    bad = (A.seq < 3) ? (int)__hip_atomic_load(&mine->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : p2p_wait(&mine->ack[side], A.seq - 2, mine, A.spin_limit);
    if (bad) p2p_fail(A, mine, bad);
  }
  __syncthreads();
  if (bad) return;   // nothing is delivered and `arrive` is not raised: the neighbour learns from its `abort` word
  // I am the neighbour's NORTHERN neighbour when I send south (side 0), its southern one when I send north
  uint4 *dst = mailbox(peer, A.seq & 1, 1 - side, A.cap);
  const long long per = (long long)A.rows_edge * A.row_q, per_state = per * A.nblocks, n = per_state * A.nstate;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long q = i / per_state, r1 = i - q * per_state, b = r1 / per, rem = r1 - b * per;
    dst[i] = reinterpret_cast<const uint4 *>(A.state[q])[b * A.block_q + (long long)A.r_send[side] * A.row_q + rem];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(&mine->cnt_post[side], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(&mine->cnt_post[side], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(&theirs->arrive[A.seq & 1][1 - side], A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// blockIdx.y = side (0: from the south, 1: from the north)
__global__ __launch_bounds__(256) void k_p2p_collect(const P2PArgs A) {
  const int side = blockIdx.y;
  char *peer = A.peer_block[side];
  if (!peer) return;
  P2PHeader *mine = reinterpret_cast<P2PHeader *>(A.my_block);
  P2PHeader *theirs = reinterpret_cast<P2PHeader *>(peer);
  __shared__ int bad;
  if (threadIdx.x == 0) {
    bad = p2p_wait(&mine->arrive[A.seq & 1][side], A.seq, mine, A.spin_limit);
    if (bad) p2p_fail(A, mine, bad);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // (every thread, system scope: the rows the flag announced)
  const uint4 *src = mailbox(A.my_block, A.seq & 1, side, A.cap);
  const uint4 poison = make_uint4(~0u, ~0u, ~0u, ~0u);   // NaN in f32 and in f64: never stale mailbox contents
  const long long per = (long long)A.rows_edge * A.row_q, per_state = per * A.nblocks, n = per_state * A.nstate;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long q = i / per_state, r1 = i - q * per_state, b = r1 / per, rem = r1 - b * per;
    reinterpret_cast<uint4 *>(const_cast<char *>(A.state[q]))[b * A.block_q + (long long)A.r_ghost[side] * A.row_q + rem] =
        bad ? poison : src[i];
  }
  __syncthreads();
  if (threadIdx.x == 0 && !bad) {
    const unsigned done = __hip_atomic_fetch_add(&mine->cnt_collect[side], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(&mine->cnt_collect[side], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // my southern neighbour sent these rows as ITS northward message: it waits on its ack[1]
      __hip_atomic_store(&theirs->ack[1 - side], A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// after the last launch of an application: a failed exchange must not leave a plausible-looking result behind
__global__ __launch_bounds__(256) void k_p2p_guard(const unsigned *failed, uint4 *out, long long n16) {
  if (!__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const uint4 poison = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) out[i] = poison;
}

}  // namespace gcmf

using namespace gcmf;

struct gcmf_p2p {
  int device = 0;
  char *block = nullptr;        // my mailbox block (fine-grained device memory)
  size_t block_bytes = 0;
  long long cap = 0;            // bytes per mailbox
  char *peer[2] = {nullptr, nullptr};
  bool peer_mapped[2] = {false, false};   // opened through IPC (to be closed), as opposed to a local pointer
  unsigned seq = 0;
  bool in_flight = false;
  unsigned *host_failed = nullptr;   // mapped host word the kernels raise on failure (host pointer)
  unsigned *dev_failed = nullptr;    // ... and its device address
  long long spin_limit = 0;
  int skip_post_at = 0;              // test hook: the exchange (sequence number) whose post this rank drops
  P2PArgs last{};
  std::mutex mu;
};

static int p2p_failed_now(gcmf_p2p *p) {
  return p->host_failed ? (int)__atomic_load_n(p->host_failed, __ATOMIC_ACQUIRE) : 0;
}

static int p2p_refuse(gcmf_p2p *p, const char *who) {
  const int why = p2p_failed_now(p);
  if (!why) return GCMF_OK;
  set_error("%s: the peer-to-peer halo exchange of this rank has failed (%s); its ghost rows and results are NaN. "
            "Destroy the gcmf_p2p and rebuild the exchange on every rank",
            who, why == 1 ? "a wait for a neighbour ran into GCMF_P2P_TIMEOUT_MS" : "a neighbour's wait failed and it raised abort");
  return GCMF_ERR_P2P_TIMEOUT;
}

extern "C" {

int gcmf_p2p_create(int device, int64_t mailbox_bytes, gcmf_p2p **out) {
  if (!out || mailbox_bytes < 16) {
    set_error("gcmf_p2p_create: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  *out = nullptr;
  GCMF_HIP(hipSetDevice(device));
  gcmf_p2p *p = new gcmf_p2p();
  p->device = device;
  p->cap = ((long long)mailbox_bytes + 255) / 256 * 256;
  p->block_bytes = sizeof(P2PHeader) + 4 * (size_t)p->cap;
  // FINE-GRAINED device memory: flags and payload are stored by a PEER GPU while this GPU's kernels poll and read them; only
  // fine-grained allocations are coherent between agents during a kernel (see the note at the top of this file)
  hipError_t e = hipExtMallocWithFlags((void **)&p->block, p->block_bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(p->block, 0, sizeof(P2PHeader));
  if (e == hipSuccess) e = hipHostMalloc((void **)&p->host_failed, 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    *p->host_failed = 0u;
    e = hipHostGetDevicePointer((void **)&p->dev_failed, p->host_failed, 0);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_error("gcmf_p2p_create: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    if (p->block) (void)hipFree(p->block);
    if (p->host_failed) (void)hipHostFree(p->host_failed);
    delete p;
    return GCMF_ERR_HIP;
  }
  long long ms = 30000;   // much longer than a host hiccup of a neighbour (plan folding, GC pause, reading the next field)
  if (const char *t = getenv("GCMF_P2P_TIMEOUT_MS")) {
    const long long v = atoll(t);
    if (v > 0) ms = v;
  }
  p->spin_limit = ms * 100000LL;   // s_memrealtime ticks at 100 MHz
  *out = p;
  return GCMF_OK;
}

int gcmf_p2p_set_timeout_ms(gcmf_p2p *p, int64_t ms) {
  if (!p || ms < 1) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  p->spin_limit = (long long)ms * 100000LL;
  return GCMF_OK;
}

int gcmf_p2p_export(gcmf_p2p *p, void *handle64) {
  if (!p || !handle64) return GCMF_ERR_INVALID_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  GCMF_HIP(hipSetDevice(p->device));
  hipIpcMemHandle_t h;
  GCMF_HIP(hipIpcGetMemHandle(&h, p->block));
  memcpy(handle64, &h, sizeof h);
  return GCMF_OK;
}

// south / north: the 64-byte handles the neighbours exported (NULL: a physical boundary).  *_is_self: that neighbour is this very
// process (a handle cannot be opened by the process that made it).  If both neighbours are the same OTHER process (a ring of two
// ranks) pass the same handle twice: it is opened once.
int gcmf_p2p_connect(gcmf_p2p *p, const void *south_handle64, const void *north_handle64, int south_is_self, int north_is_self) {
  if (!p) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  GCMF_HIP(hipSetDevice(p->device));
  const void *hs[2] = {south_handle64, north_handle64};
  const int self[2] = {south_is_self, north_is_self};
  for (int s = 0; s < 2; ++s) {
    p->peer[s] = nullptr;
    p->peer_mapped[s] = false;
    if (self[s]) {
      p->peer[s] = p->block;
      continue;
    }
    if (!hs[s]) continue;
    if (s == 1 && hs[0] && !self[0] && memcmp(hs[0], hs[1], 64) == 0) {   // the same peer on both sides
      p->peer[1] = p->peer[0];
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, hs[s], sizeof h);
    void *ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error("gcmf_p2p_connect: hipIpcOpenMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
      (void)hipGetLastError();
      return GCMF_ERR_HIP;
    }
    p->peer[s] = (char *)ptr;
    p->peer_mapped[s] = true;
  }
  return GCMF_OK;
}

// Post this rank's edge rows into its neighbours' mailboxes.  Enqueue on `stream` AFTER the launches that produced rows
// [first_owned, first_owned + halo) and [first_owned + rows_owned - halo, first_owned + rows_owned) of every state.
int gcmf_p2p_start(gcmf_p2p *p, void *const *states, int nstate, int64_t nblocks, int64_t rows_alloc, int64_t nx, int64_t first_owned,
                   int64_t rows_owned, int halo, int dtype, void *stream) {
  if (!p || !states || nstate < 1 || nstate > 4 || nblocks < 1 || halo < 1 || rows_owned < halo || first_owned + rows_owned > rows_alloc) {
    set_error("gcmf_p2p_start: bad argument");
    return GCMF_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(p->mu);
  if (int rc = p2p_refuse(p, "gcmf_p2p_start")) {
    p->in_flight = false;
    return rc;
  }
  if (p->in_flight) {
    set_error("gcmf_p2p_start: the previous exchange was not finished");
    return GCMF_ERR_INVALID_ARG;
  }
  const size_t es = dtype_size(dtype), row_bytes = (size_t)nx * es;
  if (row_bytes % 16) {
    set_error("gcmf_p2p_start: rows of %zu bytes are not a multiple of 16", row_bytes);
    return GCMF_ERR_UNSUPPORTED;
  }
  const size_t msg = (size_t)nstate * nblocks * halo * row_bytes;
  if ((long long)msg > p->cap) {
    set_error("gcmf_p2p_start: a message of %zu bytes does not fit the mailbox (%lld)", msg, p->cap);
    return GCMF_ERR_INVALID_ARG;
  }
  if ((p->peer[0] && first_owned < halo) || (p->peer[1] && rows_alloc - first_owned - rows_owned < halo)) {
    set_error("gcmf_p2p_start: the slab has fewer ghost rows than the halo");
    return GCMF_ERR_INVALID_ARG;
  }
  GCMF_HIP(hipSetDevice(p->device));
  P2PArgs A{};
  A.my_block = p->block;
  A.peer_block[0] = p->peer[0];
  A.peer_block[1] = p->peer[1];
  for (int q = 0; q < nstate; ++q) A.state[q] = (const char *)states[q];
  A.nstate = nstate;
  A.nblocks = nblocks;
  A.block_q = (long long)((size_t)rows_alloc * row_bytes / 16);
  A.row_q = (int)(row_bytes / 16);
  A.rows_edge = halo;
  A.r_send[0] = (int)first_owned;
  A.r_send[1] = (int)(first_owned + rows_owned - halo);
  A.r_ghost[0] = (int)(first_owned - halo);
  A.r_ghost[1] = (int)(first_owned + rows_owned);
  A.cap = p->cap;
  A.seq = p->seq + 1;
  A.spin_limit = p->spin_limit;
  A.host_failed = p->dev_failed;
  A.skip_post = (p->skip_post_at > 0 && (int)A.seq == p->skip_post_at) ? 1 : 0;   // gcmf_p2p_debug_skip_post
  const long long n16 = (long long)(msg / 16);
  const unsigned wgs = (unsigned)std::min<long long>(std::max<long long>((n16 + 255) / 256, 1), 64);
  hipLaunchKernelGGL(k_p2p_post, dim3(wgs, 2), dim3(256), 0, (hipStream_t)stream, A);
  GCMF_HIP(hipGetLastError());   // (a failed launch leaves seq / in_flight untouched: the exchange did not start)
  p->seq = A.seq;
  p->last = A;
  p->in_flight = true;
  return GCMF_OK;
}

// Collect the neighbours' rows into this rank's ghost rows.  Enqueue on `stream` BEFORE the next launch that reads them.
int gcmf_p2p_finish(gcmf_p2p *p, void *stream) {
  if (!p) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  if (!p->in_flight) {
    set_error("gcmf_p2p_finish: no exchange in flight");
    return GCMF_ERR_INVALID_ARG;
  }
  p->in_flight = false;   // whatever happens below, the next start is not refused for "previous exchange not finished"
  GCMF_HIP(hipSetDevice(p->device));
  const P2PArgs &A = p->last;
  const long long n16 = (long long)A.rows_edge * A.row_q * A.nblocks * A.nstate;
  const unsigned wgs = (unsigned)std::min<long long>(std::max<long long>((n16 + 255) / 256, 1), 64);
  hipLaunchKernelGGL(k_p2p_collect, dim3(wgs, 2), dim3(256), 0, (hipStream_t)stream, A);   // (on a failed rank: poisons the ghost rows)
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

// Enqueue after the LAST launch of an application: if an exchange of this rank has failed, `bytes` of `out` (a multiple of 16) become
// NaN -- the stencils ignore NaN neighbours (nan_to_num), so poisoned ghost rows alone would leave a plausible result.
int gcmf_p2p_guard(gcmf_p2p *p, void *out, int64_t bytes, void *stream) {
  if (!p || !out || bytes < 0 || bytes % 16) return GCMF_ERR_INVALID_ARG;
  GCMF_HIP(hipSetDevice(p->device));
  const long long n16 = bytes / 16;
  const unsigned wgs = (unsigned)std::min<long long>(std::max<long long>((n16 + 255) / 256, 1), 1024);
  hipLaunchKernelGGL(k_p2p_guard, dim3(wgs), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned *)(p->block + offsetof(P2PHeader, failed)), (uint4 *)out, n16);
  GCMF_HIP(hipGetLastError());
  return GCMF_OK;
}

// Has an exchange of this rank failed?  0 = no, 1 = one of its waits timed out, 2 = a neighbour aborted it.  Reads a word in mapped
// host memory: no device call, callable at any time; definitive for everything enqueued before the last synchronisation.
int gcmf_p2p_status(gcmf_p2p *p, int *failed) {
  if (!p || !failed) return GCMF_ERR_INVALID_ARG;
  *failed = p2p_failed_now(p);
  return GCMF_OK;
}

// Test hook: the post of exchange number `seq` (1, 2, ...) is dropped on this rank, as if the rank had fallen out of step; its
// neighbours' waits must fail loudly (tests/test_gpu_distributed.py).  0 = off.
int gcmf_p2p_debug_skip_post(gcmf_p2p *p, int seq) {
  if (!p || seq < 0) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  p->skip_post_at = seq;
  return GCMF_OK;
}

// Exchanges started so far (every rank of a slab run must report the same number: bench.py / tests all-gather it)
int gcmf_p2p_seq(gcmf_p2p *p, int64_t *seq) {
  if (!p || !seq) return GCMF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  *seq = p->seq;
  return GCMF_OK;
}

void gcmf_p2p_destroy(gcmf_p2p *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  for (int s = 0; s < 2; ++s)
    if (p->peer_mapped[s] && p->peer[s]) (void)hipIpcCloseMemHandle(p->peer[s]);
  if (p->block) (void)hipFree(p->block);
  if (p->host_failed) (void)hipHostFree(p->host_failed);
  delete p;
}

}  // extern "C"
