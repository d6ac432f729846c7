/*
 * gcmf.h -- C ABI of libgcmf, the MI355X (gfx950) implementation of the gcm-filters hot path.
 *
 * Drop-in boundary.  The reference (pure Python) has no FFI; the operator interface this library
 * replaces is the pair of closures created per Filter and called by xarray.apply_ufunc with raw
 * arrays whose last two axes are (y, x):
 *
 *     filter_func(field, *grid_args)            -> ndarray      gcm_filters/filter.py:177-212
 *     filter_func_vec(u, v, *grid_args)         -> (nd, nd)     gcm_filters/filter.py:242-289
 *
 * and the kernel plug-in protocol they drive                      gcm_filters/kernels.py:43-104
 *
 *     Laplacian(**grid_vars)   validation + precompute            -> gcmf_plan_create
 *     .prepare / .__call__ / .finalize                            -> gcmf_laplacian (one application)
 *     the n_steps recurrence around them                          -> gcmf_apply (whole polynomial)
 *
 * Everything is plain pointers and sizes; no torch / numpy types.  All entry points are thread-safe
 * with respect to *different* plans; calls on one plan are serialised by a per-plan mutex.
 *
 * Array conventions: C order, (nbatch, ny, nx) with x fastest.  Grid planes are 2-D (ny, nx) and
 * shared by every batch entry.  "Rows" are indices along y (the slow, sharded axis).
 */
#ifndef GCMF_H
#define GCMF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCMF_VERSION 1

/* ---- status codes -------------------------------------------------------------------------- */
typedef enum gcmf_status {
  GCMF_OK = 0,
  GCMF_ERR_INVALID_ARG = 1, /* bad pointer / size / enum / plane count                        */
  GCMF_ERR_HIP = 2,         /* a HIP runtime call failed; text in gcmf_last_error()           */
  GCMF_ERR_NO_DEVICE = 3,   /* no gfx950 device visible                                       */
  GCMF_ERR_UNSUPPORTED = 4,
  GCMF_ERR_P2P_TIMEOUT = 5, /* a peer-to-peer halo exchange of this rank failed (a neighbour never posted, or aborted): results NaN */
  /* Validation failures of the reference's Laplacian constructors.  The Python layer turns
   * them into the reference's exception type and message.                                   */
  GCMF_ERR_KAPPA_W_GT1 = 16,    /* ValueError  kernels.py:262-266                             */
  GCMF_ERR_KAPPA_S_GT1 = 17,    /* ValueError  kernels.py:268-272                             */
  GCMF_ERR_KAPPA_NONE_ONE = 18, /* ValueError  kernels.py:274-281                             */
  GCMF_ERR_WET_SOUTH_ROW = 19,  /* AssertionError kernels.py:458-459, 521-522                 */
  GCMF_ERR_DXN_FOLD = 20,       /* AssertionError kernels.py:551-554                          */
  GCMF_ERR_DYN_FOLD = 21        /* AssertionError kernels.py:559-562                          */
} gcmf_status;

/* ---- enums --------------------------------------------------------------------------------- */
/* Values equal the reference's GridType enum values (gcm_filters/kernels.py:13-28). */
typedef enum gcmf_grid_type {
  GCMF_REGULAR = 1,
  GCMF_REGULAR_AREA_WEIGHTED = 2,
  GCMF_REGULAR_WITH_LAND = 3,
  GCMF_REGULAR_WITH_LAND_AREA_WEIGHTED = 4,
  GCMF_IRREGULAR_WITH_LAND = 5,
  GCMF_MOM5U = 6,
  GCMF_MOM5T = 7,
  GCMF_TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED = 8,
  GCMF_TRIPOLAR_POP_WITH_LAND = 9,
  GCMF_VECTOR_C_GRID = 10,
  GCMF_VECTOR_B_GRID = 11
} gcmf_grid_type;

typedef enum gcmf_dtype { GCMF_F32 = 0, GCMF_F64 = 1 } gcmf_dtype;

/* gcmf_apply / gcmf_laplacian flags */
#define GCMF_DEVICE_PTRS 0x1u /* in/out are device pointers on the plan's device (else host)  */
#define GCMF_OUT_F32 0x2u     /* f32 plan only: write f32 output (default: f64, as NumPy >= 2 */
                              /* promotes p[k]*T)                       */
#define GCMF_FORWARD_RECURRENCE 0x4u /* gcmf_apply: sum the polynomial by the reference's forward     */
                              /* recurrence with its accumulation scheme (filter.py:192-206: f64 running sum   */
                              /* also for f32 state) even where the library would evaluate it backwards        */
                              /* (Clenshaw: f64 flux-form plans, C-grid plans)                                  */
#define GCMF_NO_RESIDENT 0x8u /* gcmf_apply / gcmf_slab_apply_backward: never use the on-chip (resident) kernel,   */
                              /* csrc/gcmf_resident.hip -- the strip-marching launches of 5..8 levels instead (same bits) */
#define GCMF_BACKWARD_F32 0x10u /* gcmf_apply, f32 scalar and B-grid plans: evaluate backwards (Clenshaw, all f32) like the f64 */
                              /* plans.  Faster (1.1-1.5 x) and 2-45 x further from f64 arithmetic than the reference's own f32  */
                              /* path (filter.py:192-206: f32 T_k, f64 running sum), which is the default for these plans.       */

/* Chebyshev step modes for gcmf_cheb_step */
#define GCMF_STEP_FIRST 0x1u /* T1 = A(T0);            fbar  = p0*T0 + p1*T1                  */
#define GCMF_STEP_LAST 0x2u  /* fbar result is finalised (divided by area) into fbar_out      */
#define GCMF_STEP_LAND_FIXED 0x8u /* gcmf_cheb_multi with GCMF_STEP_FIRST: the caller will overwrite the result of the cells
                                    gcmf_zero_land zeroes with gcmf_land_fix (or there are none), so the launch may take them as
                                    zero while it loads the field; the states it writes then satisfy GCMF_STEP_LAND_ZERO */
#define GCMF_STEP_CLENSHAW 0x10u /* gcmf_cheb_multi: S levels of the BACKWARD evaluation of the polynomial (see gcmf_clenshaw_cut) */
#define GCMF_STEP_LAND_ZERO 0x4u /* gcmf_cheb_multi[_vec]: the caller guarantees that the cells gcmf_zero_land zeroes are
                                    zero in both input states (lets the land-mask kernels drop their per-neighbour tests) */

typedef struct gcmf_plan gcmf_plan;

/* ---- plan ---------------------------------------------------------------------------------- */
typedef struct gcmf_plan_desc {
  int32_t grid_type;  /* gcmf_grid_type                                                        */
  int32_t dtype;      /* gcmf_dtype of the grid planes, the recurrence state T_k and L(f)      */
  int64_t ny, nx;     /* GLOBAL grid shape                                                     */
  int64_t row_begin;  /* rows [row_begin, row_end) of the global grid are owned by this plan   */
  int64_t row_end;    /*   single GPU: 0, ny                                                   */
  int32_t halo;       /* ghost rows kept on each slab edge that has a neighbour (0: one GPU)   */
  int32_t device;     /* HIP device ordinal                                                    */
  int32_t planes_on_device; /* grid planes are device pointers on `device`                     */
  int32_t flags;      /* GCMF_PLAN_* bits, 0 by default                                        */
} gcmf_plan_desc;

/* A plan that covers the whole (periodic, non-tripolar) grid but keeps `halo` ghost rows per edge like a slab whose
 * two neighbours are itself: the one-rank form of the row-slab driver (tests of the exchange path on a single GPU). */
#define GCMF_PLAN_SELF_RING 0x1
/* IRREGULAR_WITH_LAND: do not fail with GCMF_ERR_KAPPA_NONE_ONE.  For grid variables with leading (level / time) dims the
 * reference applies that check to the whole array (kernels.py:274-281), not per (y, x) plane; the Python layer then does
 * it itself and builds one plan per plane with this flag. */
#define GCMF_PLAN_SKIP_KAPPA_ONE 0x2

/*
 * Build a plan = the reference's `Laplacian(**grid_vars)` (kernels.py __post_init__ methods):
 * validates the grid planes, folds masks / kappas / metric ratios into coefficient planes resident
 * in HBM, and sizes the recurrence state.  `planes` holds `nplanes` GLOBAL (ny, nx) arrays of
 * `desc->dtype`, in the order of `required_grid_args()` of the reference class (kernels.py:58-63):
 *
 *   REGULAR: -            REGULAR_AREA_WEIGHTED: area      REGULAR_WITH_LAND: wet_mask
 *   REGULAR_WITH_LAND_AREA_WEIGHTED / TRIPOLAR_REGULAR_...: area, wet_mask
 *   IRREGULAR_WITH_LAND: wet_mask,dxw,dyw,dxs,dys,area,kappa_w,kappa_s
 *   MOM5U: wet_mask,dxt,dyt,dxu,dyu,area_u      MOM5T: wet_mask,dxt,dyt,dxu,dyu,area_t
 *   TRIPOLAR_POP_WITH_LAND: wet_mask,dxe,dye,dxn,dyn,tarea
 *   VECTOR_C_GRID: wet_mask_t,wet_mask_q,dxT,dyT,dxCu,dyCu,dxCv,dyCv,dxBu,dyBu,area_u,area_v,
 *                  kappa_iso,kappa_aniso
 *   VECTOR_B_GRID: DXU,DYU,HUS,HUW,HTE,HTN,UAREA,TAREA
 *
 * Returns GCMF_OK or a status; validation failures return the GCMF_ERR_KAPPA_x / GCMF_ERR_WET_x /
 * GCMF_ERR_Dxx_FOLD codes and create no plan.
 */
int gcmf_plan_create(const gcmf_plan_desc *desc, const void *const *planes, int nplanes, gcmf_plan **out);
void gcmf_plan_destroy(gcmf_plan *plan);

/* Static facts (no device needed). */
int gcmf_grid_nplanes(int grid_type);        /* number of grid planes, -1 if unknown           */
int gcmf_grid_ncomp(int grid_type);          /* 1 scalar, 2 vector                             */
int gcmf_grid_is_dimensional(int grid_type); /* kernels.py `is_dimensional`                    */
int gcmf_grid_is_tripolar(int grid_type);

/* Local (slab) geometry of a plan: rows allocated incl. ghosts, index of the first owned row
 * inside the allocation, number of owned rows.  State arrays passed to gcmf_cheb_step have
 * `rows_alloc` rows per batch entry. */
int gcmf_plan_rows(const gcmf_plan *plan, int64_t *rows_alloc, int64_t *first_owned, int64_t *rows_owned);

/* ---- whole filter: the body of filter_func / filter_func_vec (filter.py:185-210, 250-289) ---- */
/*
 * out = finalize( sum_k p[k] T_k ),  T_0 = prepare(in), T_1 = A(T_0), T_k = 2 A(T_{k-1}) - T_{k-2},
 * A(x) = -x - c L(x).   `p` has n_steps+1 entries (host memory), n_steps >= 1.
 * `c` = 2/s_max (dimensional Laplacians) or 2/(s_max*dx_min^2)   (filter.py:170-173).
 * `in` / `out`: ncomp pointers (1 scalar, 2 vector) to (nbatch, ny, nx) arrays; `in` has the plan
 * dtype and is not modified; `out` is f64 unless the plan is f32 and GCMF_OUT_F32 is given and must not alias `in`.
 * Only valid on single-slab plans (row_begin = 0, row_end = ny).
 * `stream`: hipStream_t to run on.  With GCMF_DEVICE_PTRS the work is enqueued asynchronously on exactly
 * that stream (NULL = the HIP default stream), ordered with the caller's other work on it.  A stream handed in here must stay
 * alive until the NEXT gcmf_apply on this plan has returned (or the plan is destroyed): when that next call arrives on another
 * stream it orders itself behind this one with an event recorded on this stream.  With host
 * pointers the call stages through HBM and is synchronous (NULL = a private stream of the plan); a batch of host
 * fields is cut into chunks whose upload, filtering and download overlap, and the input range is page-locked
 * (hipHostRegister, best effort) while the call runs -- see DESIGN.md section 6 and the GCMF_HOST_* variables.
 * Scheduling is internal: up to 8 (vector kinds: 4) recurrence steps per pass over HBM; flux-form scalar grids keep
 * cells with four closed faces (land) out of the recurrence state and add their neighbour-free polynomial at the end.
 * The result is the reference's for every cell, land included.
 */
int gcmf_apply(gcmf_plan *plan, const double *p, int n_steps, double c, const void *const *in,
               void *const *out, int64_t nbatch, uint32_t flags, void *stream);

/* One application of the Laplacian: `ALL_KERNELS[grid_type](**grid_vars)(field)` (kernels.py
 * __call__ methods; no prepare/finalize).  `out` has the plan dtype. */
int gcmf_laplacian(gcmf_plan *plan, const void *const *in, void *const *out, int64_t nbatch,
                   uint32_t flags, void *stream);

/* ---- building blocks for the multi-GPU (row-slab) driver; device pointers only --------------- */
/*
 * One Chebyshev step on rows [row_lo, row_hi) of the slab allocation (see gcmf_plan_rows):
 *   t1:      T_{k-1}  (stencil input; rows row_lo-1 .. row_hi must be valid or wrap/fold)
 *   t2:      T_{k-2}  (centre only; ignored with GCMF_STEP_FIRST)
 *   fbar_in: running sum (centre only; ignored with GCMF_STEP_FIRST)
 *   t0:      T_k out  (may alias t2; may be NULL with GCMF_STEP_LAST)
 *   fbar_out: running sum out (may alias fbar_in); with GCMF_STEP_LAST it receives finalize(fbar)
 * coef0 = p[k] (or p[0] with FIRST), coef1 = p[1] (FIRST only).
 * fbar arrays are f64 for f64 plans; for f32 plans f64 unless GCMF_OUT_F32.
 * Asynchronous on `stream` (NULL = the HIP default stream).
 */
int gcmf_cheb_step(gcmf_plan *plan, const void *const *t1, const void *const *t2,
                   const void *const *fbar_in, void *const *t0, void *const *fbar_out, double coef0,
                   double coef1, double c, uint32_t mode, uint32_t flags, int64_t nbatch,
                   int64_t row_lo, int64_t row_hi, void *stream);

/*
 * S recurrence steps in ONE pass over HBM (temporal blocking, scalar grid types; S in 2..8):
 *   u = T_{k-1}, v = T_{k-2} (ignored with GCMF_STEP_FIRST)  ->  uo = T_{k-1+S}, vo = T_{k-2+S}
 *   fbar_out = fbar_in + sum_t pk[t] T_{k+t}  (t = 0..S-1; with FIRST: p0*T_0 + pk[0]*T_1 + ...)
 * on rows [row_lo, row_hi) of the slab allocation; the inputs must be valid on [row_lo-S, row_hi+S) (clipped
 * at a wrapped / closed / folded physical boundary, which the kernel handles).  uo / vo must not alias u / v;
 * fbar_out may alias fbar_in.  prepare() is fused with FIRST, finalize() with LAST (uo/vo then unused).
 * Results are bit-identical to S calls of gcmf_cheb_step.  gcmf_multi_supported tells whether a plan/S
 * combination is available (vector grids, odd nx, very short grids are not).
 */
int gcmf_multi_supported(const gcmf_plan *plan, int S);
/*
 * Backward (Clenshaw) evaluation of the same polynomial  sum_k p[k] T_k(A) f  (reference filter.py:162-212):
 *   b_{n+1} = b_{n+2} = 0,  b_k = p[k] f + 2 A(b_{k+1}) - b_{k+2}  (k = n..1),  result = p[0] f + A(b_1) - b_2.
 * The state is two planes instead of three (no fbar); every launch re-reads the constant input f: one plane less per launch.
 * Same NaN / land semantics as the forward recurrence; results differ from it in the last bits (<= 3e-15 relative) and are
 * identical however the levels are cut.  gcmf_apply uses it by default for the f64 flux-form grid types without a tripole fold
 * and for VECTOR_C_GRID (f32 state: the whole recurrence then runs in f32, 2e-6 from the f64-accumulated forward result at
 * n = 44); env GCMF_CLENSHAW = 0 off / 1 those / 2 also the other f64 scalar types.  gcmf_cheb_multi_vec (vector slab callers)
 * runs the forward recurrence.
 * gcmf_clenshaw_cut: the launch depths (each 5..8, summing to n_steps) gcmf_apply uses for this plan and n_steps -- return value
 * = their number, 0 = it runs the forward recurrence.  A slab caller does the same with gcmf_cheb_multi and
 * GCMF_STEP_CLENSHAW:  level l = 1..n_steps computes b_{n-l};  launch with levels l0 .. l0+S-1:  u = b_{n-l0+1}, v = b_{n-l0+2}
 * (FIRST, l0 = 1: ignored, the launch forms b_n = p0 * f itself: pass p0 = p[n]),  fbar_in = f,  pk[t] = p[n - (l0 + t)],
 * uo = b_{n-l0-S+1}, vo = b_{n-l0-S+2};  LAST (l0 + S - 1 = n): fbar_out = the result (finalize() applied), uo / vo unused.
 * Isolated (land) cells are taken as zero throughout: run gcmf_land_fix on the result when gcmf_has_land(plan).
 */
int gcmf_clenshaw_cut(const gcmf_plan *plan, int n_steps, int *depths, int max_depths);
int gcmf_cheb_multi(gcmf_plan *plan, const void *u, const void *v, void *uo, void *vo, const void *fbar_in,
                    void *fbar_out, const double *pk, int S, double p0, double c, uint32_t mode, uint32_t flags,
                    int64_t nbatch, int64_t row_lo, int64_t row_hi, void *stream);

/*
 * The same for every grid type, with per-component pointer arrays like gcmf_cheb_step (ncomp entries each).
 * Scalar plans forward to gcmf_cheb_multi.  VECTOR_C_GRID (reference kernels.py:591-699 inside the recurrence of
 * filter.py:225-283) advances S in 2..5 steps per pass (f64 plans: 2..4); levels are processed by lock-step
 * workgroups of 4 (a batch that is not a multiple of 4 is padded internally, nothing is stored for the padding).
 * VECTOR_B_GRID (kernels.py:702-840): the same, f32 batches up to S = 6.
 */
int gcmf_multi_supported_vec(const gcmf_plan *plan, int S, int64_t nbatch);
int gcmf_cheb_multi_vec(gcmf_plan *plan, const void *const *u, const void *const *v, void *const *uo,
                        void *const *vo, const void *const *fbar_in, void *const *fbar_out, const double *pk,
                        int S, double p0, double c, uint32_t mode, uint32_t flags, int64_t nbatch,
                        int64_t row_lo, int64_t row_hi, void *stream);

/*
 * Land kept out of the recurrence state (flux-form and land-mask scalar grid types; what gcmf_apply does internally,
 * for slab drivers).  A land cell / a cell whose four faces are closed has L = 0 at every step (the wet mask zeroes
 * it or its fluxes, kernels.py:163-187, 286-315, 538-585) and evolves on its own.  gcmf_has_land: 1 if the plan has such cells and the two calls
 * below are available.  gcmf_zero_land: zero them in two state arrays (slab layout, e.g. the outputs of the first
 * gcmf_cheb_multi_vec call) so that NaN on land stops feeding the NaN / inf bookkeeping of the blocked kernels.
 * gcmf_land_fix: write their neighbour-free polynomial (prepare / finalize included, same operations and order as the
 * stencil kernels) over `out`, computed from the original field `in` (slab layout); p = n_steps + 1 host doubles.
 */
int gcmf_has_land(const gcmf_plan *plan);
int gcmf_zero_land(gcmf_plan *plan, void *const *a, void *const *b, int64_t nbatch, void *stream);
int gcmf_land_fix(gcmf_plan *plan, const double *p, int n_steps, double c, const void *const *in,
                  void *const *out, int64_t nbatch, uint32_t flags, void *stream);

/* ---- halo exchange of the row-slab driver, issued from C++ (RCCL send / recv over xGMI on a side stream) --------- */
/*
 * The reference has no spatial decomposition (filter.py:478-486 parallelises over non-core dims only); SURVEY 8e.
 * gcmf_comm_unique_id: 128 bytes (ncclUniqueId) generated on one rank and handed to every rank by the caller's own
 * channel (torch.distributed / MPI broadcast).  gcmf_comm_create: collective over `world` processes, one GPU each.
 * gcmf_halo_start: after everything enqueued on `stream` so far, send the `halo` owned edge rows of every block of the
 * `nstate` state arrays ((nblocks, rows_alloc, nx), nblocks = ncomp * nbatch, gcmf_dtype `dtype`) to the `south` / `north`
 * peer ranks (-1: physical boundary, nothing is exchanged there) and receive this slab's ghost rows; returns at once,
 * work enqueued on `stream` afterwards overlaps with the transfer.  gcmf_halo_finish: `stream` waits for the
 * exchange.  Between the two calls the caller may update any row except the 2 x halo sent and the ghost rows.
 */
typedef struct gcmf_comm gcmf_comm;
int gcmf_comm_unique_id(void *id128);
int gcmf_comm_create(const void *id128, int world, int rank, int device, gcmf_comm **out);
void gcmf_comm_destroy(gcmf_comm *comm);
/* What RCCL itself reports about the communicator: ncclGetVersion, ncclCommCount, ncclCommUserRank (-1 where unavailable). */
int gcmf_comm_info(gcmf_comm *comm, int *version, int *nranks, int *rank);
int gcmf_halo_start(gcmf_comm *comm, void *const *states, int nstate, int64_t nblocks, int64_t rows_alloc, int64_t nx,
                    int64_t first_owned, int64_t rows_owned, int halo, int dtype, int south, int north, void *stream);
int gcmf_halo_finish(gcmf_comm *comm, void *stream);

/* T_0 = prepare(field) = field * area for the AREA_WEIGHTED grid types (kernels.py:100-101),
 * a copy otherwise; rows [row_lo,row_hi) of the slab allocation. */
int gcmf_prepare(gcmf_plan *plan, const void *const *in, void *const *out, int64_t nbatch,
                 int64_t row_lo, int64_t row_hi, void *stream);

/* ---- timing of the last gcmf_apply (hipEvents on the stream the kernels ran on) -------------- */
/* ms_total: whole recurrence; n_launches: kernel launches it took. */
int gcmf_last_timing(const gcmf_plan *plan, float *ms_total, int *n_launches);
/* Enable/disable event timing inside gcmf_apply (adds two hipEventRecord per call).  enabled = 2: additionally one event
 * pair around every temporally blocked launch (the dominant kernel), read back with gcmf_last_kernel_timing: sum, count,
 * shortest and longest of the launches of the dominant kernel (gcmf_last_kernel) in the last gcmf_apply (device-resident calls, up to 32768 batch entries). */
int gcmf_set_timing(gcmf_plan *plan, int enabled);
int gcmf_last_kernel_timing(const gcmf_plan *plan, float *ms_sum, int *n_launches, float *ms_min, float *ms_max);
/* Name (as rocprofv3 prints it, without "void " and the argument list) of the recurrence kernel that advanced the most
 * steps per launch since this was last called; "" if none ran.  Reading resets it.  Instrumentation only: bench.py
 * refuses to quote profiled HBM traffic for a kernel other than the one that ran. */
int gcmf_last_kernel(gcmf_plan *plan, char *buf, int n);
/* Launch geometry of that kernel ("H=<rows per strip> nstrips=.. nwx=<column windows> xcd=<0|1> grid=XxY rows=..";
 * "" for kernels that are not strip-marched).  Survives gcmf_last_kernel's reset.  A PMC traffic record only
 * describes the kernel at the geometry it was profiled with. */
int gcmf_last_kernel_geometry(gcmf_plan *plan, char *buf, int n);
/* Wave strips of the register-ring kernels (k_ring) that met a NaN / inf since this was last called and were redone by the
 * general kernel (results are the same; each costs about two strip times).  Synchronises the device; reading resets.  A large
 * count on ocean data means non-finite values in wet cells (NaN on land is masked on load and costs nothing). */
int gcmf_ring_fallbacks(gcmf_plan *plan, int64_t *count);

/* ---- peer-to-peer halo exchange for the ranks of one node (csrc/gcmf_p2p.hip; SURVEY 8e "latency escape hatch": direct peer stores
 * into the neighbour's memory + flags instead of an RCCL group).  Every rank owns a mailbox block its two neighbours map through
 * HIP IPC: gcmf_p2p_create (mailbox_bytes >= the largest message: nstate * nblocks * halo * nx * sizeof(T)), gcmf_p2p_export (64-byte
 * hipIpcMemHandle_t, handed to the neighbours by the caller's own channel), gcmf_p2p_connect (south / north handles, NULL = a physical
 * boundary; *_is_self: that neighbour is this very process), then per exchange gcmf_p2p_start (after the launches that produced the edge
 * rows) ... interior launches ... gcmf_p2p_finish (before the next launch that reads the ghost rows), both enqueued on the caller's
 * compute stream: two small kernels, no events, no host round trip.  Arguments of gcmf_p2p_start as gcmf_halo_start.
 * The block is fine-grained device memory (coherent between agents while kernels run); remote memory is only written.
 * FAILURE IS LOUD: waits inside the kernels are bounded (GCMF_P2P_TIMEOUT_MS, default 30000; gcmf_p2p_set_timeout_ms).  A wait that runs
 * out marks this rank failed (sticky), raises `abort` in both neighbours (the failure travels round the ring at once), delivers NaN
 * ghost rows instead of stale ones, and gcmf_p2p_guard -- enqueued by the slab drivers after the last launch of an application --
 * turns the application's result into NaN.  gcmf_p2p_status reads the failure word from mapped host memory (0 ok, 1 timed out, 2
 * aborted by a neighbour; no device call); gcmf_p2p_start and gcmf_slab_apply_backward return GCMF_ERR_P2P_TIMEOUT on a failed
 * exchange.  gcmf_p2p_seq: exchanges started so far (equal on every rank of a healthy run).  gcmf_p2p_debug_skip_post: test hook,
 * drops the post of exchange number `seq` on this rank. */
typedef struct gcmf_p2p gcmf_p2p;
int gcmf_p2p_create(int device, int64_t mailbox_bytes, gcmf_p2p **out);
int gcmf_p2p_export(gcmf_p2p *p, void *handle64);
int gcmf_p2p_connect(gcmf_p2p *p, const void *south_handle64, const void *north_handle64, int south_is_self, int north_is_self);
int gcmf_p2p_start(gcmf_p2p *p, void *const *states, int nstate, int64_t nblocks, int64_t rows_alloc, int64_t nx, int64_t first_owned,
                   int64_t rows_owned, int halo, int dtype, void *stream);
int gcmf_p2p_finish(gcmf_p2p *p, void *stream);
int gcmf_p2p_guard(gcmf_p2p *p, void *out, int64_t bytes, void *stream);
int gcmf_p2p_status(gcmf_p2p *p, int *failed);
int gcmf_p2p_seq(gcmf_p2p *p, int64_t *seq);
int gcmf_p2p_set_timeout_ms(gcmf_p2p *p, int64_t ms);
int gcmf_p2p_debug_skip_post(gcmf_p2p *p, int seq);
void gcmf_p2p_destroy(gcmf_p2p *p);

/* ---- one whole filter application on this rank's slab, backward (Clenshaw) evaluation, scalar kinds, in ONE call: launches,
 * ghost-zone bookkeeping, the edge / interior split that overlaps the exchange, and the halo exchanges themselves (through `comm` --
 * RCCL -- or `p2p` -- mailboxes; both NULL for a slab without neighbours), all enqueued on `stream`.  X: this rank's input with its own rows
 * filled in, pool: four state planes, out: the result (f64; the state dtype with GCMF_OUT_F32), all (nbatch, rows_alloc, nx) device
 * arrays; cut / ncut: gcmf_clenshaw_cut; halo: ghost rows per side (>= the deepest launch); south / north: peer ranks or -1; overlap != 0:
 * post the exchange between the edge strips and the interior of the launch that uses up the ghost zone (slabs of >= 4 halo rows).
 * What gcm_filters_amd/distributed.py otherwise does from Python, ~15 us of host time per launch and 13-33 us per exchange. */
int gcmf_slab_apply_backward(gcmf_plan *plan, gcmf_comm *comm, gcmf_p2p *p2p, int south, int north, const double *p, int n_steps, double c,
                             const int *cut, int ncut, void *X, void *const *pool, void *out, int64_t nbatch, int halo, int overlap,
                             uint32_t flags, void *stream);

/* The same for the VECTOR kinds (VECTOR_C_GRID, VECTOR_B_GRID; reference filter.py:217-291 filter_func_vec on a row slab): X / out = the two
 * components, pool = four state plane pairs (pool[2 q + component]); the levels are cut as gcmf_apply cuts them for this plan (at most four
 * per launch), the ghost zone (halo >= 4 rows) is refreshed -- both states of both components in one message per neighbour -- when the next
 * launch needs more rows than are left.  gcmf_slab_backward_vec_supported: does this plan / batch have a backward vector kernel (every
 * rank of a run must get the same answer; halo 0 = do not check the ghost depth)? */
int gcmf_slab_backward_vec_supported(const gcmf_plan *plan, int64_t nbatch, int halo);
int gcmf_slab_apply_backward_vec(gcmf_plan *plan, gcmf_comm *comm, gcmf_p2p *p2p, int south, int north, const double *p, int n_steps, double c,
                                 void *const *X, void *const *pool, void *const *out, int64_t nbatch, int halo, uint32_t flags, void *stream);

/* ---- the on-chip (resident) kernel, csrc/gcmf_resident.hip: the north star's "one persistent field per GPU with the whole n_steps
 * polynomial fused into a single launch, 2-D blocking with LDS-staged halo tiles", for fields that fit the register files + LDS of the
 * chip (the 300-row slab of an 8-way cut of 2400 x 3600 with its ghost rows; 512 x 512; up to ~1.5 M f64 cells).  One workgroup per CU owns
 * a 2-D tile for up to 64 levels of the backward (Clenshaw) evaluation of filter.py:162-212's polynomial; tiles trade their 4-cell edge
 * bands through the memory-side cache with per-tile epoch flags every 4 levels.  Bit-identical to the strip-marching launches.
 * gcmf_apply uses it by itself for whole grids up to ~420 k cells (one launch for the whole polynomial: IRREGULAR 512 x 512, n 63: 88 us
 * against 179 us); on the row slabs of a multi-GPU run it measured slower (a tile exchange costs ~9.5 us against ~1 us per level,
 * DESIGN.md 3.6), so gcmf_slab_apply_backward uses it only with env GCMF_RESIDENT=1 (=0 forbids it everywhere; GCMF_NO_RESIDENT per
 * call); these two entries are the building block and always available.
 * gcmf_resident_supported: can L levels with output rows [row_lo, row_hi) run in one resident launch?  gcmf_resident_levels: run them --
 * (u, v) = (b_{k+1}, b_{k+2}) (GCMF_STEP_FIRST: unused), f = the constant input, pk[l] = coefficient of level l + 1, p0 = p_n;
 * GCMF_STEP_LAST: `out` gets the result, otherwise (uo, vo) the new states.  Bit-identical to the same levels through gcmf_cheb_multi
 * with GCMF_STEP_CLENSHAW.  Two PROCESSES must not run resident launches on one GPU at the same time (see the file's header). */
int gcmf_resident_supported(const gcmf_plan *plan, int64_t row_lo, int64_t row_hi, int L);
int gcmf_resident_levels(gcmf_plan *plan, const void *u, const void *v, void *uo, void *vo, const void *f, void *out, const double *pk,
                         int L, double p0, double c, uint32_t mode, int64_t row_lo, int64_t row_hi, void *stream);

/* Which of the two bit-identical paths the last gcmf_apply of a plan took, and how often each was taken since the plan was made
 * ("make the path taken observable").  The reference has no counterpart (one numpy path, filter.py:177-212).
 *   GCMF_PATH_RESIDENT            the whole polynomial in on-chip launches (k_resident)
 *   GCMF_PATH_STRIPS              the strip-marching launches (BASELINE-size grids, batches, f32, vector kinds, GCMF_RESIDENT=0 ...)
 *   GCMF_PATH_STRIPS_LOCK_BUSY    the grid qualified for the on-chip kernel, but ANOTHER PROCESS holds this GPU's resident lock
 *   GCMF_PATH_STRIPS_DISABLED     the grid qualified, but an on-chip launch of this process timed out earlier (see gcmf_resident_status)
 * counts (optional): int64[5], indexed by the path codes. */
#define GCMF_PATH_NONE 0
#define GCMF_PATH_RESIDENT 1
#define GCMF_PATH_STRIPS 2
#define GCMF_PATH_STRIPS_LOCK_BUSY 3
#define GCMF_PATH_STRIPS_DISABLED 4
int gcmf_plan_last_path(const gcmf_plan *plan, int *path, int64_t *counts);
/* This process's standing with the on-chip kernel on `device`: GCMF_RESIDENT_OK (it holds the per-GPU lock file -- given back
 * GCMF_RESIDENT_LOCK_IDLE_S seconds, default 5, after its last on-chip launch finished; 0 = kept until exit), GCMF_RESIDENT_LOCK_BUSY
 * (another process holds it), GCMF_RESIDENT_DISABLED (an on-chip launch of this process timed out: NaN result, strip-marching launches
 * from then on), GCMF_RESIDENT_OFF (never asked / GCMF_RESIDENT=0 / a CU mask).  failures: time-outs seen so far.  A time-out is
 * returned as GCMF_ERR_HIP ONCE, by the next gcmf_apply of the plan that issued the launch (the same call on the synchronising host
 * path); calls on other plans are never failed for it. */
#define GCMF_RESIDENT_OK 0
#define GCMF_RESIDENT_LOCK_BUSY 1
#define GCMF_RESIDENT_DISABLED 2
#define GCMF_RESIDENT_OFF 3
int gcmf_resident_status(int device, int *state, uint64_t *failures);


/* Tunables: rows marched per wave of the single-step kernel (0 keeps the default); XCD-aware tile order
 * (bit 0: 1 on, 0 off; bits 1-2: neighbouring strips of the backward flux kernels march in opposite directions -- 1 off, 2 on, 0 keep;
 * <0 keeps both); temporal blocking: low byte = recurrence steps fused per HBM pass (1 = off,
 * 2..8), bits 8-23 = rows per wave strip (0 = auto), bits 24-27 = operand rows in flight of the general kernels, bits 28-29 =
 * backward evaluation (1 off, 2 flux kinds, 3 all scalar kinds; 0 keep); 0 keeps the default. */
int gcmf_set_tuning(gcmf_plan *plan, int rows_per_wave, int xcd_remap, int multi);
/* Named per-plan switches (A/B testing, the parity tests): "cgrid_ring" 1 / 0 (the static-ring C-grid kernel of batched f32 levels,
 * gcmf_cgrid_ring.hip; 0 = k_cgrid_stream2c everywhere), "cgrid_ring_smax" 4 / 5 / 6 (levels per launch), "cgrid_ring_ncarry" 0 / 1 (1 = round 5's form: only
 * the last level keeps its previous row's scaled copies in registers; same bits), "cgrid_ring_hmax" (tallest strip, 0 = 96 rows), "ringc9" 1 / 0 (nine levels per k_ringc launch on whole f64 flux grids), "ringc_zip" 1 / 0 (k_ringcz: strips of f64 flux plans without a tripole
 * seam marched in pairs away from a shared seam wherever that marches fewer rows; same bits; 2 / 3 = always its early-exit / whole-period form), "ringc_smax" 0 / 5..9
 * (backward scalar launches of at most this many levels; 0 = the default cut), "zip_fold" 1 / 0 (tripolar f64 flux plans evaluated backwards: k_ringcz advances the seam's rows itself -- strips that start
 * at the seam, zipped with the strips of their mirror windows -- instead of k_fold_band beside / after the launch; same bits), "slab_nines" 0 / 1 (row-slab plans of f64 flux grids without a tripole seam: gcmf_clenshaw_cut offers nine levels per launch where that saves one; the
 * ranks of a run must set it alike and own a ghost zone of at least nine rows -- SlabFilter decides it collectively), "band_seq_cells" N (tripolar plans: a blocked launch over at most N cells runs the seam's
 * k_fold_band after itself -- 1024 threads per tile -- instead of beside itself on a side stream; default 3000000, 0 = never; same bits), "pack_batch" 1 / 0 (batches on short
 * grids: the fields as one column of rows per window, k_ringcp; same bits), "single_launch" 0 / 1 (whole f64 flux-form grids of ANY size
 * whose n_steps is a multiple of 9 or 8: the whole polynomial in ONE persistent launch, csrc/gcmf_ringc_one.hip -- the passes over HBM
 * separated by grid-wide barriers instead of launch boundaries; same bits, measured 7 % slower at 2400 x 3600, hence opt-in; also env
 * GCMF_SINGLE_LAUNCH=1), "clenshaw_f32" 0 / 1 (GCMF_BACKWARD_F32
 * for every call of this plan, the slab drivers and gcmf_clenshaw_cut included), "ring_flux_f32" 1 / 0 (the forward ring kernel of the f32
 * flux kinds, gcmf_ring_flux_f32.hip; 0 = k_flux_multi2, same bits).  Unknown names: GCMF_ERR_INVALID_ARG. */
int gcmf_set_option(gcmf_plan *plan, const char *name, int value);

/* Last error text of the calling thread (never NULL). */
const char *gcmf_last_error(void);
int gcmf_version(void);
/* sha256 (64 hex digits) of the sources and compiler flags this binary was built from; the loader
 * (gcm_filters_amd/_lib.py load()) recomputes it from csrc/ + include/gcmf.h and refuses a binary that does not match. */
const char *gcmf_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* GCMF_H */
