"""Multi-GPU filter: row slabs along y, one process per MI355X, halo rows exchanged over RCCL / xGMI.

The reference has no spatial decomposition (its only parallelism is dask over non-core dims,
gcm_filters/filter.py:478-486); this module is the MI355X-native extension the north star asks for.

Decomposition.  Rank r owns a contiguous block of rows of every (ny, nx) plane; x is never
split, so x-periodicity and the tripole fold (row ny-1 reads [ny-1, nx-1-i]) stay rank-local.  Every
Laplacian has stencil radius 1, so one Chebyshev step invalidates one ghost row per side.  Instead of a
(latency-bound, 28.8 KB) exchange per step, ranks carry `halo` = s ghost rows and exchange every s steps
(communication-avoiding / s-step halos): between exchanges each step recomputes a ghost zone that shrinks by
one row, costing s(s-1)/2 extra rows of work per side per cycle (0.3 % at s=8 on a 2400-row slab).

  * both recurrence states T_{k-1} and T_{k-2} need ghosts (ghost rows of T_k are recomputed from them);
    the running sum fbar does not (centre-only);
  * non-tripolar grids are periodic in y: ring neighbours (rank 0 <-> rank P-1 wrap);
  * tripolar grids: rank 0 has no southern neighbour (row 0 is land), rank P-1 folds locally;
  * grid coefficients need no exchange: every rank folds its slab (+ghost rows) from the global planes.

Messages per exchange and direction: ncomp * nbatch * s rows * nx * sizeof(T) * 2 states, one contiguous
buffer per peer (P2P send/recv; with 2 ranks both directions share one message so ordering is trivial).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .filter import FilterShape, _compute_filter_spec, _compute_n_steps_default
from .kernels import ALL_KERNELS, GridType


def slab_bounds(ny: int, world: int, rank: int):
    """Balanced contiguous split of ny rows."""
    base, rem = divmod(ny, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class HipSlabEngine:
    """Executes slab steps on the MI355X through libgcmf (gcmf_prepare / gcmf_cheb_step)."""

    def __init__(self, grid_type: GridType, dtype_code: int, ny: int, nx: int, planes: Sequence[np.ndarray],
                 row_begin: int, row_end: int, halo: int, device: int, self_ring: bool = False):
        self.plan = _lib.Plan(grid_type.value, dtype_code, ny, nx, planes, device=device, row_begin=row_begin,
                              row_end=row_end, halo=halo, self_ring=self_ring)
        self.rows_alloc, self.first_owned, self.rows_owned = (self.plan.rows_alloc, self.plan.first_owned,
                                                              self.plan.rows_owned)

    @staticmethod
    def _ptrs(tensors):
        return None if tensors is None else [t.data_ptr() for t in tensors]

    def _stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream

    def prepare(self, ins, outs, nbatch, row_lo, row_hi):
        self.plan.prepare(self._ptrs(ins), self._ptrs(outs), nbatch, row_lo, row_hi, stream=self._stream())

    def step(self, t1, t2, fb_in, t0, fb_out, coef0, coef1, c, mode, nbatch, row_lo, row_hi):
        self.plan.cheb_step(self._ptrs(t1), self._ptrs(t2), self._ptrs(fb_in), self._ptrs(t0), self._ptrs(fb_out),
                            coef0, coef1, c, mode, nbatch, row_lo, row_hi, stream=self._stream())

    def multi_supported(self, S, nbatch=1):
        return self.plan.multi_supported_vec(S, nbatch)

    def has_land(self):
        return self.plan.has_land()

    def zero_land(self, a, b, nbatch):
        self.plan.zero_land(self._ptrs(a), self._ptrs(b), nbatch, stream=self._stream())

    def land_fix(self, p, c, ins, outs, nbatch):
        self.plan.land_fix(p, c, self._ptrs(ins), self._ptrs(outs), nbatch, stream=self._stream())

    def clenshaw_cut(self, n_steps):
        """Launch depths of the backward evaluation gcmf_apply uses for this plan and polynomial ([] = forward recurrence)."""
        return self.plan.clenshaw_cut(n_steps) if self.plan.ncomp == 1 else []

    def multi(self, u, v, uo, vo, fb_in, fb_out, pk, p0, c, mode, nbatch, row_lo, row_hi):
        """S = len(pk) recurrence steps in one HBM pass (gcmf_cheb_multi_vec); per-component tensor lists."""
        self.plan.cheb_multi_vec(self._ptrs(u), self._ptrs(v), self._ptrs(uo), self._ptrs(vo), self._ptrs(fb_in),
                                 self._ptrs(fb_out), pk, p0, c, mode, nbatch, row_lo, row_hi, stream=self._stream())


class SlabFilter:
    """One rank's share of a filter over a (ny, nx) grid cut into `world` row slabs.

    Parameters mirror ``Filter``: ``grid_type`` (name or GridType), ``grid_vars`` = GLOBAL (ny, nx) host arrays
    (every rank folds its own slab from them), ``filter_kwargs`` = filter_scale, dx_min, filter_shape, ...
    ``engine_factory`` is for tests only (CPU stand-in for the HIP engine on the gloo backend).

    ``exchange``: who issues the halo exchange -- ``"native"``: libgcmf itself (gcmf_halo_start / gcmf_halo_finish: RCCL
    send / recv on a side stream, a few microseconds of host time per exchange), ``"torch"``: torch.distributed P2P ops
    (any backend; what the gloo tests use), ``"auto"``: native on one-GPU-per-rank RCCL groups, torch otherwise,
    ``"p2p"``: peer stores into the neighbours' IPC-mapped mailboxes + flags, two small kernels on the compute stream per
    exchange and no RCCL at all (csrc/gcmf_p2p.hip; the ranks of ONE node; any backend carries the 64-byte handles once).
    ``self_ring`` (one rank, periodic grids): keep ghost rows and exchange with itself -- the whole slab choreography
    incl. the native exchange on a single GPU.  ``evaluation``: as in ``Filter`` ("reference" = forward recurrence everywhere,
    "backward" = backward also for f32 scalar / B-grid state).
    """

    def __init__(self, grid_type, grid_vars: Dict[str, np.ndarray], filter_kwargs: dict, ny: int, nx: int, *,
                 halo: Optional[int] = None, dtype=np.float64, group=None, device=None, engine_factory=None,
                 rank: Optional[int] = None, world: Optional[int] = None, exchange: str = "auto",
                 self_ring: bool = False, evaluation: str = "auto"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.self_ring = bool(self_ring)
        if self.self_ring and self.world != 1:
            raise ValueError("self_ring is the one-rank form of the slab driver")
        self.multi = self.world > 1 or self.self_ring   # ghost rows + exchanges are in play
        self.grid_type = GridType[grid_type] if isinstance(grid_type, str) else grid_type
        self.lap_cls = ALL_KERNELS[self.grid_type]
        self.ncomp = self.lap_cls._NCOMP
        self.ny, self.nx = int(ny), int(nx)
        self.np_dtype = np.dtype(dtype)
        self.dtype_code = _lib.dtype_code(self.np_dtype)
        self.tripolar = bool(_lib.load().gcmf_grid_is_tripolar(self.grid_type.value))
        self.area_weighted = self.grid_type.name.endswith("AREA_WEIGHTED")

        fk = dict(filter_kwargs)
        shape = fk.get("filter_shape", FilterShape.GAUSSIAN)
        shape = FilterShape[shape] if isinstance(shape, str) else shape
        tw, ndim = fk.get("transition_width", np.pi), fk.get("ndim", 2)
        n = int(fk.get("n_steps", 0))
        if n < 3:
            n = int(_compute_n_steps_default(ndim, shape, fk["filter_scale"], fk["dx_min"], tw))
        self.spec = _compute_filter_spec(fk["filter_scale"], fk["dx_min"], shape, tw, ndim, n)
        self.n_steps = n
        self.c = 2 / self.spec.s_max if self.lap_cls.is_dimensional else 2 / (self.spec.s_max * self.spec.dx_min_sq)

        self.row_begin, self.row_end = slab_bounds(self.ny, self.world, self.rank)
        min_rows = self.ny // self.world
        if min_rows < 1:
            raise ValueError(f"{self.ny} rows cannot be split over {self.world} ranks")
        if not self.multi:
            self.halo = 0
        else:
            # default: as many ghost rows as the filter has steps, at most 64 -- ONE exchange per application (the input's ghost rows) for
            # filters of up to 64 steps, one more every 64 levels beyond.  An un-overlapped exchange costs 23-33 us of the stream -- as much
            # as a whole blocked 8-level launch on a 300-row slab -- while deeper ghost zones cost little there (the strips of a short slab
            # are latency-bound, and their levels ramp up, csrc/gcmf_ringc_impl.hpp).  Measured as a ring of one rank
            # (tools/measure_exchange.py, config 3, n_steps 63), ms per application at halo 32 / 48 / 64: 300 rows 0.300 / 0.312 / 0.267
            # (RCCL), 0.293 / 0.307 / 0.270 (p2p); 600 rows 0.388 / - / 0.353 (RCCL); 1200 rows 0.615 / - / 0.564 (RCCL).  (Round 3, before
            # the ramp: 32 was the optimum, 0.323 against 0.333 at 48.)
            auto = min(64, max(8, int(n)))
            self.halo = int(halo) if halo else auto
            self.halo = max(1, min(self.halo, min_rows))
        planes = [np.ascontiguousarray(np.asarray(grid_vars[k]), dtype=self.np_dtype)
                  for k in self.lap_cls.required_grid_args()]
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else -1
        self.device = torch.device("cuda", device) if device >= 0 else torch.device("cpu")
        factory = engine_factory or HipSlabEngine
        if self.self_ring:
            if self.tripolar:
                raise ValueError("self_ring needs a grid that is periodic in y")
            self.engine = factory(self.grid_type, self.dtype_code, self.ny, self.nx, planes, self.row_begin, self.row_end,
                                  self.halo, device, self_ring=True)
        else:
            self.engine = factory(self.grid_type, self.dtype_code, self.ny, self.nx, planes, self.row_begin,
                                  self.row_end, self.halo, device)
        self.rows_alloc, self.first_owned, self.rows_owned = (self.engine.rows_alloc, self.engine.first_owned,
                                                              self.engine.rows_owned)
        # neighbours; None where the slab edge is a physical boundary (tripolar) or there is a single rank
        P, r = self.world, self.rank
        self.gs = self.first_owned                                   # southern ghost rows
        self.gn = self.rows_alloc - self.first_owned - self.rows_owned  # northern ghost rows
        self.south = ((r - 1) % P) if (self.multi and self.gs > 0) else None
        self.north = ((r + 1) % P) if (self.multi and self.gn > 0) else None
        # who issues the exchange
        if exchange not in ("auto", "native", "torch", "p2p"):
            raise ValueError(f"exchange must be 'auto', 'native', 'torch' or 'p2p', not {exchange!r}")
        on_gpu = self.device.type == "cuda" and engine_factory is None
        rccl_group = self.self_ring or (dist.is_initialized() and dist.get_backend(group) == "nccl")
        if exchange == "native" and not (on_gpu and rccl_group):
            raise ValueError("exchange='native' needs one MI355X per rank (RCCL process group or self_ring)")
        self.exchange_kind = "native" if (self.multi and on_gpu and rccl_group and exchange not in ("torch", "p2p")) else "torch"
        self.p2p = None
        self._p2p_cap = 0
        if exchange == "p2p":
            if not on_gpu:
                raise ValueError("exchange='p2p' needs the HIP engine on an MI355X")
            if (self.nx * self.np_dtype.itemsize) % 16 == 0:   # (16-byte copies; every rank sees the same nx)
                self.exchange_kind = "p2p" if self.multi else "torch"
        self.comm = None
        if self.exchange_kind == "native" and self.multi:
            # "auto" falls back to torch.distributed P2P if libgcmf cannot bring up its own communicator (all ranks agree)
            err = None
            try:
                uid = [_lib.Comm.unique_id() if self.rank == 0 else None]
            except _lib.GcmfError as e:
                uid, err = [None], e
            if self.world > 1:
                dist.broadcast_object_list(uid, src=self._global_rank(0), group=group)
            if uid[0] is not None:
                try:
                    self.comm = _lib.Comm(uid[0], self.world, self.rank, device)
                except _lib.GcmfError as e:
                    err = e
            ok = torch.tensor([0 if self.comm is None else 1], device=self.device)
            if self.world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                if exchange == "native":
                    raise err if err is not None else RuntimeError("another rank could not create its gcmf_comm")
                if self.comm is not None:
                    self.comm.close()
                self.comm, self.exchange_kind = None, "torch"
        # the backward (Clenshaw) evaluation libgcmf uses on one GPU: every rank must take the same decision
        if evaluation not in ("auto", "reference", "backward"):
            raise ValueError(f"evaluation must be 'auto', 'reference' or 'backward', not {evaluation!r}")
        # evaluation="reference" (as in Filter): the forward recurrence with the reference's accumulation scheme on every rank;
        # "backward": the backward evaluation also for f32 scalar / B-grid state (the slab plan's clenshaw_f32 switch; all f32, faster)
        if evaluation == "backward" and hasattr(getattr(self.engine, "plan", None), "set_option"):
            self.engine.plan.set_option("clenshaw_f32", 1)
        # Nine levels per launch on the slabs of f64 flux grids without a tripole seam where that saves a launch (round 6: 63 = 7 x 9 instead of
        # 8 launches; k_ringcz / k_ringc at nine levels).  The plans only offer it when told to (option "slab_nines"), and the ranks of a run
        # must cut alike: decided here collectively -- every rank must qualify (64 rows, the kind, the dtype; a ghost zone nine rows deep) --
        # and taken per application by the batch and the slab's height (_cut_for), which every rank knows alike.
        plan = getattr(self.engine, "plan", None)
        self._cut9, c9, nines = [], [], 0
        can9 = hasattr(plan, "set_option") and hasattr(self.engine, "clenshaw_cut") and evaluation != "reference"
        if can9:
            try:
                plan.set_option("slab_nines", 1)
                c9 = list(self.engine.clenshaw_cut(self.n_steps))
                nines = 1 if (c9 and max(c9) == 9 and (not self.multi or self.halo >= 9) and os.environ.get("GCMF_SLAB_NINES", "1") != "0") else 0
            except _lib.GcmfError:
                c9, nines = [], 0
        # (every rank takes part in the reduction, whatever its engine: a rank that cannot run nines votes 0)
        rows_min = torch.tensor([nines, -int(self.rows_owned)], dtype=torch.int32)
        if self.world > 1 and dist.is_initialized():
            rows_min = rows_min.to(self.device if dist.get_backend(group) == "nccl" else torch.device("cpu"))
            dist.all_reduce(rows_min, op=dist.ReduceOp.MIN, group=group)
        nines, self._rows_owned_max = int(rows_min[0].item()), -int(rows_min[1].item())
        if can9:
            plan.set_option("slab_nines", 0)
        self._cut9 = c9 if nines else []
        cut = self.engine.clenshaw_cut(self.n_steps) if (hasattr(self.engine, "clenshaw_cut") and evaluation != "reference") else []
        use = 1 if (cut and (not self.multi or self.halo >= max(cut))) else 0
        if self.world > 1 and dist.is_initialized():
            flag = torch.tensor([use], dtype=torch.int32,
                                device=self.device if dist.get_backend(group) == "nccl" else torch.device("cpu"))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            use = int(flag.item())
        self.backward_cut = list(cut) if use else []
        self.evaluation = evaluation
        self._vec_backward = {}     # nbatch -> does every rank have a backward vector kernel for this batch (decided collectively, once)
        self.tdtype = torch.float64 if self.np_dtype == np.float64 else torch.float32
        self._bufs = {}
        self.kernel_ms = 0.0
        self.kernel_launches = 0
        self.kernel_apps = 0       # applications behind kernel_ms / kernel_launches
        self._pending_events = []
        self.exchanges = 0
        self.time_kernels = False  # bench.py: bracket every step launch with events on the launch stream
        self.multi_depth = 8       # most recurrence steps fused per HBM pass (1 = single steps only)
        # overlap the halo exchange with the interior rows of the last launch of a cycle: splitting that launch into two
        # edge strips + interior costs two small launches that march 2S warm-up rows for `halo` useful ones (~35 us
        # each); worth it only when the interior is long enough to hide an exchange behind (tall slabs)
        self.overlap = self.rows_owned >= 1000
        # backward scalar applications with a library-issued exchange ("native" RCCL or "p2p"): the whole choreography in ONE call into
        # libgcmf (gcmf_slab_apply_backward) instead of a Python loop of C calls
        self.native_driver = True
        # ... which can run every stretch between two exchanges as ONE on-chip launch when the slab fits the chip (csrc/gcmf_resident.hip;
        # measured slower than the strip-marching launches on the 8-way slab of 2400 x 3600, so libgcmf only does it with GCMF_RESIDENT=1,
        # DESIGN.md 3.6).  A resident kernel holds its CUs until its neighbour tiles have arrived, so two PROCESSES must never run one on the
        # same GPU at the same time: ranks that share a device (this repo's one-GPU test set-up, never a real run) are told apart here
        # and always take the strip-marching launches.
        self.resident = True
        self._shared_gpu = False
        if on_gpu and self.world > 1 and dist.is_initialized():
            import socket
            props = torch.cuda.get_device_properties(self.device)
            ident = (socket.gethostname(), str(getattr(props, "uuid", "")), str(getattr(props, "pci_bus_id", "")), self.device.index)
            every = [None] * self.world
            dist.all_gather_object(every, ident, group=group)
            self._shared_gpu = len(set(every)) < len(every)

    # -- data movement helpers -----------------------------------------------------------------
    def scatter_from_global(self, fields: Sequence[np.ndarray]):
        """Own rows of GLOBAL (..., ny, nx) host arrays -> list of device tensors (nbatch, rows_owned, nx)."""
        out = []
        for f in fields:
            f = np.asarray(f)
            loc = f[..., self.row_begin:self.row_end, :].reshape(-1, self.rows_owned, self.nx)
            out.append(self.torch.from_numpy(np.ascontiguousarray(loc, dtype=self.np_dtype)).to(self.device))
        return out

    def gather_to_global(self, local: Sequence):
        """All-gather filtered slabs back into GLOBAL (nbatch, ny, nx) host arrays (every rank gets them)."""
        res = []
        if self.device.type == "cuda":
            self.synchronize()
        for t in local:
            nb = t.shape[0]
            full = np.empty((nb, self.ny, self.nx), dtype=np.float64 if t.dtype == self.torch.float64 else np.float32)
            for src in range(self.world):
                b, e = slab_bounds(self.ny, self.world, src)
                buf = t.contiguous() if src == self.rank else self.torch.empty((nb, e - b, self.nx), dtype=t.dtype,
                                                                              device=t.device)
                if self.world > 1:
                    self.dist.broadcast(buf, src=self._global_rank(src), group=self.group)
                full[:, b:e, :] = buf.cpu().numpy()
            res.append(full)
        return res

    def _global_rank(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _state(self, nbatch: int):
        key = nbatch
        if key not in self._bufs:
            t = self.torch
            shape = (self.ncomp, nbatch, self.rows_alloc, self.nx)
            mk = lambda dt: t.zeros(shape, dtype=dt, device=self.device)
            self._bufs[key] = dict(X=mk(self.tdtype), A=mk(self.tdtype), B=mk(self.tdtype), C=mk(self.tdtype),
                                   D=mk(self.tdtype), F=mk(t.float64), F2=mk(t.float64), O=mk(t.float64))
        return self._bufs[key]

    # -- halo exchange -------------------------------------------------------------------------
    def _xbuf(self, key, n, like):
        """Persistent packed exchange buffer (one per peer / direction / size)."""
        key = key + (like.dtype,)
        b = self._bufs.get(key)
        if b is None:
            b = self.torch.empty(n, dtype=like.dtype, device=like.device)
            self._bufs[key] = b
        return b

    def _exchange(self, tensors: List):
        """Refresh all `halo` ghost rows of every tensor in `tensors` (each (ncomp, nbatch, rows_alloc, nx))."""
        self._exchange_finish(self._exchange_start(tensors))

    def _exchange_start(self, tensors: List):
        """Pack the boundary rows and post the sends / receives; returns a ticket for ``_exchange_finish``.  Work
        enqueued on the compute stream AFTER this call (the interior of the slab) overlaps with the transfer."""
        if not self.multi or (self.south is None and self.north is None):
            return None
        if self.exchange_kind == "p2p":   # peer stores into the neighbours' mailboxes, on the compute stream
            x0 = tensors[0]
            nblocks = x0.shape[0] * x0.shape[1]
            self._p2p_ready(2 * nblocks * self.halo * self.nx * x0.element_size())   # (sized for two states)
            self.p2p.start([x.data_ptr() for x in tensors], nblocks, self.rows_alloc, self.nx, self.first_owned, self.rows_owned,
                           self.halo, self.dtype_code, stream=self.torch.cuda.current_stream().cuda_stream)
            return "p2p"
        if self.comm is not None:  # libgcmf issues the RCCL send / recv pairs on its side stream
            x0 = tensors[0]
            self.comm.halo_start([x.data_ptr() for x in tensors], x0.shape[0] * x0.shape[1], self.rows_alloc, self.nx,
                                 self.first_owned, self.rows_owned, self.halo, self.dtype_code, self.south, self.north,
                                 stream=self.torch.cuda.current_stream().cuda_stream)
            return "native"
        t, dist = self.torch, self.dist
        s, fo, ro = self.halo, self.first_owned, self.rows_owned
        top = [x[:, :, fo + ro - s: fo + ro, :] for x in tensors]      # -> northern neighbour's south ghosts
        bot = [x[:, :, fo: fo + s, :] for x in tensors]                # -> southern neighbour's north ghosts
        ghost_s = [x[:, :, 0: s, :] for x in tensors] if self.gs else []
        ghost_n = [x[:, :, fo + ro: fo + ro + s, :] for x in tensors] if self.gn else []
        sends: Dict[int, List] = {}
        recvs: Dict[int, List] = {}
        # fixed order inside a peer's message: [what I send north | what I send south]; the receiver unpacks
        # [peer's northward rows -> my south ghosts | peer's southward rows -> my north ghosts]
        if self.north is not None:
            sends.setdefault(self.north, []).extend(top)
        if self.south is not None:
            sends.setdefault(self.south, []).extend(bot)
        if self.south is not None:
            recvs.setdefault(self.south, []).extend(ghost_s)
        if self.north is not None:
            recvs.setdefault(self.north, []).extend(ghost_n)
        # RCCL moves device buffers directly over xGMI; under gloo (CPU tests, or several ranks sharing one GPU
        # in the single-GPU parity test) the packed buffers are staged through host memory
        stage = self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"
        ops, unpack = [], []
        for peer, parts in sends.items():
            n = sum(p.numel() for p in parts)
            buf = self._xbuf(("s", peer, n), n, parts[0])
            off = 0
            for p in parts:  # pack straight into the persistent send buffer
                buf[off: off + p.numel()].view(p.shape).copy_(p)
                off += p.numel()
            if stage:
                buf = buf.cpu()
            ops.append(dist.P2POp(dist.isend, buf, self._global_rank(peer), group=self.group))
        for peer, parts in recvs.items():
            n = sum(p.numel() for p in parts)
            buf = (t.empty(n, dtype=parts[0].dtype, device="cpu") if stage
                   else self._xbuf(("r", peer, n), n, parts[0]))
            ops.append(dist.P2POp(dist.irecv, buf, self._global_rank(peer), group=self.group))
            unpack.append((buf, parts))
        works = dist.batch_isend_irecv(ops)
        return works, unpack, stage

    def _exchange_finish(self, ticket):
        if ticket is None:
            return
        if ticket == "p2p":
            self.p2p.finish(stream=self.torch.cuda.current_stream().cuda_stream)
            self.exchanges += 1
            return
        if ticket == "native":
            self.comm.halo_finish(stream=self.torch.cuda.current_stream().cuda_stream)
            self.exchanges += 1
            return
        works, unpack, stage = ticket
        for w in works:
            w.wait()
        for buf, parts in unpack:
            off = 0
            if stage:
                buf = buf.to(self.device)
            for p in parts:
                p.copy_(buf[off: off + p.numel()].view(p.shape))
                off += p.numel()
        self.exchanges += 1

    def _p2p_ready(self, nbytes: int):
        """The mailbox block of this rank, mapped by its neighbours (collective: every rank calls it with the same size)."""
        if self.p2p is not None and nbytes <= self._p2p_cap:
            return
        t, dist = self.torch, self.dist
        if self.p2p is not None:
            t.cuda.synchronize()
            if self.world > 1:
                dist.barrier(group=self.group)     # nobody may still be writing into a block that is about to go
            self.p2p.close()
        self.p2p = _lib.P2P(self.device.index, nbytes)
        self._p2p_cap = nbytes
        mine = self.p2p.export()
        if self.world > 1:
            handles = [None] * self.world
            dist.all_gather_object(handles, mine, group=self.group)
        else:
            handles = [mine]
        pick = lambda r: None if r is None else handles[r]
        self.p2p.connect(pick(self.south), pick(self.north), south_is_self=(self.south == self.rank),
                         north_is_self=(self.north == self.rank))
        if self.world > 1:
            dist.barrier(group=self.group)         # every block is mapped before the first post

    def p2p_timed_out(self) -> bool:
        """Has a peer-to-peer exchange of this rank failed (a wait that timed out, or a neighbour's abort)?  Definitive for
        everything enqueued before the last synchronisation; reads a mapped host word, no device call."""
        return self.p2p is not None and self.p2p.timed_out()

    def check_exchange(self):
        """Raise if a peer-to-peer exchange of this rank has failed.  Called at the start of every application, by ``synchronize``,
        ``gather_to_global`` and ``collect_kernel_times``: a failed exchange is never silent (its results are NaN on the device,
        gcmf_p2p_guard, and the host raises here)."""
        if self.p2p is not None:
            why = self.p2p.failed()
            if why:
                raise _lib.GcmfError(_lib.ERR_P2P_TIMEOUT,
                                     "the peer-to-peer halo exchange of rank %d failed (%s): ghost rows and results since then are NaN; "
                                     "rebuild the SlabFilter on every rank" % (
                                         self.rank, "a wait for a neighbour ran into GCMF_P2P_TIMEOUT_MS -- a rank out of step or gone"
                                         if why == 1 else "a neighbour's wait failed and it raised abort"))

    def synchronize(self):
        """Wait for everything this rank has enqueued and raise if one of its halo exchanges failed."""
        self.torch.cuda.synchronize(self.device)
        self.check_exchange()

    def _p2p_guard(self, outs):
        """After the last launch of an application (Python choreography): NaN over the result if an exchange has failed."""
        if self.p2p is not None and self.multi:
            stream = self.torch.cuda.current_stream().cuda_stream
            for o in outs:
                self.p2p.guard(o.data_ptr(), o.numel() * o.element_size() // 16 * 16, stream=stream)

    def collect_kernel_times(self):
        """Fold the launch events recorded since the last call into ``kernel_ms`` / ``kernel_launches`` (synchronises)."""
        if self._pending_events:
            self.torch.cuda.synchronize()
            self.kernel_ms += sum(a.elapsed_time(b) for a, b, _ in self._pending_events)
            self.kernel_launches += sum(n for _, _, n in self._pending_events)
            self.kernel_apps += len(self._pending_events)
            self._pending_events = []
            self.check_exchange()

    def _apply_backward_native(self, cut, st, p, nbatch):
        """The same application in ONE call into libgcmf (gcmf_slab_apply_backward: the choreography below in C++, exchanges through the
        library's own gcmf_comm / gcmf_p2p): ~30 us of host time instead of 250-350."""
        t = self.torch
        X, O = st["X"], st["O"]
        if self.exchange_kind == "p2p" and self.multi:
            self._p2p_ready(2 * nbatch * self.halo * self.nx * X.element_size())
        if self.time_kernels:
            e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            e0.record()
        call = lambda resident: self.engine.plan.slab_apply_backward(
            self.comm if self.exchange_kind == "native" else None, self.p2p if self.exchange_kind == "p2p" else None,
            self.south, self.north, p, self.c, cut, X[0].data_ptr(), [st[k][0].data_ptr() for k in "ABCD"], O[0].data_ptr(), nbatch,
            self.halo, self.overlap, stream=t.cuda.current_stream().cuda_stream, resident=resident)
        # (ranks sharing ONE GPU: never the on-chip kernel, see __init__; its slab path is exercised on a ring of one rank)
        call(self.resident and not self._shared_gpu)
        if self.multi:
            self.exchanges += 1 + sum(1 for _ in self._exchange_points(cut))
        if self.time_kernels:
            e1.record()
            self._pending_events.append((e0, e1, len(cut)))
        fo, ro = self.first_owned, self.rows_owned
        return [O[0][:, fo: fo + ro, :]]

    def _vector_backward_ok(self, nbatch: int) -> bool:
        """Vector kinds (C-grid, B-grid) with a library-issued exchange: the backward application in ONE call into libgcmf
        (gcmf_slab_apply_backward_vec), like the scalar kinds -- if every rank's plan has the backward kernel for this batch size."""
        if (self.ncomp != 2 or self.evaluation == "reference" or not self.native_driver or not isinstance(self.engine, HipSlabEngine)
                or (self.multi and self.exchange_kind not in ("native", "p2p"))):
            return False
        ok = self._vec_backward.get(nbatch)
        if ok is None:
            ok = 1 if self.engine.plan.slab_backward_vec_supported(nbatch, self.halo if self.multi else 0) else 0
            if self.world > 1 and self.dist.is_initialized():
                flag = self.torch.tensor([ok], dtype=self.torch.int32,
                                         device=self.device if self.dist.get_backend(self.group) == "nccl" else self.torch.device("cpu"))
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
                ok = int(flag.item())
            self._vec_backward[nbatch] = ok
        return bool(ok)

    def _apply_backward_vec_native(self, st, p, nbatch):
        t = self.torch
        X, O = st["X"], st["O"]
        if self.exchange_kind == "p2p" and self.multi:
            self._p2p_ready(4 * nbatch * self.halo * self.nx * X.element_size())     # (two states of two components)
        if self.time_kernels:
            e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            e0.record()
        pool = [st[k][comp].data_ptr() for k in "ABCD" for comp in range(2)]
        self.engine.plan.slab_apply_backward_vec(
            self.comm if (self.exchange_kind == "native" and self.multi) else None, self.p2p if (self.exchange_kind == "p2p" and self.multi) else None,
            self.south, self.north, p, self.c, [X[0].data_ptr(), X[1].data_ptr()], pool, [O[0].data_ptr(), O[1].data_ptr()], nbatch, self.halo,
            stream=t.cuda.current_stream().cuda_stream)
        nlaunch, left, valid, nex = 0, self.n_steps, self.halo, 1
        while left > 0:      # (the bookkeeping of the C++ driver, for the counters)
            S = next((cnd for cnd in (4, 3, 2) if cnd <= left and left - cnd != 1), left)
            if self.multi and valid < S:
                nex, valid = nex + 1, self.halo
            valid -= S
            left -= S
            nlaunch += 1
        if self.multi:
            self.exchanges += nex
        if self.time_kernels:
            e1.record()
            self._pending_events.append((e0, e1, nlaunch))
        fo, ro = self.first_owned, self.rows_owned
        return [O[k][:, fo: fo + ro, :] for k in range(2)]

    def _exchange_points(self, cut):
        """Launches of a backward application that are followed (or preceded) by an exchange of the state: the bookkeeping of
        _apply_backward without the launches (for the `exchanges` counter)."""
        s, valid = self.halo, self.halo
        for q, S in enumerate(cut):
            if valid < S:
                yield q
                valid = s
            v_out = valid - S
            nxt = cut[q + 1] if q + 1 < len(cut) else 0
            if self.overlap and q + 1 < len(cut) and v_out < nxt and self.rows_owned >= 4 * s:
                yield q
                v_out = s
            valid = v_out

    def _apply_backward(self, cut, st, p, nbatch):
        """The backward (Clenshaw) evaluation libgcmf uses on one GPU for this plan, on the slab: the state is
        (b_{k+1}, b_{k+2}), the constant input keeps its ghost rows from ONE exchange at the start, a launch of S levels uses
        up S ghost rows of the state, the state's ghost rows are refreshed when the next launch needs more than are left."""
        t = self.torch
        X, O = st["X"], st["O"]
        pool = [st["A"], st["B"], st["C"], st["D"]]
        fo, ro, s, n = self.first_owned, self.rows_owned, self.halo, self.n_steps
        comps = lambda buf: [buf[k] for k in range(self.ncomp)]
        if self.multi:
            self._exchange([X])          # f's ghost rows: the first launch forms b_n = p_n f on them, later ones read f on the rows they compute
        u = v = None
        valid, lvl, nlaunch = (s if self.multi else 0), 1, 0
        if self.time_kernels:
            e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            e0.record()
        for q, S in enumerate(cut):
            if self.multi and valid < S:
                self._exchange([u, v])
                valid = s
            free = [b for b in pool if b is not u and b is not v]
            v_out = (valid - S) if self.multi else 0
            lo = fo - (v_out if self.gs else 0)
            hi = fo + ro + (v_out if self.gn else 0)
            last = (q == len(cut) - 1)
            mode = _lib.STEP_CLENSHAW | (_lib.STEP_FIRST if q == 0 else 0) | (_lib.STEP_LAST if last else 0)
            pk = p[n - lvl - S + 1: n - lvl + 1][::-1]       # level l uses p[n - l]
            args = (None if u is None else comps(u), None if v is None else comps(v), comps(free[0]), comps(free[1]), comps(X),
                    comps(O), pk, p[n], self.c, mode, nbatch)
            nxt = cut[q + 1] if not last else 0
            overlap = (self.overlap and self.multi and not last and v_out < nxt and ro >= 4 * s)
            if overlap:
                # the next launch needs fresh ghost rows: advance the rows the neighbours need first, post the exchange of the
                # NEW state, and let the interior rows run while the messages are in flight
                # The messages carry the OWNED edge rows [fo, fo+s) / [fo+ro-s, fo+ro): with ghost rows left over (v_out > 0)
                # the edge launches start v_out rows outside them and must reach to their inner end
                ilo = fo + s if self.gs else lo
                ihi = fo + ro - s if self.gn else hi
                if self.gs:
                    self.engine.multi(*args, lo, ilo)
                if self.gn:
                    self.engine.multi(*args, ihi, hi)
                pending = self._exchange_start([free[0], free[1]])
                self.engine.multi(*args, ilo, ihi)
                self._exchange_finish(pending)
                v_out = s
            else:
                self.engine.multi(*args, lo, hi)
            u, v = free[0], free[1]
            valid = v_out
            lvl += S
            nlaunch += 1
        if self.engine.has_land():
            self.engine.land_fix(p, self.c, comps(X), comps(O), nbatch)
        self._p2p_guard([O])
        if self.time_kernels:
            e1.record()
            self._pending_events.append((e0, e1, nlaunch))
        return [O[k][:, fo: fo + ro, :] for k in range(self.ncomp)]

    # -- the filter ----------------------------------------------------------------------------
    MULTI_DEPTHS = (8, 7, 6, 5, 4, 3, 2)  # scalar kinds support all of them, the vector kinds 4 / 3 / 2

    def _cut_for(self, nbatch):
        """The launch depths of a backward application of `nbatch` fields: nines where they were agreed on (__init__) and pay -- a batch, or a
        lone field on slabs of 700 rows and more (measured, experiments/scripts/slab_nines_ab.py: one field 600 rows 301 against 295 us,
        800 rows 358 against 372, 1200 rows 490 against 543; four fields 8-9 % at every height).  Every rank takes the same branch: the
        batch is the same everywhere and the tallest slab of the run was reduced over the ranks."""
        use9 = bool(self._cut9) and (nbatch >= 2 or getattr(self, "_rows_owned_max", 0) >= int(os.environ.get("GCMF_SLAB_NINES_MIN_ROWS", "700")))
        plan = getattr(self.engine, "plan", None)
        if hasattr(plan, "set_option") and self._cut9:
            plan.set_option("slab_nines", 1 if use9 else 0)     # (every time: plans are cached and shared between SlabFilters)
        return self._cut9 if use9 else self.backward_cut

    def apply_local(self, local: Sequence):
        """Filter this rank's rows.  `local`: ncomp tensors (nbatch, rows_owned, nx) on the device.  Returns
        ncomp float64 tensors of the same shape (views into an internal buffer, valid until the next call).

        Between two halo exchanges the recurrence advances `halo` steps; scalar grids and batched C-grid fields do
        that with the temporally blocked kernels (up to 8 / 4 steps per HBM pass, consuming one ghost row per
        step), the rest with single steps on a row range that shrinks by one per step."""
        t = self.torch
        assert len(local) == self.ncomp
        self.check_exchange()
        nbatch = int(local[0].shape[0])
        st = self._state(nbatch)
        X, F, O = st["X"], st["F"], st["O"]
        Fn = st["F2"]  # fbar ping-pongs between two planes: the static-ring kernels redo a strip from its inputs when they
        #                meet a NaN / inf, so a blocked launch must not accumulate fbar in place (gcmf_ring_impl.hpp)
        pool = [st["A"], st["B"], st["C"], st["D"]]
        fo, ro, s = self.first_owned, self.rows_owned, self.halo
        for k in range(self.ncomp):
            X[k, :, fo: fo + ro, :].copy_(local[k].to(self.tdtype))
        comps = lambda buf: [buf[k] for k in range(self.ncomp)]
        p = np.asarray(self.spec.p, dtype=np.float64)
        n = self.n_steps
        can_multi = hasattr(self.engine, "multi") and self.multi_depth >= 2
        prepared = False
        keep_land_out = can_multi and hasattr(self.engine, "has_land") and self.engine.has_land()
        land_zeroed = False
        if self.backward_cut:
            cut = self._cut_for(nbatch)
            if self.native_driver and self.ncomp == 1 and isinstance(self.engine, HipSlabEngine) and self.exchange_kind in ("native", "p2p"):
                return self._apply_backward_native(cut, st, p, nbatch)
            return self._apply_backward(cut, st, p, nbatch)
        if self._vector_backward_ok(nbatch):
            return self._apply_backward_vec_native(st, p, nbatch)
        u, v = X, None          # T_{k-1}, T_{k-2}
        valid = 0               # ghost rows of u (and at least valid-1 of v) that are up to date
        events = []
        nlaunch = 0
        k = 1
        while k <= n:
            left = n - k + 1
            if self.multi and valid == 0:
                self._exchange([u] if v is None else [u, v])
                valid = s
                if land_zeroed:
                    # a neighbour that overlapped its exchange with the first launch sent its rows before it zeroed
                    # them (ranks with unequal row counts decide differently): the ghost rows must honour LAND_ZERO too
                    self.engine.zero_land(comps(u), comps(v), nbatch)
            budget = min(left, valid if self.multi else left)
            S = 1
            if can_multi:
                for cand in self.MULTI_DEPTHS:
                    if budget - cand == 1 and left == budget:
                        continue  # do not strand a lone single step at the very end
                    if cand <= budget and cand <= self.multi_depth and self.engine.multi_supported(cand, nbatch):
                        S = cand
                        break
            free = [b for b in pool if b is not u and b is not v]
            v_out = (valid - S) if self.multi else 0
            lo = fo - (v_out if self.gs else 0)
            hi = fo + ro + (v_out if self.gn else 0)
            is_last = (k + S - 1 == n)
            mode = (_lib.STEP_FIRST if k == 1 else 0) | (_lib.STEP_LAST if is_last else 0) | (_lib.STEP_LAND_ZERO if land_zeroed else 0)
            if k == 1 and keep_land_out and S >= 2 and not is_last:
                # land_fix below restores the isolated cells: the first launch may drop them while it loads the field
                # (k_ring's first-launch variant; the general kernels ignore the flag and zero_land does it after them)
                mode |= _lib.STEP_LAND_FIXED
            if self.time_kernels and k == 1:  # one pair of events per application (per-launch pairs cost 4 %)
                e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                e0.record()
            if S >= 2:
                args = (comps(u), None if v is None else comps(v), comps(free[0]), comps(free[1]), comps(F),
                        comps(O) if is_last else comps(Fn), p[k: k + S], p[0], self.c, mode, nbatch)
                overlap = (self.overlap and self.multi and v_out == 0 and not is_last and ro >= 4 * s
                           and self.engine.multi_supported(S, nbatch))
                if overlap:
                    # this launch uses up the ghost zone: advance the rows the neighbours need first, post the halo
                    # exchange of the NEW states, and let the interior rows run while the messages are in flight
                    if self.gs:
                        self.engine.multi(*args, lo, lo + s)
                    if self.gn:
                        self.engine.multi(*args, hi - s, hi)
                    pending = self._exchange_start([free[0], free[1]])
                    self.engine.multi(*args, lo + (s if self.gs else 0), hi - (s if self.gn else 0))
                    self._exchange_finish(pending)
                    v_out = s
                else:
                    self.engine.multi(*args, lo, hi)
                u, v = free[0], free[1]
                F, Fn = Fn, F
                if k == 1 and not is_last and keep_land_out:
                    # flux kinds: land (cells with four closed faces) leaves the state here; its own polynomial is
                    # written into the result by land_fix below (what gcmf_apply does internally)
                    self.engine.zero_land(comps(u), comps(v), nbatch)
                    land_zeroed = True
            else:
                if k == 1 and self.area_weighted and not prepared:
                    # T_0 = field * area on every valid row (the blocked kernel fuses this, single steps do not)
                    self.engine.prepare(comps(X), comps(X), nbatch, fo - (valid if self.gs else 0),
                                        fo + ro + (valid if self.gn else 0))
                    prepared = True
                self.engine.step(comps(u), None if v is None else comps(v), comps(F), comps(free[0]),
                                 comps(O) if is_last else comps(F), p[0] if k == 1 else p[k], p[1], self.c, mode,
                                 nbatch, lo, hi)
                u, v = free[0], u
            nlaunch += 1
            if self.time_kernels and is_last:
                e1.record()
                events.append((e0, e1, nlaunch))
            valid = v_out
            k += S
        if land_zeroed:
            self.engine.land_fix(p, self.c, comps(X), comps(O), nbatch)
        self._p2p_guard([O])
        if events:  # read back later (collect_kernel_times): a synchronisation here would serialise consecutive calls
            self._pending_events.extend(events)
        return [O[k][:, fo: fo + ro, :] for k in range(self.ncomp)]
