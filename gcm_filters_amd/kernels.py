"""Laplacian kernel registry -- the host-side mirror of ``gcm_filters/kernels.py``.

Same public surface as the reference module (``GridType``, ``ALL_KERNELS``, ``required_grid_vars``, the
eleven Laplacian classes with ``required_grid_args() / prepare / __call__ / finalize / is_dimensional``),
but every class is a thin handle on a device-resident ``gcmf_plan`` (include/gcmf.h): constructing one
uploads the grid planes, runs the reference's validation and folds masks / kappas / metric ratios into
coefficient planes in HBM; calling it launches the HIP stencil.  There is no numpy fallback.
"""
from __future__ import annotations

import enum
import contextlib
import contextvars
import os
import threading
import warnings
import weakref
from collections import OrderedDict
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib, host_blocks

_BLOCKS_LOCK = threading.Lock()

GridType = enum.Enum(
    "GridType",
    [
        "REGULAR",
        "REGULAR_AREA_WEIGHTED",
        "REGULAR_WITH_LAND",
        "REGULAR_WITH_LAND_AREA_WEIGHTED",
        "IRREGULAR_WITH_LAND",
        "MOM5U",
        "MOM5T",
        "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED",
        "TRIPOLAR_POP_WITH_LAND",
        "VECTOR_C_GRID",
        "VECTOR_B_GRID",
    ],
)  # values 1..11 == gcmf_grid_type == reference kernels.py:13-28

ALL_KERNELS: Dict[GridType, Any] = {}

ArrayType = np.ndarray


# ------------------------------------------------------------------------------------------------
# array plumbing: numpy (host) and torch (host or MI355X-resident) arrays
# ------------------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _unwrap(x):
    """xarray.DataArray / Variable -> its array; everything else unchanged."""
    if _is_torch(x) or isinstance(x, np.ndarray):
        return x
    if hasattr(x, "__dlpack__") and not hasattr(x, "dims"):
        # other device-array libraries (the reference's cupy path, gpu_compat.py:5-10): zero-copy via DLPack
        import torch
        return torch.from_dlpack(x)
    if hasattr(x, "__cuda_array_interface__"):
        import torch
        return torch.as_tensor(x, device="cuda")
    data = getattr(x, "data", None)
    if data is not None and hasattr(x, "dims"):
        return data if (_is_torch(data) or isinstance(data, np.ndarray)) else np.asarray(data)
    return np.asarray(x)


def _on_gpu(x) -> bool:
    return _is_torch(x) and x.is_cuda


def _foreign_device_array(x) -> bool:
    """A device array of another library (cupy, a DLPack / `__cuda_array_interface__` producer): neither numpy, torch nor xarray."""
    return (not _is_torch(x) and not isinstance(x, np.ndarray) and not hasattr(x, "dims")
            and (hasattr(x, "__dlpack__") or hasattr(x, "__cuda_array_interface__")))


def _same_kind(out, like):
    """Hand a result (a torch tensor on the GPU) back as the kind of array the caller passed in -- the reference returns cupy arrays for
    cupy input (gpu_compat.py:5-10: `get_array_module`).  Zero-copy through the producer's own DLPack consumer: its array-API namespace
    (`like.__array_namespace__().from_dlpack`), its class (`type(like).from_dlpack`) or its module (`cupy.from_dlpack`), whichever it
    has; a producer with none of them gets the torch tensor (documented in README: the buffer is shared, `torch.Tensor` is itself a
    DLPack / `__cuda_array_interface__` producer)."""
    if not _foreign_device_array(like) or not _is_torch(out):
        return out
    import sys
    cands = []
    ns = getattr(like, "__array_namespace__", None)
    if callable(ns):
        try:
            cands.append(getattr(ns(), "from_dlpack", None))
        except Exception:   # noqa: BLE001
            pass
    cands.append(getattr(type(like), "from_dlpack", None))
    mod = sys.modules.get(type(like).__module__)
    top = sys.modules.get(type(like).__module__.split(".")[0])
    for m in (mod, top):
        cands.append(getattr(m, "from_dlpack", None) if m is not None else None)
    for fn in cands:
        if callable(fn):
            try:
                return fn(out)
            except Exception:   # noqa: BLE001  (e.g. a consumer that wants a capsule, not an object: try the next one)
                continue
    return out


def _np_dtype_of(x):
    if _is_torch(x):
        import torch
        return {torch.float32: np.dtype("f4"), torch.float64: np.dtype("f8"), torch.float16: np.dtype("f2"),
                torch.int64: np.dtype("i8"), torch.int32: np.dtype("i4"), torch.bool: np.dtype("?"),
                torch.uint8: np.dtype("u1"), torch.int8: np.dtype("i1"), torch.int16: np.dtype("i2"),
                torch.bfloat16: np.dtype("f4")}[x.dtype]
    return x.dtype


def compute_dtype(arrays: Sequence) -> int:
    """float32 only if everything is (at most) float32, else float64 -- numpy's promotion, f16 lifted to f32."""
    dts = [_np_dtype_of(a) for a in arrays]
    rt = np.result_type(*dts) if dts else np.dtype("f8")
    if rt.kind == "c":
        raise TypeError("complex fields are not supported")
    return _lib.F32 if (rt.kind == "f" and rt.itemsize <= 4) else _lib.F64


# How a cached device plan is tied to the host grid planes it was folded from (the reference rebuilds and re-validates its
# Laplacian on every call, gcm_filters/filter.py:183; a plan lives in HBM and is reused while the planes are unchanged):
#   * the planes (and the arrays that own their buffers) are made READ-ONLY while a cached plan refers to them, so an
#     in-place edit raises instead of silently filtering with stale coefficients; writability comes back when the plan
#     leaves the cache (LRU eviction, clear_plan_cache()).  torch tensors carry a version counter instead;
#   * the cache key holds the buffer address, layout and a 256-value strided sample, and an entry is only reused while the
#     owner of the buffer is still alive (no false hit on a freed-and-reallocated buffer).  A writable view created BEFORE
#     the plan was cached can still edit the buffer behind its back: GCMF_PLAN_CACHE_VERIFY=full hashes the whole plane on
#     every call (~5 ms per 2400x3600 f64 plane), GCMF_PLAN_CACHE=0 rebuilds the plan on every call like the reference.
_VERIFY_FULL = os.environ.get("GCMF_PLAN_CACHE_VERIFY", "sample") == "full"
_CACHE_ON = os.environ.get("GCMF_PLAN_CACHE", "1") != "0"
# The same three behaviours per Filter (keyword `plan_cache`, filter.py) instead of per process:
#   "protect" (default)  cache the plan, write-protect the host grid planes it was folded from, sampled fingerprint per call
#   "verify"             cache the plan, leave the planes WRITABLE, hash every plane on every call: an in-place edit of wet_mask between
#                        two calls just works, as with the reference (a fresh plan is folded), for ~5 ms per 2400x3600 plane and call
#   "off"                a fresh plan per call, like the reference's per-call Laplacian (gcm_filters/filter.py:183)
PLAN_CACHE_MODES = ("protect", "verify", "off")
_MODE = contextvars.ContextVar("gcmf_plan_cache_mode", default=None)


def cache_mode() -> str:
    m = _MODE.get()
    if m is not None:
        return m
    return "protect" if _CACHE_ON else "off"   # (GCMF_PLAN_CACHE_VERIFY=full keeps the protection and hashes whole planes on top)


@contextlib.contextmanager
def plan_cache_mode(mode):
    """Run the enclosed Laplacian constructions / calls under `mode` (None: the process default from the environment)."""
    if mode is not None and mode not in PLAN_CACHE_MODES:
        raise ValueError(f"plan_cache must be one of {PLAN_CACHE_MODES}, not {mode!r}")
    tok = _MODE.set(mode)
    try:
        yield
    finally:
        _MODE.reset(tok)


def _content_hash(a) -> int:
    buf = np.ascontiguousarray(a)
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(buf)
    except ImportError:
        import zlib
        return zlib.crc32(buf)


def _fingerprint(a) -> Tuple:
    """Identity of a grid plane: buffer address, layout, and a strided sample of its values (256 of them: the check runs
    on every filter call, 8 planes of a 2400x3600 grid cost ~40 us) -- or a hash of all of it (GCMF_PLAN_CACHE_VERIFY=full)."""
    if _is_torch(a):
        return ("t", a.data_ptr(), tuple(a.shape), tuple(a.stride()), str(a.dtype), a._version, str(a.device))
    if _VERIFY_FULL or cache_mode() == "verify":
        return ("n", a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str, _content_hash(a))
    n = a.size
    flat = a.reshape(-1) if a.flags.c_contiguous else a.ravel()
    sample = flat[:: max(1, n // 256)]
    return ("n", a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str, hash(sample.tobytes()))


def _owner(a: np.ndarray) -> np.ndarray:
    """The ndarray at the bottom of a chain of views (it owns the buffer or wraps a foreign one)."""
    while isinstance(a.base, np.ndarray):
        a = a.base
    return a


class _HostLocks:
    """Reference-counted write protection of host arrays that cached plans were folded from."""

    def __init__(self):
        self._mu = threading.Lock()
        self._held: Dict[int, list] = {}   # id(array) -> [weakref, count]

    def acquire(self, arrays) -> list:
        """Make `arrays` and the owners of their buffers read-only; returns the tickets for ``release``."""
        tickets = []
        with self._mu:
            for a in arrays:
                if not isinstance(a, np.ndarray):
                    continue
                for x in {id(a): a, id(_owner(a)): _owner(a)}.values():
                    ent = self._held.get(id(x))
                    if ent is not None and ent[0]() is x:
                        ent[1] += 1
                        tickets.append(ent[0])
                    elif x.flags.writeable:
                        x.flags.writeable = False
                        ref = weakref.ref(x)
                        self._held[id(x)] = [ref, 1]
                        tickets.append(ref)
        return tickets

    def release(self, tickets):
        with self._mu:
            for ref in reversed(tickets):   # owners of the buffers first: a view cannot be made writable before its base
                x = ref()
                if x is None:
                    continue
                ent = self._held.get(id(x))
                if ent is None or ent[0] is not ref:
                    continue
                ent[1] -= 1
                if ent[1] > 0:
                    continue
                del self._held[id(x)]
                try:
                    x.flags.writeable = True
                except ValueError:
                    # a view whose base is still protected on behalf of another plan: it gets its flag back with the base
                    own = self._held.get(id(_owner(x)))
                    if own is not None:
                        own.append(ref)
                    continue
                for late in ent[2:]:
                    v = late()
                    if v is not None:
                        try:
                            v.flags.writeable = True
                        except ValueError:
                            pass
            for k in [k for k, e in self._held.items() if e[0]() is None]:
                del self._held[k]


_HOST_LOCKS = _HostLocks()


# Host outputs: a fresh pageable numpy array costs more than the transfer that fills it (69 MB: 2.5 ms of first-touch
# page zeroing + 1.5 ms of munmap when it is dropped, against 1.2 ms of D2H).  Large outputs are therefore views of
# page-locked blocks from torch's caching host allocator: the DMA writes them directly and a dropped result goes back to
# the pool instead of to the kernel.  Page-locked memory is a shared resource: at most GCMF_PINNED_OUT_MAX bytes
# (default 4 GiB) are handed out at a time, beyond that (or without torch) results are ordinary numpy arrays.
_PINNED_MIN = 1 << 20
_PINNED_MAX = int(os.environ.get("GCMF_PINNED_OUT_MAX", str(4 << 30)))
_pinned_lock = threading.Lock()
_pinned_out = 0


def _pinned_release(nbytes):
    global _pinned_out
    with _pinned_lock:
        _pinned_out -= nbytes


def _host_output(shape, dtype):
    global _pinned_out
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes >= _PINNED_MIN and _PINNED_MAX > 0:
        with _pinned_lock:
            ok = _pinned_out + nbytes <= _PINNED_MAX
            if ok:
                _pinned_out += nbytes
        if ok:
            try:
                import torch
                t = torch.empty(tuple(shape), dtype=torch.float64 if np.dtype(dtype) == np.float64 else torch.float32,
                                pin_memory=True)
                a = t.numpy()  # keeps `t` alive as its base; the block returns to torch's pool with the array
                weakref.finalize(t, _pinned_release, nbytes)
                return a
            except Exception:
                _pinned_release(nbytes)
    return np.empty(shape, dtype=dtype)


class _PlanCache:
    """LRU of device plans.  The reference rebuilds its Laplacian object on every filter call
    (filter.py:183); here the equivalent state lives in HBM and is reused while the grid arrays are
    unchanged (same buffers, same sampled contents)."""

    def __init__(self, capacity: int = int(os.environ.get("GCMF_PLAN_CACHE_SIZE", "64"))):
        self.capacity = capacity
        self._d: "OrderedDict[Tuple, Tuple]" = OrderedDict()   # key -> (plan, owner weakrefs, lock tickets)
        self._lock = threading.Lock()
        self._building: Dict[Tuple, threading.Lock] = {}      # key -> lock held by the thread folding that plan

    @staticmethod
    def _drop(entry):
        """Explicit clear only: frees the HBM now.  Nobody may be running the plan."""
        plan, _, tickets = entry
        plan.close()
        _HOST_LOCKS.release(tickets)

    @staticmethod
    def _forget(entry):
        """Take an entry out of the cache WITHOUT destroying its plan: another (dask worker) thread may have been handed
        it a moment ago and be inside gcmf_apply.  The plan frees its HBM when its last holder lets go (Plan.__del__)."""
        _HOST_LOCKS.release(entry[2])

    def _lookup(self, key):
        ent = self._d.get(key)
        if ent is None:
            return None
        if all(r() is not None for r in ent[1]):
            self._d.move_to_end(key)
            return ent[0]
        self._forget(self._d.pop(key))   # a buffer was freed since: same address, unknown contents
        return None

    def get(self, key, factory, host_planes=()):
        """The plan for `key`, built by `factory` if absent.  `host_planes`: the numpy grid planes the plan is folded from
        (write-protected while the plan is cached; the entry dies with the owners of their buffers).  Threads that miss on
        the same key together build ONE plan: the first folds it, the others wait and are handed the same object."""
        with self._lock:
            p = self._lookup(key)
            if p is not None:
                return p
            gate = self._building.setdefault(key, threading.Lock())
        with gate:
            with self._lock:
                p = self._lookup(key)     # folded by the thread that held the gate before us
                if p is not None:
                    return p
            try:
                p = factory()
            except BaseException:
                with self._lock:
                    self._building.pop(key, None)
                raise
            nps = [a for a in host_planes if isinstance(a, np.ndarray)]
            owners = [weakref.ref(_owner(a)) for a in nps]
            tickets = _HOST_LOCKS.acquire(nps)
            with self._lock:
                old = self._d.pop(key, None)
                if old is not None:       # cannot happen under the gate; never destroy what another thread may hold
                    self._forget(old)
                self._d[key] = (p, owners, tickets)
                self._building.pop(key, None)
                while len(self._d) > self.capacity:
                    _, ev = self._d.popitem(last=False)
                    self._forget(ev)
        return p

    def clear(self):
        with self._lock:
            for ent in self._d.values():
                self._drop(ent)
            self._d.clear()


PLAN_CACHE = _PlanCache()


def clear_plan_cache():
    """Drop every cached device plan (frees their HBM) and give the grid arrays they were built from back their writability."""
    PLAN_CACHE.clear()


def kernels_cache_enabled() -> bool:
    """Plans are cached (GCMF_PLAN_CACHE != 0) and sampled verification is in use: callers may then reuse a Laplacian object
    for unchanged grid arrays (filter._LaplacianMemo); full verification re-hashes the planes on every call."""
    return cache_mode() == "protect" and not _VERIFY_FULL


_VALUE_ERRORS = {_lib.ERR_KAPPA_W_GT1, _lib.ERR_KAPPA_S_GT1, _lib.ERR_KAPPA_NONE_ONE}
_ASSERT_ERRORS = {_lib.ERR_WET_SOUTH_ROW, _lib.ERR_DXN_FOLD, _lib.ERR_DYN_FOLD}


def _translate(err: _lib.GcmfError):
    """libgcmf status -> the exception the reference raises (kernels.py:262-281, 458-459, 551-562)."""
    if err.status in _VALUE_ERRORS:
        return ValueError(err.message)
    if err.status in _ASSERT_ERRORS:
        return AssertionError(err.message)
    if err.status == _lib.ERR_INVALID_ARG and "even nx" in err.message:
        return ValueError(err.message)
    return err


def current_device() -> int:
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return 0


# ------------------------------------------------------------------------------------------------
# base classes (reference kernels.py:43-104)
# ------------------------------------------------------------------------------------------------
class _DeviceLaplacian:
    GRID_TYPE: GridType = None  # type: ignore
    _ARGS: Tuple[str, ...] = ()
    _NCOMP = 1
    is_dimensional = False

    def __init__(self, *args, **kwargs):
        self._skip_kappa_one = bool(kwargs.pop("_skip_kappa_one", False))
        names = self._ARGS
        if len(args) > len(names):
            raise TypeError(f"{type(self).__name__}() takes {len(names)} grid arguments but {len(args)} were given")
        vals = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in names:
                raise TypeError(f"{type(self).__name__}() got an unexpected keyword argument '{k}'")
            if k in vals:
                raise TypeError(f"{type(self).__name__}() got multiple values for argument '{k}'")
            vals[k] = v
        missing = [n for n in names if n not in vals]
        if missing:
            raise TypeError(f"{type(self).__name__}() missing required grid arguments: {missing}")
        for n in names:
            setattr(self, n, vals[n])
        self._planes = [_unwrap(vals[n]) for n in names]
        for n, a in zip(names, self._planes):
            if a.ndim < 2:
                raise ValueError(f"grid variable {n!r} needs at least two (y, x) dimensions, got shape {tuple(a.shape)}")
        self._levels = None
        if any(a.ndim > 2 for a in self._planes):
            self._init_levels()
            return
        self._fp = tuple(_fingerprint(a) for a in self._planes)
        if self._planes:  # validation happens at construction, like the reference's __post_init__
            self._plan(compute_dtype(self._planes), tuple(self._planes[0].shape))

    # -- grid variables with leading (level / time) dims ------------------------------------------
    def _init_levels(self):
        """Grid variables such as wet_mask(z, y, x) or kappa(z, y, x).  The reference's kernels roll along the last two
        axes only (kernels.py:113-121, 163-187, 297-315) and xarray.apply_ufunc broadcasts the remaining dims of field
        and grid variables against each other (filter.py:478-486), so every index of the grid variables' broadcast
        leading shape is an independent 2-D problem: one device plan per index, each batch entry filtered with its own."""
        leads = [tuple(a.shape[:-2]) for a in self._planes]
        self._glead = tuple(np.broadcast_shapes(*leads))
        if self.GRID_TYPE is GridType.IRREGULAR_WITH_LAND and not self._skip_kappa_one:
            # the reference tests the WHOLE kappa arrays (kernels.py:262-281); the per-plane plans repeat the > 1 test
            kw, ks = (np.asarray(x.detach().cpu() if _is_torch(x) else x) for x in (self.kappa_w, self.kappa_s))
            if not (np.any(np.isclose(kw, 1.0, atol=1e-5)) or np.any(np.isclose(ks, 1.0, atol=1e-5))) \
                    and not (np.any(kw > 1.0) or np.any(ks > 1.0)):
                raise ValueError("At least one place in the domain must have either kappa_w = 1 or kappa_s = 1. "
                                 "Otherwise the filter's scale will not be equal to filter_scale anywhere in the domain.")
        self._levels = {}
        for g in np.ndindex(*self._glead):
            sub = []
            for a, lead in zip(self._planes, leads):
                if a.ndim == 2:
                    sub.append(a)
                else:  # align the plane's own leading dims with the tail of the broadcast shape
                    gi = g[len(g) - len(lead):]
                    sub.append(a[tuple(0 if n == 1 else i for i, n in zip(gi, lead))])
            self._levels[g] = type(self)(*sub, _skip_kappa_one=True)

    def _run_levels(self, fields, spec, out_f32, forward=False, backward_f32=False):
        given = list(fields)
        fields = [_unwrap(f) for f in fields]
        shape = tuple(fields[0].shape)
        if len(shape) < 2:
            raise ValueError("fields need at least two (y, x) dimensions")
        core = shape[-2:]
        out_lead = tuple(np.broadcast_shapes(shape[:-2], self._glead))
        pad = len(out_lead) - len(self._glead)
        outs = None
        for g, lap in self._levels.items():
            idx = tuple([slice(None)] * pad + [slice(None) if n == 1 else i for i, n in zip(g, self._glead)])
            sub = []
            for f in fields:
                fb = f.expand(*out_lead, *core) if _is_torch(f) else np.broadcast_to(f, out_lead + core)
                sub.append(fb[idx])
            res = lap._run(sub, spec=spec, out_f32=out_f32, forward=forward, backward_f32=backward_f32)
            if outs is None:
                if _is_torch(res[0]):
                    import torch
                    outs = [torch.empty(out_lead + core, dtype=r.dtype, device=r.device) for r in res]
                else:
                    outs = [np.empty(out_lead + core, dtype=r.dtype) for r in res]
            for o, r in zip(outs, res):
                o[idx] = r
        return [_same_kind(o, g) for o, g in zip(outs, given)]

    # -- protocol ----------------------------------------------------------------------------
    @classmethod
    def required_grid_args(cls):
        return list(cls._ARGS)

    def _plan(self, dtype: int, shape: Tuple[int, int], device: Optional[int] = None) -> _lib.Plan:
        if self._planes and tuple(self._planes[0].shape) != tuple(shape):
            raise ValueError(f"field has spatial shape {tuple(shape)} but the grid variables have "
                             f"shape {tuple(self._planes[0].shape)}")
        on_gpu = bool(self._planes) and all(_on_gpu(a) for a in self._planes)
        if device is None:
            device = self._planes[0].device.index if on_gpu else current_device()
        key = (self.GRID_TYPE.value, dtype, tuple(shape), device, self._skip_kappa_one, self._fp)

        def factory():
            try:
                if on_gpu:
                    import torch
                    tdt = torch.float64 if dtype == _lib.F64 else torch.float32
                    keep = [a.to(tdt).contiguous() for a in self._planes]
                    torch.cuda.synchronize(device)
                    plan = _lib.Plan(self.GRID_TYPE.value, dtype, shape[0], shape[1], [t.data_ptr() for t in keep],
                                     device=device, planes_on_device=True, skip_kappa_one=self._skip_kappa_one)
                    del keep
                    return plan
                host = [a.detach().cpu().numpy() if _is_torch(a) else a for a in self._planes]
                return _lib.Plan(self.GRID_TYPE.value, dtype, shape[0], shape[1], host, device=device,
                                 skip_kappa_one=self._skip_kappa_one)
            except _lib.GcmfError as e:
                raise _translate(e) from None

        mode = cache_mode()
        if mode == "off":   # the reference's behaviour: a fresh Laplacian (validation + precompute) per call
            old = getattr(self, "_own_plan", None)
            if old is not None:
                old.close()
            self._own_plan = factory()
            return self._own_plan
        # ("verify": the key holds a hash of every plane -- nothing to protect, the caller's arrays stay writable)
        return PLAN_CACHE.get(key, factory, () if (on_gpu or mode == "verify") else self._planes)

    def _run(self, fields: Sequence, spec=None, out_f32: bool = False, forward: bool = False, backward_f32: bool = False):
        """Shared driver of __call__ (spec None: one Laplacian) and of filter_func (spec: whole polynomial).
        forward: evaluate the polynomial by the reference's forward recurrence with its accumulation scheme (f64 running sum
        also for f32 state) even where the library would evaluate it backwards (Filter(evaluation="reference")).
        backward_f32: evaluate it backwards also for f32 scalar / B-grid fields, whose default is the forward recurrence
        (Filter(evaluation="backward"): faster, all f32)."""
        if self._levels is not None:
            return self._run_levels(fields, spec, out_f32, forward, backward_f32)
        given = list(fields)
        fields = [_unwrap(f) for f in fields]
        shape = tuple(fields[0].shape)
        if len(shape) < 2:
            raise ValueError("fields need at least two (y, x) dimensions")
        for f in fields[1:]:
            if tuple(f.shape) != shape:
                raise ValueError("u and v must have the same shape")
        dtype = compute_dtype(list(fields) + list(self._planes))
        ny, nx = shape[-2:]
        nbatch = int(np.prod(shape[:-2], dtype=np.int64)) if len(shape) > 2 else 1
        gpu = all(_on_gpu(f) for f in fields)
        if spec is None:
            out_np = _lib.np_dtype(dtype)
        else:
            out_np = np.float32 if (dtype == _lib.F32 and out_f32) else np.float64
        if gpu:
            import torch
            dev = fields[0].device.index
            plan = self._plan(dtype, (ny, nx), dev)
            tdt = torch.float64 if dtype == _lib.F64 else torch.float32
            ins = [f if (f.dtype == tdt and f.is_contiguous()) else f.to(tdt).contiguous() for f in fields]
            outs = [torch.empty(shape, dtype=torch.float64 if out_np == np.float64 else torch.float32,
                                device=fields[0].device) for _ in fields]
            if nbatch:
                # (the device index is passed explicitly: torch.cuda.current_stream() without one costs ~10 us per call, and
                # libgcmf selects the plan's device itself -- no device context manager on this path)
                cur = torch.cuda.current_stream(dev)
                self._call(plan, spec, [t.data_ptr() for t in ins], [t.data_ptr() for t in outs], nbatch,
                           True, out_f32, cur.cuda_stream, forward, backward_f32)
                for t, f in zip(ins, fields):   # inputs converted above are temporaries: keep them alive until the stream is done
                    if t is not f:
                        t.record_stream(cur)
            return [_same_kind(o, g) for o, g in zip(outs, given)]   # (cupy in, cupy out -- as the reference's gpu_compat path)
        plan = self._plan(dtype, (ny, nx))
        host = [f.detach().cpu().numpy() if _is_torch(f) else np.asarray(f) for f in fields]
        ins = [np.ascontiguousarray(f, dtype=_lib.np_dtype(dtype)) for f in host]
        outs = [_host_output(shape, out_np) for _ in fields]
        if nbatch == 1 and spec is not None and self._NCOMP == 1 and ny * nx >= host_blocks.MIN_CELLS and not forward and not backward_f32:
            # one large host field: upload / recurrence / download overlapped by row blocks (host_blocks.py)
            pipe = self._row_blocks(plan, dtype, ny, nx, spec)
            if pipe is not None:
                c = 2 / spec.s_max if self.is_dimensional else 2 / (spec.s_max * spec.dx_min_sq)
                try:
                    pipe.apply(np.asarray(spec.p, dtype=np.float64), c, ins[0].reshape(ny, nx), outs[0].reshape(ny, nx),
                               bool(out_f32 and dtype == _lib.F32))
                except _lib.GcmfError as e:
                    raise _translate(e) from None
                _PATH.last = "host-row-blocks"
                return outs
        if nbatch:
            self._call(plan, spec, [a.ctypes.data for a in ins], [a.ctypes.data for a in outs], nbatch, False,
                       out_f32, 0, forward, backward_f32)
        return outs

    def _row_blocks(self, plan, dtype, ny, nx, spec):
        """The row-block pipeline of `plan` for this polynomial length, built once the plan has seen a few single-field
        host calls (it costs K more plans); None while it is not (yet) worth it."""
        n = int(spec.n_steps)
        with _BLOCKS_LOCK:
            st = plan.__dict__.setdefault("_host_blocks", {"calls": 0, "pipes": {}})
            st["calls"] += 1
            if n in st["pipes"]:
                return st["pipes"][n]
            if st["calls"] <= host_blocks.BUILD_AFTER_CALLS:
                return None
            nblocks = host_blocks.choose_blocks(ny, n)
            pipe = None
            if nblocks:
                host = [a.detach().cpu().numpy() if _is_torch(a) else np.asarray(a) for a in self._planes]
                try:
                    pipe = host_blocks.RowBlockPipeline(self.GRID_TYPE.value, dtype, ny, nx, host, plan.device, n, nblocks,
                                                        skip_kappa_one=self._skip_kappa_one, cut=plan.clenshaw_cut(n))
                except (_lib.GcmfError, RuntimeError, MemoryError):
                    pipe = None     # e.g. not enough device memory for the extra plans and state planes: stay on the plain path
                if pipe is not None and not pipe.ok:
                    pipe.close()
                    pipe = None
            while len(st["pipes"]) >= 2:    # polynomial lengths come and go with the filter scale: keep two
                old = st["pipes"].pop(next(iter(st["pipes"])))
                if old is not None:
                    old.close()
            st["pipes"][n] = pipe
            return pipe

    def _call(self, plan, spec, ins, outs, nbatch, device_ptrs, out_f32, stream, forward=False, backward_f32=False):
        try:
            if spec is None:
                plan.laplacian(ins, outs, nbatch, device_ptrs=device_ptrs, stream=stream)
            else:
                # shift of the spectrum to [-1, 1]: reference filter.py:170-173
                c = 2 / spec.s_max if self.is_dimensional else 2 / (spec.s_max * spec.dx_min_sq)
                plan.apply(np.asarray(spec.p, dtype=np.float64), c, ins, outs, nbatch, device_ptrs=device_ptrs,
                           out_f32=out_f32, stream=stream, forward=forward, backward_f32=backward_f32)
                _note_path(plan.last_path(), plan.device)
        except _lib.GcmfError as e:
            raise _translate(e) from None


# ---- which path ran ------------------------------------------------------------------------------------------
_PATH = threading.local()
_PATH_WARNED = set()


def _note_path(path, device):
    """Remember, per thread, which of the two bit-identical paths the last filter application took (Filter.last_path), and say ONCE
    per process when a grid that qualifies for the on-chip kernel is kept off it by another process's lock -- same bits, about half the
    speed for a small grid, and otherwise silent."""
    _PATH.last = path
    if path in ("resident-lock-busy", "resident-disabled") and path not in _PATH_WARNED:
        _PATH_WARNED.add(path)
        why = ("another process holds the on-chip (resident) lock of GPU %s" % device if path == "resident-lock-busy" else
               "an on-chip launch of this process timed out earlier on GPU %s (gcm_filters_amd._lib.resident_status)" % device)
        warnings.warn(f"gcm_filters_amd: {why}: small grids run the strip-marching launches instead -- the same bits, slower "
                      f"(Filter.last_path tells which path a call took).", RuntimeWarning, stacklevel=4)


def last_path():
    """The path the calling thread's last filter application took: "resident", "strips", "resident-lock-busy", "resident-disabled",
    "host-row-blocks" (a large host array streamed through row blocks), or None."""
    return getattr(_PATH, "last", None)


class BaseScalarLaplacian(_DeviceLaplacian):
    """Base class for scalar Laplacians (reference kernels.py:43-63)."""

    def prepare(self, field):
        return field

    def __call__(self, field):
        return self._run([field])[0]

    def finalize(self, field):
        return field


class BaseVectorLaplacian(_DeviceLaplacian):
    """Base class for vector Laplacians (reference kernels.py:66-86)."""

    _NCOMP = 2

    def prepare(self, ufield, vfield):
        return (ufield, vfield)

    def __call__(self, ufield, vfield):
        u, v = self._run([ufield, vfield])
        return (u, v)

    def finalize(self, ufield, vfield):
        return (ufield, vfield)


class AreaWeightedMixin:
    """Weight / de-weight by the cell area (reference kernels.py:89-104).  Inside the fused filter the two
    scalings are folded into the first / last kernel; these methods exist for protocol compatibility."""

    def prepare(self, field):
        return field * _unwrap(self.area)

    def finalize(self, field):
        return field / _unwrap(self.area)


def _register(grid_type: GridType, args: Tuple[str, ...], dimensional: bool, bases, name: str, doc: str):
    cls = type(name, bases, {"GRID_TYPE": grid_type, "_ARGS": tuple(args), "is_dimensional": dimensional,
                             "__doc__": doc, "__annotations__": {a: ArrayType for a in args}})
    ALL_KERNELS[grid_type] = cls
    return cls


RegularLaplacian = _register(
    GridType.REGULAR, (), False, (BaseScalarLaplacian,), "RegularLaplacian",
    "5-point Laplacian on a unit Cartesian grid, periodic in x and y (reference kernels.py:107-124).")
RegularLaplacianWithArea = _register(
    GridType.REGULAR_AREA_WEIGHTED, ("area",), False, (AreaWeightedMixin, BaseScalarLaplacian),
    "RegularLaplacianWithArea", "REGULAR applied to field*area, result divided by area (reference kernels.py:127-147).")
RegularLaplacianWithLandMask = _register(
    GridType.REGULAR_WITH_LAND, ("wet_mask",), False, (BaseScalarLaplacian,), "RegularLaplacianWithLandMask",
    "5-point Laplacian with no-flux land boundaries (reference kernels.py:150-190).")
RegularLaplacianWithLandMaskAndArea = _register(
    GridType.REGULAR_WITH_LAND_AREA_WEIGHTED, ("area", "wet_mask"), False, (AreaWeightedMixin, BaseScalarLaplacian),
    "RegularLaplacianWithLandMaskAndArea", "Land-masked REGULAR with area weighting (reference kernels.py:193-219).")
IrregularLaplacianWithLandMask = _register(
    GridType.IRREGULAR_WITH_LAND, ("wet_mask", "dxw", "dyw", "dxs", "dys", "area", "kappa_w", "kappa_s"), True,
    (BaseScalarLaplacian,), "IrregularLaplacianWithLandMask",
    "Flux-form Laplacian on a locally orthogonal grid with land and kappa_w/kappa_s (reference kernels.py:222-318).")
MOM5LaplacianU = _register(
    GridType.MOM5U, ("wet_mask", "dxt", "dyt", "dxu", "dyu", "area_u"), True, (BaseScalarLaplacian,),
    "MOM5LaplacianU", "MOM5 B-grid Laplacian at velocity points (reference kernels.py:321-375).")
MOM5LaplacianT = _register(
    GridType.MOM5T, ("wet_mask", "dxt", "dyt", "dxu", "dyu", "area_t"), True, (BaseScalarLaplacian,),
    "MOM5LaplacianT", "MOM5 B-grid Laplacian at tracer points (reference kernels.py:378-432).")
TripolarRegularLaplacianTpoint = _register(
    GridType.TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED, ("area", "wet_mask"), False,
    (AreaWeightedMixin, BaseScalarLaplacian), "TripolarRegularLaplacianTpoint",
    "Land-masked, area-weighted REGULAR with the tripole north fold (reference kernels.py:435-492).")
POPTripolarLaplacianTpoint = _register(
    GridType.TRIPOLAR_POP_WITH_LAND, ("wet_mask", "dxe", "dye", "dxn", "dyn", "tarea"), True, (BaseScalarLaplacian,),
    "POPTripolarLaplacianTpoint", "POP T-point flux-form Laplacian with the tripole fold (reference kernels.py:495-588).")
CgridVectorLaplacian = _register(
    GridType.VECTOR_C_GRID,
    ("wet_mask_t", "wet_mask_q", "dxT", "dyT", "dxCu", "dyCu", "dxCv", "dyCv", "dxBu", "dyBu", "area_u", "area_v",
     "kappa_iso", "kappa_aniso"), True, (BaseVectorLaplacian,), "CgridVectorLaplacian",
    "C-grid viscous vector Laplacian after Griffies & Hallberg 2000 (reference kernels.py:591-699).")
BgridVectorLaplacian = _register(
    GridType.VECTOR_B_GRID, ("DXU", "DYU", "HUS", "HUW", "HTE", "HTN", "UAREA", "TAREA"), True, (BaseVectorLaplacian,),
    "BgridVectorLaplacian", "POP B-grid vector Laplacian, periodic (reference kernels.py:702-840).")


def required_grid_vars(grid_type: GridType):
    """Names of the grid variables a grid type needs (reference kernels.py:843-858)."""
    return ALL_KERNELS[grid_type].required_grid_args()
