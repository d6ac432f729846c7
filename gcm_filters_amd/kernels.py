"""Laplacian kernel registry -- the host-side mirror of ``gcm_filters/kernels.py``.

Same public surface as the reference module (``GridType``, ``ALL_KERNELS``, ``required_grid_vars``, the
eleven Laplacian classes with ``required_grid_args() / prepare / __call__ / finalize / is_dimensional``),
but every class is a thin handle on a device-resident ``gcmf_plan`` (include/gcmf.h): constructing one
uploads the grid planes, runs the reference's validation and folds masks / kappas / metric ratios into
coefficient planes in HBM; calling it launches the HIP stencil.  There is no numpy fallback.
"""
from __future__ import annotations

import enum
import os
import threading
import weakref
from collections import OrderedDict
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib

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
    if hasattr(x, "__cuda_array_interface__") or (hasattr(x, "__dlpack__") and not hasattr(x, "dims")):
        # other device-array libraries (the reference's cupy path, gpu_compat.py:5-10): zero-copy via DLPack
        import torch
        return torch.from_dlpack(x)
    data = getattr(x, "data", None)
    if data is not None and hasattr(x, "dims"):
        return data if (_is_torch(data) or isinstance(data, np.ndarray)) else np.asarray(data)
    return np.asarray(x)


def _on_gpu(x) -> bool:
    return _is_torch(x) and x.is_cuda


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


def _fingerprint(a) -> Tuple:
    """Cheap identity of a grid plane: buffer address, layout, and a strided sample of its values (256 of them:
    the check runs on every filter call, 8 planes of a 2400x3600 grid cost ~40 us; 2048 samples cost 280 us)."""
    if _is_torch(a):
        return ("t", a.data_ptr(), tuple(a.shape), tuple(a.stride()), str(a.dtype), a._version, str(a.device))
    n = a.size
    flat = a.reshape(-1) if a.flags.c_contiguous else a.ravel()
    sample = flat[:: max(1, n // 256)]
    return ("n", a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str, hash(sample.tobytes()))


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

    def __init__(self, capacity: int = 6):
        self.capacity = capacity
        self._d: "OrderedDict[Tuple, _lib.Plan]" = OrderedDict()
        self._lock = threading.Lock()

    def get(self, key, factory):
        with self._lock:
            p = self._d.get(key)
            if p is not None:
                self._d.move_to_end(key)
                return p
        p = factory()
        with self._lock:
            self._d[key] = p
            while len(self._d) > self.capacity:
                _, old = self._d.popitem(last=False)
                old.close()
        return p

    def clear(self):
        with self._lock:
            for p in self._d.values():
                p.close()
            self._d.clear()


PLAN_CACHE = _PlanCache()


def clear_plan_cache():
    """Drop every cached device plan (frees their HBM)."""
    PLAN_CACHE.clear()


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
        for a in self._planes:
            if a.ndim != 2:
                raise NotImplementedError(
                    "grid variables must be 2-D (y, x) planes shared by all batch entries; "
                    f"got a grid variable of shape {tuple(a.shape)}")
        self._fp = tuple(_fingerprint(a) for a in self._planes)
        if self._planes:  # validation happens at construction, like the reference's __post_init__
            self._plan(compute_dtype(self._planes), tuple(self._planes[0].shape))

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
        key = (self.GRID_TYPE.value, dtype, tuple(shape), device, self._fp)

        def factory():
            try:
                if on_gpu:
                    import torch
                    tdt = torch.float64 if dtype == _lib.F64 else torch.float32
                    keep = [a.to(tdt).contiguous() for a in self._planes]
                    torch.cuda.synchronize(device)
                    plan = _lib.Plan(self.GRID_TYPE.value, dtype, shape[0], shape[1], [t.data_ptr() for t in keep],
                                     device=device, planes_on_device=True)
                    del keep
                    return plan
                host = [a.detach().cpu().numpy() if _is_torch(a) else a for a in self._planes]
                return _lib.Plan(self.GRID_TYPE.value, dtype, shape[0], shape[1], host, device=device)
            except _lib.GcmfError as e:
                raise _translate(e) from None

        return PLAN_CACHE.get(key, factory)

    def _run(self, fields: Sequence, spec=None, out_f32: bool = False):
        """Shared driver of __call__ (spec None: one Laplacian) and of filter_func (spec: whole polynomial)."""
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
            ins = [f.to(tdt).contiguous() for f in fields]
            outs = [torch.empty(shape, dtype=torch.float64 if out_np == np.float64 else torch.float32,
                                device=fields[0].device) for _ in fields]
            if nbatch:
                with torch.cuda.device(dev):
                    stream = torch.cuda.current_stream().cuda_stream
                    self._call(plan, spec, [t.data_ptr() for t in ins], [t.data_ptr() for t in outs], nbatch,
                               True, out_f32, stream)
                    # inputs converted above may be temporaries: keep them alive until the stream is done
                    for t in ins:
                        t.record_stream(torch.cuda.current_stream())
            return outs
        plan = self._plan(dtype, (ny, nx))
        host = [f.detach().cpu().numpy() if _is_torch(f) else np.asarray(f) for f in fields]
        ins = [np.ascontiguousarray(f, dtype=_lib.np_dtype(dtype)) for f in host]
        outs = [_host_output(shape, out_np) for _ in fields]
        if nbatch:
            self._call(plan, spec, [a.ctypes.data for a in ins], [a.ctypes.data for a in outs], nbatch, False,
                       out_f32, 0)
        return outs

    def _call(self, plan, spec, ins, outs, nbatch, device_ptrs, out_f32, stream):
        try:
            if spec is None:
                plan.laplacian(ins, outs, nbatch, device_ptrs=device_ptrs, stream=stream)
            else:
                # shift of the spectrum to [-1, 1]: reference filter.py:170-173
                c = 2 / spec.s_max if self.is_dimensional else 2 / (spec.s_max * spec.dx_min_sq)
                plan.apply(np.asarray(spec.p, dtype=np.float64), c, ins, outs, nbatch, device_ptrs=device_ptrs,
                           out_f32=out_f32, stream=stream)
        except _lib.GcmfError as e:
            raise _translate(e) from None


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
