"""ctypes binding of libgcmf.so (the C ABI declared in include/gcmf.h).

There is no CPU fallback: if the HIP library cannot be loaded (or no MI355X is visible when a plan is
created) the calls raise.  ``Plan`` is the Python handle of a ``gcmf_plan`` -- the device-resident
counterpart of one reference ``Laplacian(**grid_vars)`` object (gcm_filters/kernels.py __post_init__s).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgcmf.so")

# status codes (include/gcmf.h)
OK, ERR_INVALID_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_P2P_TIMEOUT = 0, 1, 2, 3, 4, 5
ERR_KAPPA_W_GT1, ERR_KAPPA_S_GT1, ERR_KAPPA_NONE_ONE = 16, 17, 18
ERR_WET_SOUTH_ROW, ERR_DXN_FOLD, ERR_DYN_FOLD = 19, 20, 21
F32, F64 = 0, 1
DEVICE_PTRS, OUT_F32, FORWARD_RECURRENCE, NO_RESIDENT, BACKWARD_F32 = 0x1, 0x2, 0x4, 0x8, 0x10
STEP_FIRST, STEP_LAST, STEP_LAND_ZERO, STEP_LAND_FIXED, STEP_CLENSHAW = 0x1, 0x2, 0x4, 0x8, 0x10

EXPORTS = [
    "gcmf_plan_create", "gcmf_plan_destroy", "gcmf_grid_nplanes", "gcmf_grid_ncomp", "gcmf_grid_is_dimensional",
    "gcmf_grid_is_tripolar", "gcmf_plan_rows", "gcmf_apply", "gcmf_laplacian", "gcmf_cheb_step", "gcmf_prepare",
    "gcmf_last_timing", "gcmf_set_timing", "gcmf_set_tuning", "gcmf_set_option", "gcmf_last_error", "gcmf_version",
    "gcmf_multi_supported", "gcmf_cheb_multi", "gcmf_multi_supported_vec", "gcmf_cheb_multi_vec",
    "gcmf_has_land", "gcmf_zero_land", "gcmf_land_fix", "gcmf_last_kernel", "gcmf_last_kernel_timing", "gcmf_ring_fallbacks", "gcmf_clenshaw_cut",
    "gcmf_comm_unique_id", "gcmf_comm_create", "gcmf_comm_destroy", "gcmf_halo_start", "gcmf_halo_finish", "gcmf_comm_info",
    "gcmf_build_id", "gcmf_last_kernel_geometry",
    "gcmf_slab_apply_backward", "gcmf_slab_backward_vec_supported", "gcmf_slab_apply_backward_vec", "gcmf_resident_supported", "gcmf_resident_levels", "gcmf_p2p_create", "gcmf_p2p_export", "gcmf_p2p_connect", "gcmf_p2p_start", "gcmf_p2p_finish", "gcmf_p2p_status", "gcmf_p2p_destroy", "gcmf_p2p_guard", "gcmf_p2p_seq", "gcmf_p2p_set_timeout_ms", "gcmf_p2p_debug_skip_post",
    "gcmf_plan_last_path", "gcmf_resident_status",
]
PATH_NAMES = {0: None, 1: "resident", 2: "strips", 3: "resident-lock-busy", 4: "resident-disabled"}
RESIDENT_STATES = {0: "ok", 1: "lock-busy", 2: "disabled", 3: "off"}
PLAN_SELF_RING, PLAN_SKIP_KAPPA_ONE = 0x1, 0x2


class PlanDesc(C.Structure):
    _fields_ = [
        ("grid_type", C.c_int32), ("dtype", C.c_int32), ("ny", C.c_int64), ("nx", C.c_int64),
        ("row_begin", C.c_int64), ("row_end", C.c_int64), ("halo", C.c_int32), ("device", C.c_int32),
        ("planes_on_device", C.c_int32), ("flags", C.c_int32),
    ]


class GcmfError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libgcmf status {status}: {message}")
        self.status = status
        self.message = message


_lib = None
_lock = threading.Lock()


class StaleLibraryError(RuntimeError):
    """csrc/libgcmf.so was not built from the sources next to it and cannot be rebuilt here."""


def ensure_fresh_library() -> str:
    """Make sure csrc/libgcmf.so comes from the sources beside it: the binary carries the sha256 of csrc/* +
    include/gcmf.h + flags it was compiled from (gcmf_build_id()).  Missing or different: rebuild with hipcc when
    there is one, raise StaleLibraryError otherwise -- a stale binary is never loaded silently."""
    from . import _build
    want = _build.source_build_id()
    have = _build.binary_build_id(LIB_PATH)
    if have == want:
        return want
    try:
        _build.hipcc()
    except RuntimeError:
        what = "is missing" if have is None and not os.path.exists(LIB_PATH) else f"has build id {have}"
        raise StaleLibraryError(f"{LIB_PATH} {what}, the sources have {want}, and there is no hipcc to rebuild it") from None
    _build.build_library()
    have = _build.binary_build_id(LIB_PATH)
    if have != want:
        raise StaleLibraryError(f"rebuilt {LIB_PATH} carries build id {have}, the sources have {want}")
    return want


def load() -> C.CDLL:
    """dlopen libgcmf.so after checking that it was built from the sources beside it (rebuilt with hipcc if not)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        want = ensure_fresh_library()
        lib = C.CDLL(LIB_PATH)
        lib.gcmf_build_id.argtypes = []
        lib.gcmf_build_id.restype = C.c_char_p
        if lib.gcmf_build_id().decode() != want:
            raise StaleLibraryError(f"{LIB_PATH}: gcmf_build_id() disagrees with the marker in the file")
        vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
        lib.gcmf_plan_create.argtypes = [C.POINTER(PlanDesc), vpp, C.c_int, vpp]
        lib.gcmf_plan_create.restype = C.c_int
        lib.gcmf_plan_destroy.argtypes = [vp]
        lib.gcmf_plan_destroy.restype = None
        for name in ("gcmf_grid_nplanes", "gcmf_grid_ncomp", "gcmf_grid_is_dimensional", "gcmf_grid_is_tripolar"):
            getattr(lib, name).argtypes = [C.c_int]
            getattr(lib, name).restype = C.c_int
        lib.gcmf_plan_rows.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.gcmf_plan_rows.restype = C.c_int
        lib.gcmf_apply.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.c_double, vpp, vpp, C.c_int64, C.c_uint32, vp]
        lib.gcmf_apply.restype = C.c_int
        lib.gcmf_laplacian.argtypes = [vp, vpp, vpp, C.c_int64, C.c_uint32, vp]
        lib.gcmf_laplacian.restype = C.c_int
        lib.gcmf_cheb_step.argtypes = [vp, vpp, vpp, vpp, vpp, vpp, C.c_double, C.c_double, C.c_double, C.c_uint32,
                                       C.c_uint32, C.c_int64, C.c_int64, C.c_int64, vp]
        lib.gcmf_cheb_step.restype = C.c_int
        lib.gcmf_multi_supported.argtypes = [vp, C.c_int]
        lib.gcmf_multi_supported.restype = C.c_int
        lib.gcmf_cheb_multi.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double,
                                        C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int64, vp]
        lib.gcmf_cheb_multi.restype = C.c_int
        lib.gcmf_multi_supported_vec.argtypes = [vp, C.c_int, C.c_int64]
        lib.gcmf_multi_supported_vec.restype = C.c_int
        lib.gcmf_cheb_multi_vec.argtypes = [vp, vpp, vpp, vpp, vpp, vpp, vpp, C.POINTER(C.c_double), C.c_int, C.c_double,
                                            C.c_double, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int64, vp]
        lib.gcmf_cheb_multi_vec.restype = C.c_int
        lib.gcmf_has_land.argtypes = [vp]
        lib.gcmf_has_land.restype = C.c_int
        lib.gcmf_zero_land.argtypes = [vp, vpp, vpp, C.c_int64, vp]
        lib.gcmf_zero_land.restype = C.c_int
        lib.gcmf_land_fix.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.c_double, vpp, vpp, C.c_int64, C.c_uint32, vp]
        lib.gcmf_land_fix.restype = C.c_int
        lib.gcmf_prepare.argtypes = [vp, vpp, vpp, C.c_int64, C.c_int64, C.c_int64, vp]
        lib.gcmf_prepare.restype = C.c_int
        lib.gcmf_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        lib.gcmf_last_timing.restype = C.c_int
        lib.gcmf_slab_backward_vec_supported.argtypes = [vp, C.c_int64, C.c_int]
        lib.gcmf_slab_backward_vec_supported.restype = C.c_int
        lib.gcmf_slab_apply_backward_vec.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double, vpp, vpp, vpp, C.c_int64,
                                                     C.c_int, C.c_uint32, vp]
        lib.gcmf_slab_apply_backward_vec.restype = C.c_int
        lib.gcmf_resident_supported.argtypes = [vp, C.c_int64, C.c_int64, C.c_int]
        lib.gcmf_resident_supported.restype = C.c_int
        lib.gcmf_resident_levels.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_uint32,
                                             C.c_int64, C.c_int64, vp]
        lib.gcmf_resident_levels.restype = C.c_int
        lib.gcmf_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.gcmf_comm_info.restype = C.c_int
        lib.gcmf_comm_unique_id.argtypes = [C.c_char_p]
        lib.gcmf_comm_unique_id.restype = C.c_int
        lib.gcmf_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vpp]
        lib.gcmf_comm_create.restype = C.c_int
        lib.gcmf_comm_destroy.argtypes = [vp]
        lib.gcmf_comm_destroy.restype = None
        lib.gcmf_halo_start.argtypes = [vp, vpp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.c_int, vp]
        lib.gcmf_halo_start.restype = C.c_int
        lib.gcmf_halo_finish.argtypes = [vp, vp]
        lib.gcmf_halo_finish.restype = C.c_int
        lib.gcmf_slab_apply_backward.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double,
                                                 C.POINTER(C.c_int), C.c_int, vp, vpp, vp, C.c_int64, C.c_int, C.c_int, C.c_uint32, vp]
        lib.gcmf_slab_apply_backward.restype = C.c_int
        lib.gcmf_p2p_create.argtypes = [C.c_int, C.c_int64, vpp]
        lib.gcmf_p2p_create.restype = C.c_int
        lib.gcmf_p2p_export.argtypes = [vp, C.c_char_p]
        lib.gcmf_p2p_export.restype = C.c_int
        lib.gcmf_p2p_connect.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        lib.gcmf_p2p_connect.restype = C.c_int
        lib.gcmf_p2p_start.argtypes = [vp, vpp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, vp]
        lib.gcmf_p2p_start.restype = C.c_int
        lib.gcmf_p2p_finish.argtypes = [vp, vp]
        lib.gcmf_p2p_finish.restype = C.c_int
        lib.gcmf_p2p_status.argtypes = [vp, C.POINTER(C.c_int)]
        lib.gcmf_p2p_status.restype = C.c_int
        lib.gcmf_p2p_destroy.argtypes = [vp]
        lib.gcmf_p2p_destroy.restype = None
        lib.gcmf_p2p_guard.argtypes = [vp, vp, C.c_int64, vp]
        lib.gcmf_p2p_guard.restype = C.c_int
        lib.gcmf_p2p_seq.argtypes = [vp, C.POINTER(C.c_int64)]
        lib.gcmf_p2p_seq.restype = C.c_int
        lib.gcmf_p2p_set_timeout_ms.argtypes = [vp, C.c_int64]
        lib.gcmf_p2p_set_timeout_ms.restype = C.c_int
        lib.gcmf_p2p_debug_skip_post.argtypes = [vp, C.c_int]
        lib.gcmf_p2p_debug_skip_post.restype = C.c_int
        lib.gcmf_last_kernel_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float),
                                                C.POINTER(C.c_float)]
        lib.gcmf_last_kernel_timing.restype = C.c_int
        lib.gcmf_clenshaw_cut.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_int]
        lib.gcmf_clenshaw_cut.restype = C.c_int
        lib.gcmf_ring_fallbacks.argtypes = [vp, C.POINTER(C.c_int64)]
        lib.gcmf_ring_fallbacks.restype = C.c_int
        lib.gcmf_last_kernel.argtypes = [vp, C.c_char_p, C.c_int]
        lib.gcmf_last_kernel.restype = C.c_int
        lib.gcmf_plan_last_path.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        lib.gcmf_plan_last_path.restype = C.c_int
        lib.gcmf_resident_status.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        lib.gcmf_resident_status.restype = C.c_int
        lib.gcmf_last_kernel_geometry.argtypes = [vp, C.c_char_p, C.c_int]
        lib.gcmf_last_kernel_geometry.restype = C.c_int
        lib.gcmf_set_timing.argtypes = [vp, C.c_int]
        lib.gcmf_set_timing.restype = C.c_int
        lib.gcmf_set_tuning.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        lib.gcmf_set_tuning.restype = C.c_int
        lib.gcmf_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        lib.gcmf_set_option.restype = C.c_int
        lib.gcmf_last_error.argtypes = []
        lib.gcmf_last_error.restype = C.c_char_p
        lib.gcmf_version.argtypes = []
        lib.gcmf_version.restype = C.c_int
        _lib = lib
        return lib


def last_error() -> str:
    return load().gcmf_last_error().decode("utf-8", "replace")


def check(status: int):
    if status != OK:
        raise GcmfError(status, last_error())


def _ptr_array(ptrs: Sequence[Optional[int]]):
    arr = (C.c_void_p * max(len(ptrs), 1))()
    for k, p in enumerate(ptrs):
        arr[k] = p
    return arr


def np_dtype(code: int):
    return np.float64 if code == F64 else np.float32


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float64:
        return F64
    if dt == np.float32:
        return F32
    raise TypeError(f"libgcmf computes in float32 or float64, not {dt}")


def resident_status(device: int = 0) -> dict:
    """This process's standing with the on-chip kernel on `device`: {"state": "ok" | "lock-busy" | "disabled" | "off", "failures": n}
    (include/gcmf.h: gcmf_resident_status)."""
    st, nf = C.c_int(), C.c_uint64()
    check(load().gcmf_resident_status(int(device), C.byref(st), C.byref(nf)))
    return {"state": RESIDENT_STATES.get(st.value, "off"), "failures": int(nf.value)}


class Plan:
    """Owning handle of a gcmf_plan."""

    def __init__(self, grid_type: int, dtype: int, ny: int, nx: int, planes: Sequence, *, device: int = 0,
                 row_begin: int = 0, row_end: Optional[int] = None, halo: int = 0, planes_on_device: bool = False,
                 self_ring: bool = False, skip_kappa_one: bool = False):
        lib = load()
        self._h = None
        self.grid_type, self.dtype, self.ny, self.nx, self.device = int(grid_type), int(dtype), int(ny), int(nx), int(device)
        self.ncomp = lib.gcmf_grid_ncomp(self.grid_type)
        desc = PlanDesc(self.grid_type, self.dtype, self.ny, self.nx, int(row_begin),
                        int(self.ny if row_end is None else row_end), int(halo), self.device,
                        1 if planes_on_device else 0,
                        (PLAN_SELF_RING if self_ring else 0) | (PLAN_SKIP_KAPPA_ONE if skip_kappa_one else 0))
        if planes_on_device:
            ptrs = [int(p) for p in planes]
            keep = None
        else:
            keep = [np.ascontiguousarray(p, dtype=np_dtype(self.dtype)) for p in planes]
            for a in keep:
                if a.shape != (self.ny, self.nx):
                    raise ValueError(f"grid plane has shape {a.shape}, expected {(self.ny, self.nx)}")
            ptrs = [a.ctypes.data for a in keep]
        out = C.c_void_p()
        st = lib.gcmf_plan_create(C.byref(desc), _ptr_array(ptrs), len(ptrs), C.byref(out))
        del keep
        check(st)
        self._h = out
        ra, fo, ro = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.gcmf_plan_rows(self._h, C.byref(ra), C.byref(fo), C.byref(ro)))
        self.rows_alloc, self.first_owned, self.rows_owned = ra.value, fo.value, ro.value

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        hb = self.__dict__.pop("_host_blocks", None)   # row-block pipelines hung on this plan (host_blocks.py)
        if hb:
            for pipe in hb["pipes"].values():
                if pipe is not None:
                    pipe.close()
        if getattr(self, "_h", None):
            load().gcmf_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- whole-filter / one-Laplacian calls ----------------------------------------------------
    def apply(self, p: np.ndarray, c: float, ins: Sequence[int], outs: Sequence[int], nbatch: int, *,
              device_ptrs: bool, out_f32: bool = False, stream: int = 0, forward: bool = False, backward_f32: bool = False):
        p = np.ascontiguousarray(p, dtype=np.float64)
        flags = ((DEVICE_PTRS if device_ptrs else 0) | (OUT_F32 if out_f32 else 0) | (FORWARD_RECURRENCE if forward else 0)
                 | (BACKWARD_F32 if backward_f32 else 0))
        check(load().gcmf_apply(self._h, p.ctypes.data_as(C.POINTER(C.c_double)), len(p) - 1, float(c),
                                _ptr_array(ins), _ptr_array(outs), int(nbatch), flags, C.c_void_p(stream or None)))

    def laplacian(self, ins: Sequence[int], outs: Sequence[int], nbatch: int, *, device_ptrs: bool, stream: int = 0):
        flags = DEVICE_PTRS if device_ptrs else 0
        check(load().gcmf_laplacian(self._h, _ptr_array(ins), _ptr_array(outs), int(nbatch), flags,
                                    C.c_void_p(stream or None)))

    # -- row-slab building blocks (device pointers) --------------------------------------------
    def cheb_step(self, t1, t2, fb_in, t0, fb_out, coef0, coef1, c, mode, nbatch, row_lo, row_hi, *,
                  out_f32: bool = False, stream: int = 0):
        n = self.ncomp
        z = [None] * n
        check(load().gcmf_cheb_step(self._h, _ptr_array(t1), _ptr_array(t2 or z), _ptr_array(fb_in or z),
                                    _ptr_array(t0 or z), _ptr_array(fb_out), float(coef0), float(coef1), float(c),
                                    int(mode), OUT_F32 if out_f32 else 0, int(nbatch), int(row_lo), int(row_hi),
                                    C.c_void_p(stream or None)))

    def multi_supported(self, S: int) -> bool:
        return bool(load().gcmf_multi_supported(self._h, int(S)))

    def cheb_multi(self, u, v, uo, vo, fb_in, fb_out, pk, p0, c, mode, nbatch, row_lo, row_hi, *,
                   out_f32: bool = False, stream: int = 0):
        pk = np.ascontiguousarray(pk, dtype=np.float64)
        check(load().gcmf_cheb_multi(self._h, C.c_void_p(u or None), C.c_void_p(v or None), C.c_void_p(uo or None),
                                     C.c_void_p(vo or None), C.c_void_p(fb_in or None), C.c_void_p(fb_out),
                                     pk.ctypes.data_as(C.POINTER(C.c_double)), len(pk), float(p0), float(c), int(mode),
                                     OUT_F32 if out_f32 else 0, int(nbatch), int(row_lo), int(row_hi),
                                     C.c_void_p(stream or None)))

    def slab_apply_backward(self, comm, p2p, south, north, p, c, cut, X, pool, out, nbatch, halo, overlap, *, out_f32=False, stream=0,
                            resident=True):
        """One whole backward (Clenshaw) application on this slab incl. its halo exchanges, enqueued by libgcmf in one call
        (gcmf_slab_apply_backward).  comm / p2p: a Comm / P2P object or None; X / out: device pointers, pool: four of them.
        resident=False: never the on-chip kernel (the strip-marching launches instead: same bits)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        cut = (C.c_int * len(cut))(*[int(x) for x in cut])
        check(load().gcmf_slab_apply_backward(
            self._h, None if comm is None else comm._h, None if p2p is None else p2p._h, -1 if south is None else int(south),
            -1 if north is None else int(north), p.ctypes.data_as(C.POINTER(C.c_double)), len(p) - 1, float(c), cut, len(cut),
            C.c_void_p(X), _ptr_array(pool), C.c_void_p(out), int(nbatch), int(halo), int(bool(overlap)),
            (OUT_F32 if out_f32 else 0) | (0 if resident else NO_RESIDENT), C.c_void_p(stream or None)))

    def slab_backward_vec_supported(self, nbatch: int, halo: int = 0) -> bool:
        return bool(load().gcmf_slab_backward_vec_supported(self._h, int(nbatch), int(halo)))

    def slab_apply_backward_vec(self, comm, p2p, south, north, p, c, X, pool, out, nbatch, halo, *, out_f32=False, stream=0):
        """One whole backward application of a VECTOR plan on this slab incl. its halo exchanges (gcmf_slab_apply_backward_vec).
        X / out: two device pointers (u, v); pool: eight (four state pairs, pool[2 q + component])."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        check(load().gcmf_slab_apply_backward_vec(
            self._h, None if comm is None else comm._h, None if p2p is None else p2p._h, -1 if south is None else int(south),
            -1 if north is None else int(north), p.ctypes.data_as(C.POINTER(C.c_double)), len(p) - 1, float(c), _ptr_array(X), _ptr_array(pool),
            _ptr_array(out), int(nbatch), int(halo), OUT_F32 if out_f32 else 0, C.c_void_p(stream or None)))

    def resident_supported(self, row_lo: int, row_hi: int, L: int) -> bool:
        """Can L levels of the backward evaluation with output rows [row_lo, row_hi) run in one on-chip launch (gcmf_resident.hip)?"""
        return bool(load().gcmf_resident_supported(self._h, int(row_lo), int(row_hi), int(L)))

    def resident_levels(self, u, v, uo, vo, f, out, pk, p0, c, mode, row_lo, row_hi, *, stream: int = 0):
        """len(pk) levels of the backward evaluation in ONE on-chip launch (gcmf_resident_levels); device pointers or None."""
        pk = np.ascontiguousarray(pk, dtype=np.float64)
        vp = lambda x: C.c_void_p(x) if x else None
        check(load().gcmf_resident_levels(self._h, vp(u), vp(v), vp(uo), vp(vo), vp(f), vp(out), pk.ctypes.data_as(C.POINTER(C.c_double)),
                                          len(pk), float(p0), float(c), int(mode), int(row_lo), int(row_hi), C.c_void_p(stream or None)))

    def clenshaw_cut(self, n_steps: int):
        """Launch depths of the backward evaluation gcmf_apply uses for this polynomial length ([] = forward recurrence)."""
        buf = (C.c_int * 1024)()
        n = load().gcmf_clenshaw_cut(self._h, int(n_steps), buf, 1024)
        return [buf[i] for i in range(n)]

    def multi_supported_vec(self, S: int, nbatch: int) -> bool:
        return bool(load().gcmf_multi_supported_vec(self._h, int(S), int(nbatch)))

    def cheb_multi_vec(self, u, v, uo, vo, fb_in, fb_out, pk, p0, c, mode, nbatch, row_lo, row_hi, *,
                       out_f32: bool = False, stream: int = 0):
        """Per-component pointer lists (ncomp entries; None for the ones FIRST / LAST do not need)."""
        pk = np.ascontiguousarray(pk, dtype=np.float64)
        z = [None] * self.ncomp
        check(load().gcmf_cheb_multi_vec(self._h, _ptr_array(u or z), _ptr_array(v or z), _ptr_array(uo or z),
                                         _ptr_array(vo or z), _ptr_array(fb_in or z), _ptr_array(fb_out),
                                         pk.ctypes.data_as(C.POINTER(C.c_double)), len(pk), float(p0), float(c),
                                         int(mode), OUT_F32 if out_f32 else 0, int(nbatch), int(row_lo), int(row_hi),
                                         C.c_void_p(stream or None)))

    def has_land(self) -> bool:
        return bool(load().gcmf_has_land(self._h))

    def zero_land(self, a, b, nbatch, *, stream: int = 0):
        check(load().gcmf_zero_land(self._h, _ptr_array(a), _ptr_array(b), int(nbatch), C.c_void_p(stream or None)))

    def land_fix(self, p, c, ins, outs, nbatch, *, out_f32: bool = False, stream: int = 0):
        p = np.ascontiguousarray(p, dtype=np.float64)
        check(load().gcmf_land_fix(self._h, p.ctypes.data_as(C.POINTER(C.c_double)), len(p) - 1, float(c), _ptr_array(ins),
                                   _ptr_array(outs), int(nbatch), OUT_F32 if out_f32 else 0, C.c_void_p(stream or None)))

    def prepare(self, ins, outs, nbatch, row_lo, row_hi, *, stream: int = 0):
        check(load().gcmf_prepare(self._h, _ptr_array(ins), _ptr_array(outs), int(nbatch), int(row_lo), int(row_hi),
                                  C.c_void_p(stream or None)))

    # -- instrumentation -----------------------------------------------------------------------
    def set_timing(self, enabled):
        """False / True: event pair around the whole recurrence; 2: also one pair per temporally blocked launch."""
        check(load().gcmf_set_timing(self._h, int(enabled)))

    def last_kernel_timing(self):
        """(sum ms, launches, shortest ms, longest ms) of the blocked launches of the last apply (set_timing(2))."""
        ms, n, lo, hi = C.c_float(), C.c_int(), C.c_float(), C.c_float()
        check(load().gcmf_last_kernel_timing(self._h, C.byref(ms), C.byref(n), C.byref(lo), C.byref(hi)))
        return ms.value, n.value, lo.value, hi.value

    def last_timing(self):
        ms, n = C.c_float(), C.c_int()
        check(load().gcmf_last_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def ring_fallbacks(self) -> int:
        """Wave strips the k_ring kernels handed to the general kernel (NaN / inf met) since the last call; synchronises."""
        n = C.c_int64()
        check(load().gcmf_ring_fallbacks(self._h, C.byref(n)))
        return n.value

    def last_path(self):
        """Which of the two bit-identical paths the last gcmf_apply of this plan took: "resident" (the whole polynomial on the chip),
        "strips" (the strip-marching launches), "resident-lock-busy" (strips, because ANOTHER PROCESS holds this GPU's on-chip lock),
        "resident-disabled" (strips, because an on-chip launch of this process timed out earlier); None before the first call."""
        path = C.c_int()
        check(load().gcmf_plan_last_path(self._h, C.byref(path), None))
        return PATH_NAMES.get(path.value)

    def path_counts(self) -> dict:
        """How often each path was taken since the plan was made: {"resident": n, "strips": n, ...}."""
        path, counts = C.c_int(), (C.c_int64 * 5)()
        check(load().gcmf_plan_last_path(self._h, C.byref(path), counts))
        return {PATH_NAMES[k]: int(counts[k]) for k in range(1, 5)}

    def last_kernel(self) -> str:
        buf = C.create_string_buffer(256)
        check(load().gcmf_last_kernel(self._h, buf, 256))
        return buf.value.decode()

    def last_kernel_geometry(self) -> dict:
        """Launch geometry of the kernel last_kernel() named: {"H", "nstrips", "nwx", "xcd", "grid", "rows"} ({} if none)."""
        buf = C.create_string_buffer(256)
        check(load().gcmf_last_kernel_geometry(self._h, buf, 256))
        out = {}
        for tok in buf.value.decode().split():
            k, _, v = tok.partition("=")
            out[k] = v if "x" in v else int(v)
        return out

    def set_tuning(self, rows_per_wave: int = 0, xcd_remap: int = -1, multi_s: int = 0, strip_rows: int = 0,
                   prefetch_rows: int = 0, clenshaw: int = -1, zigzag: int = -1):
        """`clenshaw` (needs multi_s > 0): backward evaluation 0 = off, 1 = flux kinds, 2 = all scalar kinds; -1 keeps it.
        `zigzag` (needs xcd_remap >= 0): neighbouring strips of the backward flux kernels march in opposite directions, 1 / 0; -1 keeps it."""
        if zigzag >= 0:
            if xcd_remap < 0:
                raise ValueError("set_tuning: zigzag travels with xcd_remap; pass both")
            xcd_remap = (int(xcd_remap) & 1) | ((int(zigzag) + 1) << 1)
        check(load().gcmf_set_tuning(self._h, int(rows_per_wave), int(xcd_remap),
                                     (int(multi_s) & 0xFF) | ((int(strip_rows) & 0xFFFF) << 8)
                                     | ((int(prefetch_rows) & 0xF) << 24) | (((int(clenshaw) + 1) & 3) << 28 if clenshaw >= 0 else 0)))


    def set_option(self, name: str, value: int):
        """Named per-plan switch (gcmf_set_option, include/gcmf.h): "cgrid_ring", "cgrid_ring_smax", "cgrid_ring_hmax", "cgrid_ring_ncarry", "pack_batch",
        "single_launch", "ringc9", "ringc_zip", "ringc_smax", "band_seq_cells", "zip_fold", "slab_nines", "clenshaw_f32", "ring_flux_f32"."""
        check(load().gcmf_set_option(self._h, name.encode(), int(value)))


class Comm:
    """Owning handle of a gcmf_comm: the RCCL communicator + side stream libgcmf issues halo exchanges on."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(load().gcmf_comm_unique_id(buf))
        return buf.raw

    def __init__(self, uid: bytes, world: int, rank: int, device: int):
        self._h = None
        out = C.c_void_p()
        check(load().gcmf_comm_create(C.create_string_buffer(bytes(uid), 128), int(world), int(rank), int(device),
                                      C.byref(out)))
        self._h = out

    def close(self):
        if getattr(self, "_h", None):
            load().gcmf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self) -> dict:
        """What RCCL reports about this communicator (version code, ranks, this rank)."""
        v, n, r = C.c_int(), C.c_int(), C.c_int()
        check(load().gcmf_comm_info(self._h, C.byref(v), C.byref(n), C.byref(r)))
        return {"rccl_version_code": v.value, "nranks": n.value, "rank": r.value}

    def halo_start(self, states: Sequence[int], nblocks, rows_alloc, nx, first_owned, rows_owned, halo, dtype, south, north,
                   stream: int = 0):
        check(load().gcmf_halo_start(self._h, _ptr_array(states), len(states), int(nblocks), int(rows_alloc), int(nx),
                                     int(first_owned), int(rows_owned), int(halo), int(dtype),
                                     -1 if south is None else int(south), -1 if north is None else int(north),
                                     C.c_void_p(stream or None)))

    def halo_finish(self, stream: int = 0):
        check(load().gcmf_halo_finish(self._h, C.c_void_p(stream or None)))


class P2P:
    """Owning handle of a gcmf_p2p: this rank's mailbox block + the mapped blocks of its two neighbours (csrc/gcmf_p2p.hip)."""

    def __init__(self, device: int, mailbox_bytes: int):
        self._h = None
        out = C.c_void_p()
        check(load().gcmf_p2p_create(int(device), int(mailbox_bytes), C.byref(out)))
        self._h = out

    def export(self) -> bytes:
        buf = C.create_string_buffer(64)
        check(load().gcmf_p2p_export(self._h, buf))
        return buf.raw

    def connect(self, south: Optional[bytes], north: Optional[bytes], south_is_self: bool = False, north_is_self: bool = False):
        mk = lambda h: None if h is None else C.create_string_buffer(bytes(h), 64)
        self._keep = (mk(south), mk(north))
        check(load().gcmf_p2p_connect(self._h, self._keep[0], self._keep[1], int(bool(south_is_self)), int(bool(north_is_self))))

    def start(self, states: Sequence[int], nblocks, rows_alloc, nx, first_owned, rows_owned, halo, dtype, stream: int = 0):
        check(load().gcmf_p2p_start(self._h, _ptr_array(states), len(states), int(nblocks), int(rows_alloc), int(nx), int(first_owned),
                                    int(rows_owned), int(halo), int(dtype), C.c_void_p(stream or None)))

    def finish(self, stream: int = 0):
        check(load().gcmf_p2p_finish(self._h, C.c_void_p(stream or None)))

    def guard(self, out_ptr: int, nbytes: int, stream: int = 0):
        """After the last launch of an application: a failed exchange turns `nbytes` of the result into NaN."""
        check(load().gcmf_p2p_guard(self._h, C.c_void_p(out_ptr), int(nbytes), C.c_void_p(stream or None)))

    def failed(self) -> int:
        """0 = healthy, 1 = a wait of this rank timed out, 2 = a neighbour aborted the exchange (mapped host word: no device call)."""
        v = C.c_int()
        check(load().gcmf_p2p_status(self._h, C.byref(v)))
        return int(v.value)

    def timed_out(self) -> bool:
        return self.failed() != 0

    def seq(self) -> int:
        v = C.c_int64()
        check(load().gcmf_p2p_seq(self._h, C.byref(v)))
        return int(v.value)

    def set_timeout_ms(self, ms: int):
        check(load().gcmf_p2p_set_timeout_ms(self._h, int(ms)))

    def debug_skip_post(self, seq: int):
        check(load().gcmf_p2p_debug_skip_post(self._h, int(seq)))

    def close(self):
        if getattr(self, "_h", None):
            load().gcmf_p2p_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
