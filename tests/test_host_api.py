"""CPU-only checks: the C-ABI library loads and exports every symbol include/gcmf.h declares, the host-side
Filter logic (polynomial fit, defaults, error / warning contract) matches the reference, and the xarray
adapter reproduces apply_ufunc semantics (against a test-only xarray model with an oracle-backed filter_func)."""
import os
import re
import sys
import warnings

import numpy as np
import pytest
from conftest import XARRAY_KINDS, xarray_backend

import make_golden as MG
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, required_grid_vars, testing as T
from gcm_filters_amd import filter as F
from gcm_filters_amd.kernels import ALL_KERNELS, AreaWeightedMixin, BaseScalarLaplacian, BaseVectorLaplacian
from oracle import gcmf_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "gcmf.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+)?(?:int|void|char)\s*\*?\s*(gcmf_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libgcmf.so does not export {name}"
    assert declared == set(_lib.EXPORTS)
    assert lib.gcmf_version() == 1


def test_binary_is_bound_to_its_sources(monkeypatch):
    """libgcmf.so carries the sha256 of the sources it was compiled from; the loader rebuilds a binary that does not
    match when hipcc is there and refuses it otherwise -- it never loads a stale binary silently."""
    from gcm_filters_amd import _build
    want = _build.source_build_id()
    assert re.fullmatch(r"[0-9a-f]{64}", want)
    assert _build.binary_build_id() == want                      # the in-tree binary is the sources' binary
    assert _lib.load().gcmf_build_id().decode() == want          # ... and says so through the C ABI
    assert _lib.ensure_fresh_library() == want                   # fresh: neither branch below is taken

    built = []
    monkeypatch.setattr(_build, "source_build_id", lambda: "f" * 64)   # pretend the sources changed
    monkeypatch.setattr(_build, "build_library", lambda *a, **k: built.append(1))
    with pytest.raises(_lib.StaleLibraryError, match="rebuilt .* carries build id"):   # hipcc present: a rebuild is tried
        _lib.ensure_fresh_library()
    assert built == [1]

    def no_hipcc():
        raise RuntimeError("hipcc not found")
    monkeypatch.setattr(_build, "hipcc", no_hipcc)
    with pytest.raises(_lib.StaleLibraryError, match="no hipcc to rebuild"):            # no compiler: refuse
        _lib.ensure_fresh_library()
    assert built == [1]


def test_no_build_by_products_are_tracked():
    import subprocess
    if not os.path.isdir(os.path.join(REPO, ".git")):
        pytest.skip("not a git checkout")
    files = subprocess.run(["git", "ls-files"], cwd=REPO, capture_output=True, text=True).stdout.split()
    assert not [f for f in files if "libgcmf.so" in f or f.endswith((".o", ".so"))]


def test_static_grid_facts_match_reference_table():
    lib = _lib.load()
    for name, val in O.GRID_TYPE_VALUES.items():
        assert GridType[name].value == val
        assert lib.gcmf_grid_nplanes(val) == len(O.GRID_ARGS[name])
        assert lib.gcmf_grid_ncomp(val) == (2 if name in O.VECTOR else 1)
        assert bool(lib.gcmf_grid_is_dimensional(val)) == O.DIMENSIONAL[name]
        assert bool(lib.gcmf_grid_is_tripolar(val)) == name.startswith("TRIPOLAR")
    assert lib.gcmf_grid_nplanes(99) == -1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback():
    """Without an MI355X the product path fails loudly (status GCMF_ERR_NO_DEVICE), it never computes on the CPU."""
    f, gv = T.scalar_case("REGULAR_WITH_LAND", (8, 8))
    with pytest.raises(_lib.GcmfError) as ei:
        ALL_KERNELS[GridType.REGULAR_WITH_LAND](**gv)
    assert ei.value.status == _lib.ERR_NO_DEVICE
    with pytest.raises(_lib.GcmfError):
        Filter(filter_scale=4, dx_min=1).apply(np.zeros((8, 8)))


def test_bad_arguments_are_rejected_before_touching_the_device():
    import ctypes as C
    lib = _lib.load()
    out = C.c_void_p()
    desc = _lib.PlanDesc(99, _lib.F64, 8, 8, 0, 8, 0, 0, 0, 0)
    assert lib.gcmf_plan_create(C.byref(desc), None, 0, C.byref(out)) == _lib.ERR_INVALID_ARG
    assert b"grid_type" in lib.gcmf_last_error()
    desc = _lib.PlanDesc(GridType.REGULAR_WITH_LAND.value, _lib.F64, 8, 8, 0, 8, 0, 0, 0, 0)
    assert lib.gcmf_plan_create(C.byref(desc), None, 0, C.byref(out)) == _lib.ERR_INVALID_ARG
    desc = _lib.PlanDesc(GridType.REGULAR.value, 7, 8, 8, 0, 8, 0, 0, 0, 0)
    assert lib.gcmf_plan_create(C.byref(desc), None, 0, C.byref(out)) == _lib.ERR_INVALID_ARG
    desc = _lib.PlanDesc(GridType.REGULAR.value, _lib.F64, 8, 8, 4, 2, 0, 0, 0, 0)
    assert lib.gcmf_plan_create(C.byref(desc), None, 0, C.byref(out)) == _lib.ERR_INVALID_ARG


# ---------------------------------------------------------------------------------------------------
# kernel registry (upstream tests/test_kernels.py:39-61, 285-299)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid", T.ALL_GRIDS)
def test_registry(grid):
    cls = ALL_KERNELS[GridType[grid]]
    assert required_grid_vars(GridType[grid]) == list(O.GRID_ARGS[grid])
    assert cls.required_grid_args() == list(O.GRID_ARGS[grid])
    assert cls.is_dimensional == O.DIMENSIONAL[grid]
    assert issubclass(cls, BaseVectorLaplacian) == (grid in O.VECTOR)
    assert issubclass(cls, BaseScalarLaplacian) == (grid not in O.VECTOR)
    assert issubclass(cls, AreaWeightedMixin) == (grid in O.AREA_WEIGHTED)


# ---------------------------------------------------------------------------------------------------
# filter polynomial (upstream tests/test_filter.py:13-92)
# ---------------------------------------------------------------------------------------------------
def test_filter_spec_known_answers():
    f = Filter(filter_scale=10.0, dx_min=1.0, filter_shape=FilterShape.GAUSSIAN, transition_width=np.pi, ndim=2)
    assert f.filter_spec.n_steps == 11 and f.filter_spec.s_max == 8.0
    np.testing.assert_allclose(f.filter_spec.p,
                               [0.09887381, -0.19152534, 0.1748326, -0.14975371, 0.12112337, -0.09198484, 0.0662522,
                                -0.04479323, 0.02895827, -0.0173953, 0.00995974, -0.00454758], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(f.filter_spec.dx_min_sq, 1.0)
    f = Filter(filter_scale=2.0, dx_min=1.0, filter_shape=FilterShape.TAPER, transition_width=np.pi, ndim=1)
    assert f.filter_spec.n_steps == 6 and f.filter_spec.s_max == 4.0
    np.testing.assert_allclose(f.filter_spec.p, [0.83380304, -0.23622724, -0.06554041, 0.01593978, 0.00481014,
                                                 -0.00495532, 0.00168445], rtol=1e-7, atol=1e-7)
    assert F._compute_n_steps_default(2, FilterShape.GAUSSIAN, 1.5, 1, np.pi) >= 3


@pytest.mark.parametrize("row", MG.SPEC_TABLE, ids=lambda r: f"{r[0]}-{r[1]}-{r[2]}-{r[4]}-{r[5]}")
def test_filter_spec_matches_reference_table(row, golden_spec):
    shape, scale, dx_min, tw, ndim, n = row
    key = f"{shape}|{scale!r}|{dx_min!r}|{tw!r}|{ndim}|{n}"
    nd, nn, s_max, dxsq = golden_spec[key + "|meta"]
    flt = Filter(filter_scale=scale, dx_min=dx_min, filter_shape=FilterShape[shape], transition_width=tw, ndim=ndim,
                 n_steps=n)
    assert flt.n_steps == int(nn) and flt.filter_spec.n_steps == int(nn)
    assert flt.filter_spec.s_max == s_max and flt.filter_spec.dx_min_sq == dxsq
    np.testing.assert_allclose(flt.filter_spec.p, golden_spec[key + "|p"], rtol=0, atol=5e-14)
    assert abs(sum((-1) ** k * c for k, c in enumerate(flt.filter_spec.p)) - 1) < 1e-13  # p(-1) = 1


def test_own_pchip_equals_scipy():
    from scipy.interpolate import PchipInterpolator
    rng = np.random.default_rng(3)
    for _ in range(20):
        x = np.sort(rng.random(6)) * 10 + np.arange(6)
        y = rng.normal(size=6)
        q = np.linspace(x[0], x[-1], 301)
        np.testing.assert_allclose(F._pchip(x, y)(q), PchipInterpolator(x, y)(q), rtol=1e-12, atol=1e-13)


# ---------------------------------------------------------------------------------------------------
# constructor contract (upstream tests/test_filter.py:140-169, 284-290)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid", T.ALL_GRIDS)
def test_constructor_contract(grid):
    vec = grid in T.VECTOR_GRIDS
    gv = T.vector_grid_vars(grid, (8, 8)) if vec else T.scalar_grid_vars(grid, (8, 8))
    gt = GridType[grid]
    args = dict(filter_scale=3.0, dx_min=1.0, n_steps=0, filter_shape=FilterShape.GAUSSIAN)
    flt = Filter(grid_type=gt, grid_vars=gv, **args)
    assert flt.n_steps == int(O.n_steps_default(2, "GAUSSIAN", 3.0, 1.0))
    for name in gv:
        with pytest.raises(ValueError, match=r"Provided `grid_vars` .*"):
            Filter(grid_type=gt, grid_vars={k: v for k, v in gv.items() if k != name}, **args)
    with pytest.raises(ValueError, match=r"Transition width .*"):
        Filter(grid_type=gt, grid_vars=gv, **dict(args, transition_width=1))
    with pytest.raises(ValueError, match=r"When ndim > 2, you .*"):
        Filter(grid_type=gt, grid_vars=gv, **dict(args, ndim=3))
    with pytest.warns(UserWarning, match=r"You have set n_steps .*"):
        Filter(grid_type=gt, grid_vars=gv, **dict(args, n_steps=3))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        Filter(grid_type=gt, grid_vars=gv, **dict(args, n_steps=16))  # above the default: silent
    if grid in O.AREA_WEIGHTED:
        with pytest.raises(ValueError, match=r"Provided Laplacian .*"):
            Filter(grid_type=gt, grid_vars=gv, **dict(args, dx_min=3))
    if vec:
        with pytest.raises(ValueError, match=r"Provided Laplacian *"):
            flt.apply(np.zeros((8, 8)))
    else:
        with pytest.raises(ValueError, match=r"Provided Laplacian *"):
            flt.apply_to_vector(np.zeros((8, 8)), np.zeros((8, 8)))


# ---------------------------------------------------------------------------------------------------
# xarray front door (upstream tests/test_filter.py:172-252) against the test-only xarray model
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(params=XARRAY_KINDS)
def xr_model(request, monkeypatch):
    """(named for its first resident: the model of xarray; "real" = the installed xarray, skipped where there is none)"""
    backend = xarray_backend(request.param, monkeypatch)

    def oracle_filter_func(spec, Laplacian, evaluation="auto"):
        o = O.FilterSpec(spec.n_steps, spec.s_max, np.asarray(spec.p), spec.dx_min_sq)
        names = Laplacian.required_grid_args()
        return lambda field, *args: O.filter_func(o, Laplacian.GRID_TYPE.name, field, dict(zip(names, args)))

    def oracle_filter_func_vec(spec, Laplacian, evaluation="auto"):
        o = O.FilterSpec(spec.n_steps, spec.s_max, np.asarray(spec.p), spec.dx_min_sq)
        names = Laplacian.required_grid_args()
        return lambda u, v, *args: O.filter_func_vec(o, Laplacian.GRID_TYPE.name, u, v, dict(zip(names, args)))

    monkeypatch.setattr(F, "_create_filter_func", oracle_filter_func)
    monkeypatch.setattr(F, "_create_filter_func_vec", oracle_filter_func_vec)
    return backend


def test_application_to_dataset(xr_model):
    xr = xr_model
    rng = np.random.default_rng(0)
    ds = xr.Dataset(data_vars=dict(spatial=(("y", "x"), rng.normal(size=(30, 40))),
                                   temporal=(("time",), rng.normal(size=(10,))),
                                   spatiotemporal=(("time", "y", "x"), rng.normal(size=(10, 30, 40))),
                                   transposed=(("y", "time", "x"), rng.normal(size=(30, 10, 40)))))
    flt = Filter(filter_scale=4, dx_min=1, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType.REGULAR)
    spatial_before = ds.spatial.data.copy()
    out = flt.apply(ds, ["y", "x"])
    assert np.array_equal(out.temporal.data, ds.temporal.data)
    assert not np.allclose(out.spatial.data, ds.spatial.data)
    spec = O.make_spec(4, 1, "GAUSSIAN")
    np.testing.assert_allclose(out.spatial.data, O.filter_func(spec, "REGULAR", ds.spatial.data, {}), rtol=1e-12)
    assert out.spatiotemporal.dims == ("time", "y", "x")
    np.testing.assert_allclose(out.spatiotemporal.data.mean(axis=(1, 2)), ds.spatiotemporal.data.mean(axis=(1, 2)),
                               atol=1e-12)
    # core dims are moved to the end, exactly like xarray.apply_ufunc does
    assert out.transposed.dims == ("time", "y", "x")
    np.testing.assert_allclose(out.transposed.data[3],
                               O.filter_func(spec, "REGULAR", ds.transposed.data[:, 3, :], {}), rtol=1e-12)
    assert np.array_equal(ds.spatial.data, spatial_before)  # input untouched
    with pytest.warns(UserWarning, match=r".* nothing was filtered."):
        flt.apply(ds, ["foo", "bar"])
    with pytest.warns(UserWarning, match=r".* nothing was filtered."):
        flt.apply(ds, ["yy", "x"])


def test_dataarray_with_grid_vars_and_vector(xr_model):
    xr = xr_model
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (20, 24))
    gvx = {k: xr.DataArray(v, dims=["y", "x"]) for k, v in gv.items()}
    flt = Filter(filter_scale=3.0, dx_min=1.0, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gvx)
    out = flt.apply(xr.DataArray(f, dims=["y", "x"]), dims=["y", "x"])
    np.testing.assert_allclose(out.data, O.filter_func(O.make_spec(3.0, 1.0), "IRREGULAR_WITH_LAND", f, gv), rtol=1e-12)
    with pytest.raises(AssertionError):
        flt.apply(xr.DataArray(f, dims=["y", "x"]), dims=["y"])
    (u, v), gvv = T.vector_case("VECTOR_B_GRID", (20, 24))
    fv = Filter(filter_scale=5.0, dx_min=1.0, n_steps=10, filter_shape=FilterShape.TAPER, grid_type=GridType.VECTOR_B_GRID,
                grid_vars={k: xr.DataArray(a, dims=["y", "x"]) for k, a in gvv.items()})
    uo, vo = fv.apply_to_vector(xr.DataArray(u, dims=["y", "x"]), xr.DataArray(v, dims=["y", "x"]), dims=["y", "x"])
    wu, wv = O.filter_func_vec(O.make_spec(5.0, 1.0, "TAPER", n_steps=10), "VECTOR_B_GRID", u, v, gvv)
    np.testing.assert_allclose(uo.data, wu, rtol=1e-12)
    np.testing.assert_allclose(vo.data, wv, rtol=1e-12)


def test_nondimensional_invariance(xr_model):
    """upstream tests/test_filter.py:219-252: (filter_scale 4, dx_min 1) and (8, 2) are the same non-dimensional filter."""
    xr = xr_model
    ds = xr.Dataset(data_vars=dict(spatial=(("y", "x"), np.random.default_rng(1).normal(size=(40, 40)))),
                    coords=dict(x=np.linspace(0, 1e6, 40), y=np.linspace(0, 1e6, 40)))
    a = Filter(filter_scale=4, dx_min=1, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType.REGULAR).apply(ds, ["y", "x"])
    b = Filter(filter_scale=8, dx_min=2, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType.REGULAR).apply(ds, ["y", "x"])
    xr.testing.assert_allclose(a.spatial, b.spatial)


def test_apply_ufunc_contract_of_the_model(xr_model):
    """What the adapter relies on in xarray.apply_ufunc (reference filter.py:478-486), pinned on the model (README: "xarray semantics
    assumed"): operands are aligned by dimension NAME, a grid variable with fewer dims arrives with fewer axes, one with dims in the
    middle missing gets length-1 axes there, an operand without a core dim is an error, and so is a third name in `dims`."""
    xr = xr_model
    rng = np.random.default_rng(2)
    nt, nz, ny, nx = 3, 4, 12, 16
    fld = xr.DataArray(rng.normal(size=(nt, nz, ny, nx)), dims=["time", "z", "y", "x"])
    mask3 = (rng.random((nz, ny, nx)) > 0.2).astype(float)          # a depth-dependent wet mask: (z, y, x)
    mask3[:, 0, :] = 0
    flt = Filter(filter_scale=3.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND,
                 grid_vars={"wet_mask": xr.DataArray(mask3, dims=["z", "y", "x"])})
    out = flt.apply(fld, dims=["y", "x"])
    assert out.dims == ("time", "z", "y", "x")
    spec = O.make_spec(3.0, 1.0)
    for k in range(nz):
        np.testing.assert_allclose(out.data[1, k], O.filter_func(spec, "REGULAR_WITH_LAND", fld.data[1, k], {"wet_mask": mask3[k]}), rtol=1e-12)
    # a (time, y, x) mask next to a (time, z, y, x) field: a length-1 axis is inserted where "z" is missing
    mask_t = np.broadcast_to(mask3[0], (nt, ny, nx)).copy()
    flt_t = Filter(filter_scale=3.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND,
                   grid_vars={"wet_mask": xr.DataArray(mask_t, dims=["time", "y", "x"])})
    out_t = flt_t.apply(fld, dims=["y", "x"])
    np.testing.assert_allclose(out_t.data[2, 3], O.filter_func(spec, "REGULAR_WITH_LAND", fld.data[2, 3], {"wet_mask": mask3[0]}), rtol=1e-12)
    # dims in another order on the field: core dims are moved to the end, the others keep their order
    out_p = flt.apply(xr.DataArray(fld.data.transpose(2, 0, 3, 1), dims=["y", "time", "x", "z"]), dims=["y", "x"])
    assert out_p.dims == ("time", "z", "y", "x")
    np.testing.assert_allclose(out_p.data, out.data, rtol=1e-12)
    with pytest.raises(ValueError, match="core dimensions"):        # a field that lacks one of the dims
        flt.apply(xr.DataArray(rng.normal(size=(nz, ny)), dims=["z", "y"]), dims=["y", "x"])
    with pytest.raises(AssertionError):                             # reference filter.py:476: assert len(dims) == 2
        flt.apply(fld, dims=["z", "y", "x"])
    with pytest.raises(ValueError, match="mismatched lengths"):     # same name, other size
        flt.apply(xr.DataArray(rng.normal(size=(nt, nz + 1, ny, nx)), dims=["time", "z", "y", "x"]), dims=["y", "x"])


def test_plan_cache_fingerprint_and_host_outputs():
    """Host-side plumbing that runs on every call: the plan-cache key of a grid plane (address, layout, 256-value
    sample) and the result allocator (page-locked pool on a GPU box, plain numpy here)."""
    from gcm_filters_amd import kernels as K
    a = np.arange(40 * 64, dtype=np.float64).reshape(40, 64)
    fa = K._fingerprint(a)
    assert fa == K._fingerprint(a)                       # stable
    assert fa != K._fingerprint(a.copy())                # another buffer
    assert fa != K._fingerprint(a.astype(np.float32))    # another dtype
    assert K._fingerprint(a[:, ::2])[2:5] == ((40, 32), a[:, ::2].strides, "<f8")  # views keep their own layout
    a[0, 0] += 1.0                                       # in-place edit of a sampled value is seen
    assert fa != K._fingerprint(a)
    big = np.zeros((600, 700))                           # > 256 samples: stride through the plane
    fb = K._fingerprint(big)
    big[-1, -1] = 1.0                                    # not necessarily sampled: only documented as "cheap check"
    big[0, 0] = 2.0
    assert fb != K._fingerprint(big)
    out = K._host_output((3, 40, 64), np.float64)        # no HIP device here: ordinary array, right shape / dtype
    assert out.shape == (3, 40, 64) and out.dtype == np.float64 and out.flags.c_contiguous and out.flags.writeable
    small = K._host_output((4, 4), np.float32)
    assert small.dtype == np.float32 and K._pinned_out >= 0


def test_host_planes_are_write_protected_while_a_plan_refers_to_them():
    """The plan cache's guard against stale coefficients (kernels._HostLocks; no device needed): arrays and the owners of
    their buffers become read-only, nested holders are reference counted, views get their flag back with their base."""
    from gcm_filters_amd import kernels as K
    a = np.zeros((6, 8))
    v = a[1:]
    t1 = K._HOST_LOCKS.acquire([v])              # plan 1 was folded from a view
    assert not v.flags.writeable and not a.flags.writeable
    with pytest.raises(ValueError, match="read-only"):
        a[0, 0] = 1.0
    assert not a[2:].flags.writeable             # views made from now on inherit the protection
    t2 = K._HOST_LOCKS.acquire([a])              # plan 2 from the owner itself
    K._HOST_LOCKS.release(t1)
    assert not a.flags.writeable and not v.flags.writeable
    K._HOST_LOCKS.release(t2)
    assert a.flags.writeable and v.flags.writeable
    a[0, 0] = 1.0
    ro = np.zeros(4)
    ro.flags.writeable = False                   # the user's own read-only array is left alone
    K._HOST_LOCKS.release(K._HOST_LOCKS.acquire([ro]))
    assert not ro.flags.writeable
    assert K._owner(v) is a and K._fingerprint(a) == K._fingerprint(a)


# ---------------------------------------------------------------------------------------------------
# plan cache (a displaced / evicted plan must never be destroyed under a thread that holds it)
# ---------------------------------------------------------------------------------------------------
class _FakePlan:
    def __init__(self):
        self.closed = 0

    def close(self):
        self.closed += 1


def test_plan_cache_cold_miss_from_many_threads_builds_one_plan():
    import threading
    import time
    from gcm_filters_amd.kernels import _PlanCache

    cache = _PlanCache(capacity=4)
    built, got = [], []
    start = threading.Barrier(8)

    def factory():
        time.sleep(0.05)              # long enough for every thread to have missed
        built.append(_FakePlan())
        return built[-1]

    def worker():
        start.wait()
        got.append(cache.get(("k",), factory))

    th = [threading.Thread(target=worker) for _ in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert len(built) == 1 and all(g is built[0] for g in got) and built[0].closed == 0


def test_plan_cache_eviction_and_dead_owner_do_not_destroy_a_plan_in_use():
    from gcm_filters_amd.kernels import _PlanCache

    cache = _PlanCache(capacity=2)
    plans = [cache.get((k,), _FakePlan) for k in range(4)]       # 0 and 1 are evicted while a caller still holds them
    assert [p.closed for p in plans] == [0, 0, 0, 0]
    assert cache.get((3,), _FakePlan) is plans[3] and cache.get((0,), _FakePlan) is not plans[0]
    a = np.ones((4, 4))
    held = cache.get(("arr",), _FakePlan, host_planes=[a])
    assert not a.flags.writeable                                   # write-protected while cached
    del a                                                          # owner dies: the entry is forgotten, the plan survives
    assert cache.get(("arr",), _FakePlan) is not held and held.closed == 0
    cache.clear()                                                  # the explicit clear is the only thing that closes


def test_plan_cache_factory_error_leaves_no_gate_behind():
    from gcm_filters_amd.kernels import _PlanCache

    cache = _PlanCache()

    def boom():
        raise ValueError("kappa")
    with pytest.raises(ValueError):
        cache.get(("k",), boom)
    assert cache.get(("k",), _FakePlan).closed == 0 and not cache._building


def test_rendezvous_ports_come_from_below_the_ephemeral_range():
    """tests / bench.py pick their local torch.distributed rendezvous port with testing.free_port(): free now, and outside the range the kernel
    hands out as source ports of outgoing connections (an OS-chosen port was found taken -- EADDRINUSE -- once in ~50 multi-process runs)."""
    import socket
    from gcm_filters_amd.testing import free_port
    lo = 32768
    try:
        lo = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError):
        pass
    ports = {free_port() for _ in range(20)}
    assert len(ports) >= 10 and all(10000 <= p < max(min(lo, 32768), 12000) for p in ports)
    p = free_port()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", p))      # really free


def test_build_is_serialised_and_a_fresh_binary_is_not_rebuilt(tmp_path):
    """_build.build_library: an inter-process lock around the build (the ranks of a torchrun job all find a stale binary at once), an early
    return when the binary already carries the sources' build id, the link through a temporary name (advisor finding, round 3)."""
    import inspect
    from gcm_filters_amd import _build
    src = inspect.getsource(_build.build_library) + inspect.getsource(_build._build_library_locked)
    assert "fcntl.flock" in src and "os.replace(tmp, LIB)" in src
    if _build.binary_build_id() == _build.source_build_id():
        before = os.path.getmtime(_build.LIB)
        assert _build.build_library() == _build.LIB and os.path.getmtime(_build.LIB) == before


def test_evaluation_keyword_values():
    """`evaluation` is a keyword-only extension of the reference's dataclass: "auto" (default), "reference" (the reference's forward
    recurrence and accumulation scheme), "backward" (backward evaluation also for f32 scalar / B-grid fields); anything else raises at
    construction, like the reference's own field validation (gcm_filters/filter.py:389-434)."""
    from gcm_filters_amd.filter import EVALUATIONS
    assert EVALUATIONS == ("auto", "reference", "backward")
    for ev in EVALUATIONS:
        flt = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR, evaluation=ev)
        assert flt.evaluation == ev and "evaluation" not in repr(flt)
    with pytest.raises(ValueError, match="evaluation must be one of"):
        Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR, evaluation="fast")
    with pytest.raises(TypeError):
        Filter(4.0, 1.0, FilterShape.GAUSSIAN, np.pi, 2, 0, GridType.REGULAR, {}, "reference")   # keyword only
