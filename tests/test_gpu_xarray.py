"""The xarray front door on the HIP path: ``Filter.apply(ds, dims)`` / ``apply_to_vector`` driven through
``xarray.apply_ufunc`` down to libgcmf, nothing monkeypatched but the xarray module itself (xarray is not installed in
this image: tests/fake_xarray.py models the documented semantics Filter relies on, incl. dask="parallelized" calling
filter_func concurrently from worker threads, block by block).  Every test also runs against the INSTALLED xarray (dask-backed inputs
where dask imports) wherever there is one -- skipped here.  Mirrors upstream tests/test_filter.py:172-252."""
import sys

import numpy as np
import pytest
from conftest import XARRAY_KINDS, xarray_backend

from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=XARRAY_KINDS)
def xr(request, monkeypatch):
    """The model of xarray (always) and the installed xarray (+ dask-backed inputs) wherever it imports."""
    return xarray_backend(request.param, monkeypatch)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max())


def test_application_to_dataset(xr):
    rng = np.random.default_rng(0)
    ds = xr.Dataset(data_vars=dict(spatial=(("y", "x"), rng.normal(size=(100, 120))),
                                   temporal=(("time",), rng.normal(size=(10,))),
                                   spatiotemporal=(("time", "y", "x"), rng.normal(size=(10, 100, 120))),
                                   transposed=(("y", "time", "x"), rng.normal(size=(100, 10, 120)))))
    # (evaluation="reference": the forward recurrence, bit-exact with numpy on REGULAR -- this test is about the adapter)
    flt = Filter(filter_scale=4, dx_min=1, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType.REGULAR, evaluation="reference")
    out = flt.apply(ds, ["y", "x"])
    # the same Chebyshev coefficients on both sides (Filter's own fit agrees with the reference's to 5e-14, not to the bit)
    fs = flt.filter_spec
    spec = O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq)
    assert np.array_equal(out.temporal.data, ds.temporal.data)                       # no spatial dims: carried along
    assert np.array_equal(out.spatial.data, O.filter_func(spec, "REGULAR", ds.spatial.data, {}))   # REGULAR is bit-exact
    assert out.spatiotemporal.dims == ("time", "y", "x")
    assert np.array_equal(out.spatiotemporal.data, O.filter_func(spec, "REGULAR", ds.spatiotemporal.data, {}))
    assert out.transposed.dims == ("time", "y", "x")                                  # core dims moved to the end
    assert np.array_equal(out.transposed.data, O.filter_func(spec, "REGULAR", ds.transposed.data.transpose(1, 0, 2), {}))
    assert out.spatial.data.dtype == np.float64
    with pytest.warns(UserWarning, match=r".* nothing was filtered."):
        flt.apply(ds, ["foo", "bar"])
    with pytest.warns(UserWarning, match=r".* nothing was filtered."):
        flt.apply(ds, ["yy", "x"])


@pytest.mark.parametrize("grid", ["REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND", "MOM5U"])
def test_dataarray_with_xarray_grid_vars(xr, grid):
    f, gv = T.scalar_case(grid, (64, 96))
    f = np.where(gv["wet_mask"] == 0, np.nan, f)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    gvx = {k: xr.DataArray(v, dims=["y", "x"]) for k, v in gv.items()}
    flt = Filter(filter_scale=5.0 * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gvx)
    assert isinstance(flt.grid_ds, xr.Dataset)
    out = flt.apply(xr.DataArray(f, dims=["y", "x"]), dims=["y", "x"])
    with np.errstate(all="ignore"):
        want = O.filter_func(O.make_spec(5.0 * dx, dx), grid, f, gv)
    assert out.dims == ("y", "x") and np.array_equal(np.isnan(out.data), np.isnan(want))
    ok = ~np.isnan(want)
    assert rel(out.data[ok], want[ok]) <= 1e-11
    with pytest.raises(AssertionError):
        flt.apply(xr.DataArray(f, dims=["y", "x"]), dims=["y"])                       # upstream filter.py:476


@pytest.mark.parametrize("grid", T.VECTOR_GRIDS)
def test_vector_through_apply_ufunc(xr, grid):
    (u, v), gv = T.vector_case(grid, (64, 96))
    dx = T.grid_dx_min(grid, gv)
    flt = Filter(filter_scale=6.0 * dx, dx_min=dx, grid_type=GridType[grid],
                 grid_vars={k: xr.DataArray(a, dims=["y", "x"]) for k, a in gv.items()})
    u3 = np.stack([u, 2 * u, u + v])
    v3 = np.stack([v, v - u, 0.5 * v])
    uo, vo = flt.apply_to_vector(xr.DataArray(u3, dims=["lev", "y", "x"]), xr.DataArray(v3, dims=["lev", "y", "x"]),
                                 dims=["y", "x"])
    wu, wv = O.filter_func_vec(O.make_spec(6.0 * dx, dx), grid, u3, v3, gv)
    assert uo.dims == ("lev", "y", "x") and rel(uo.data, wu) <= 1e-11 and rel(vo.data, wv) <= 1e-11
    with pytest.raises(ValueError, match=r"Provided Laplacian .* is a vector Laplacian.*"):
        flt.apply(xr.DataArray(u, dims=["y", "x"]), dims=["y", "x"])


def test_dask_like_blocks_from_worker_threads(xr):
    """dask="parallelized": filter_func is called per block from concurrent worker threads (reference filter.py:485).  The
    blocks share one cached device plan; results must equal the one-shot call and the oracle."""
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (96, 160))
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    gvx = {k: xr.DataArray(v, dims=["y", "x"]) for k, v in gv.items()}
    flt = Filter(filter_scale=8 * dx, dx_min=dx, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gvx)
    data = np.stack([T.random_field((96, 160), 300 + k) for k in range(13)])
    whole = flt.apply(xr.DataArray(data, dims=["time", "y", "x"]), dims=["y", "x"])
    lazy = xr.chunked(xr.DataArray(data, dims=["time", "y", "x"]), "time", 5)
    blocks = flt.apply(lazy, dims=["y", "x"])
    assert blocks.dims == ("time", "y", "x") and np.array_equal(np.asarray(blocks.data), whole.data)
    want = O.filter_func(O.make_spec(8 * dx, dx), "IRREGULAR_WITH_LAND", data, gv)
    assert rel(np.asarray(blocks.data), want) <= 1e-11
    # transposed lazy input: every block arrives as a non-contiguous view with the core dims moved last
    lazy_t = xr.chunked(xr.DataArray(data.transpose(1, 0, 2).copy(), dims=["y", "time", "x"]), "time", 4)
    assert np.array_equal(np.asarray(flt.apply(lazy_t, dims=["y", "x"]).data), whole.data)
    # vector filter, blocks over levels
    (u, v), gvv = T.vector_case("VECTOR_C_GRID", (64, 96))
    dxv = T.grid_dx_min("VECTOR_C_GRID", gvv)
    fv = Filter(filter_scale=5 * dxv, dx_min=dxv, grid_type=GridType.VECTOR_C_GRID,
                grid_vars={k: xr.DataArray(a, dims=["y", "x"]) for k, a in gvv.items()})
    U = np.stack([u * (1 + 0.1 * k) for k in range(9)])
    V = np.stack([v * (1 - 0.1 * k) for k in range(9)])
    uo, vo = fv.apply_to_vector(xr.chunked(xr.DataArray(U, dims=["lev", "y", "x"]), "lev", 3),
                                xr.DataArray(V, dims=["lev", "y", "x"]), dims=["y", "x"])
    wu, wv = O.filter_func_vec(O.make_spec(5 * dxv, dxv), "VECTOR_C_GRID", U, V, gvv)
    assert rel(np.asarray(uo.data), wu) <= 1e-11 and rel(np.asarray(vo.data), wv) <= 1e-11


def test_depth_dependent_mask_through_xarray(xr):
    """wet_mask(z, y, x) as an xarray grid variable: apply_ufunc broadcasts z between field and mask (filter.py:478-486)."""
    import make_golden as MG
    fields, gv, fk = MG.build_gridbatched_case("REGULAR_WITH_LAND")
    gvx = {"wet_mask": xr.DataArray(gv["wet_mask"], dims=["z", "y", "x"])}
    flt = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gvx,
                 evaluation="reference")
    out = flt.apply(xr.DataArray(fields[0], dims=["time", "z", "y", "x"]), dims=["y", "x"])
    fs = flt.filter_spec
    want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "REGULAR_WITH_LAND", fields[0], gv)
    assert out.dims == ("time", "z", "y", "x") and np.array_equal(out.data, want)
    auto = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gvx)
    got = auto.apply(xr.DataArray(fields[0], dims=["time", "z", "y", "x"]), dims=["y", "x"]).data    # the default (backward evaluation)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(np.abs(got - want)) <= 1e-13 * np.nanmax(np.abs(want))
