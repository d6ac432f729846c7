"""Pin the CPU oracle (oracle/gcmf_oracle.py) to the reference: its 18 zarr goldens, its known-answer
polynomial coefficients, and fp64 vectors captured from the imported reference."""
import numpy as np
import pytest

import make_golden as MG
from gcm_filters_amd import testing as T
from oracle import gcmf_oracle as O


def _stack(res):
    return np.stack(res) if isinstance(res, tuple) else res


@pytest.mark.parametrize("grid", T.REFERENCE_TESTED_GRIDS)
def test_kernel_matches_reference_zarr(grid, golden_zarr):
    """upstream tests/test_kernels_validation.py:68-75 -- one Laplacian application, stored as f4."""
    if grid in T.VECTOR_GRIDS:
        fields, gv = T.vector_case(grid)
    else:
        f, gv = T.scalar_case(grid)
        fields = (f,)
    res = _stack(O.make_laplacian(grid, gv)(*fields)).astype("f4")
    np.testing.assert_allclose(golden_zarr[f"test_data_kernels/{grid}"], res)
    assert np.array_equal(golden_zarr[f"test_data_kernels/{grid}"], res)  # bit-equal after the f4 cast


@pytest.mark.parametrize("grid", T.REFERENCE_TESTED_GRIDS)
def test_filter_matches_reference_zarr(grid, golden_zarr):
    """upstream tests/test_filter_validation.py:75-93 -- Gaussian, scale 8, dx_min 1 (n_steps 9)."""
    spec = O.make_spec(8.0, 1.0, "GAUSSIAN")
    assert spec.n_steps == 9
    if grid in T.VECTOR_GRIDS:
        (u, v), gv = T.vector_case(grid)
        res = np.stack(O.filter_func_vec(spec, grid, u, v, gv))
    else:
        f, gv = T.scalar_case(grid)
        res = O.filter_func(spec, grid, f, gv)
    res = res.astype("f4")
    np.testing.assert_allclose(golden_zarr[f"test_data_filter/{grid}"], res)
    assert np.array_equal(golden_zarr[f"test_data_filter/{grid}"], res)


def test_filter_spec_known_answers():
    """upstream tests/test_filter.py:23-84."""
    s = O.make_spec(10.0, 1.0, "GAUSSIAN")
    assert s.n_steps == 11 and s.s_max == 8.0 and s.dx_min_sq == 1.0
    np.testing.assert_allclose(s.p, [0.09887381, -0.19152534, 0.1748326, -0.14975371, 0.12112337, -0.09198484,
                                     0.0662522, -0.04479323, 0.02895827, -0.0173953, 0.00995974, -0.00454758],
                               rtol=1e-7, atol=1e-7)
    s = O.make_spec(2.0, 1.0, "TAPER", ndim=1)
    assert s.n_steps == 6 and s.s_max == 4.0
    np.testing.assert_allclose(s.p, [0.83380304, -0.23622724, -0.06554041, 0.01593978, 0.00481014, -0.00495532,
                                     0.00168445], rtol=1e-7, atol=1e-7)
    assert O.n_steps_default(2, "GAUSSIAN", 1.5, 1, np.pi) >= 3


@pytest.mark.parametrize("row", MG.SPEC_TABLE, ids=lambda r: f"{r[0]}-{r[1]}-{r[2]}-{r[4]}-{r[5]}")
def test_filter_spec_matches_reference(row, golden_spec):
    shape, scale, dx_min, tw, ndim, n = row
    key = f"{shape}|{scale!r}|{dx_min!r}|{tw!r}|{ndim}|{n}"
    nd, nn, s_max, dxsq = golden_spec[key + "|meta"]
    if ndim <= 2:
        assert int(O.n_steps_default(ndim, shape, scale, dx_min, tw)) == int(nd)
    spec = O.make_spec(scale, dx_min, shape, tw, ndim, n)
    assert spec.n_steps == int(nn) and spec.s_max == s_max and spec.dx_min_sq == dxsq
    assert np.array_equal(spec.p, golden_spec[key + "|p"])


@pytest.mark.parametrize("name", [n for n in MG.case_names() if n != "REGULAR/config1"])
def test_oracle_matches_imported_reference(name, golden_generated):
    """fp64, bit-exact, incl. MOM5U/T (untested upstream), NaN-on-land, batches, f32 promotion."""
    grid, fields, gv, fk = MG.build_case(name)
    with np.errstate(divide="ignore"):
        if fk is None:
            res = _stack(O.make_laplacian(grid, gv)(*fields))
        else:
            spec = O.make_spec(fk["filter_scale"], fk["dx_min"], fk["filter_shape"])
            if len(fields) == 2:
                res = np.stack(O.filter_func_vec(spec, grid, *fields, gv))
            else:
                res = O.filter_func(spec, grid, fields[0], gv)
    want = golden_generated[name]
    assert res.dtype == want.dtype and res.shape == want.shape
    assert np.array_equal(res, want, equal_nan=True)


def test_oracle_config1(golden_generated):
    """BASELINE config 1: REGULAR 512x512 f64, Gaussian filter_scale 4, n_steps 16."""
    f = T.random_field((512, 512), 100)
    res = O.filter_func(O.make_spec(4.0, 1.0, "GAUSSIAN", n_steps=16), "REGULAR", f, {})
    assert np.array_equal(res[::8, ::8], golden_generated["REGULAR/config1/probe"])
    np.testing.assert_allclose([res.sum(), (res * res).sum(), np.abs(res).max()],
                               golden_generated["REGULAR/config1/sums"], rtol=1e-13)


def test_oracle_error_contract():
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (16, 24))
    bad = dict(gv, kappa_w=gv["kappa_w"].copy())
    bad["kappa_w"][9, 7] = 2.0
    with pytest.raises(ValueError, match=r"There are kappa_.*"):
        O.make_laplacian("IRREGULAR_WITH_LAND", bad)
    bad = dict(gv, kappa_w=np.full_like(f, 0.5), kappa_s=np.full_like(f, 0.5))
    with pytest.raises(ValueError, match=r"At least one place*"):
        O.make_laplacian("IRREGULAR_WITH_LAND", bad)
    for grid in ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", "TRIPOLAR_POP_WITH_LAND"):
        f, gv = T.tripolar_unit_case(grid, (16, 24))
        bad = dict(gv, wet_mask=gv["wet_mask"].copy())
        bad["wet_mask"][0, 10] = 1
        with pytest.raises(AssertionError, match=r"Wet mask requires .*"):
            O.make_laplacian(grid, bad)
    f, gv = T.tripolar_unit_case("TRIPOLAR_POP_WITH_LAND", (16, 24))
    for nm in ("dxn", "dyn"):
        bad = dict(gv)
        bad[nm] = gv[nm].copy()
        bad[nm][-1, 3] = 10
        with pytest.raises(AssertionError, match=rf"Northernmost row of {nm} .*"):
            O.make_laplacian("TRIPOLAR_POP_WITH_LAND", bad)


def test_oracle_vs_reference_fullsize_fixture():
    """BASELINE config 2 (REGULAR_WITH_LAND 2400 x 3600, n_steps 11) through the oracle against the probes and checksums
    the imported reference produced at full size (make_golden.py --fullsize); ~10 s.  The longer cases of that fixture
    are checked against the GPU path in tests/test_gpu_fullsize.py."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fullsize.npz")
    with np.load(path) as z:
        probe, sums, meta, p = (z["cfg2_n11/" + k] for k in ("probe", "sums", "meta", "p"))
    wl = T.baseline_workload(2, T.BASELINE_SHAPE, scale=10.0)
    spec = O.make_spec(wl["fk"]["filter_scale"], wl["fk"]["dx_min"], wl["fk"]["filter_shape"])
    assert spec.n_steps == int(meta[0]) == 11 and np.array_equal(np.asarray(spec.p), p)
    res = O.filter_func(spec, wl["grid"], wl["fields"][0], wl["grid_vars"])
    jj, ii = T.probe_points(T.BASELINE_SHAPE)
    assert np.array_equal(res[jj, ii], probe.reshape(-1))
    assert np.array_equal(np.array([res.sum(), (res * res).sum(), np.abs(res).max(), 0.0]), sums)


@pytest.mark.parametrize("grid", MG.GRIDBATCHED_GRIDS)
def test_oracle_grid_variables_with_leading_dims(grid):
    """wet_mask(z, y, x) / kappa(z, y, x) against fields (2, z, y, x): the reference's kernels roll along the last two
    axes only and numpy broadcasts the rest (upstream kernels.py:163-187 under filter.py:478-486); fixture from the
    imported reference (make_golden.py --gridbatched)."""
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_gridbatched.npz")) as z:
        want_l, want_f = z[f"{grid}/lap/gridbatched"], z[f"{grid}/gauss/gridbatched"]
    fields, gv, fk = MG.build_gridbatched_case(grid)
    with np.errstate(all="ignore"):
        assert np.array_equal(_stack(O.make_laplacian(grid, gv)(*fields)), want_l)
        spec = O.make_spec(fk["filter_scale"], fk["dx_min"], "GAUSSIAN")
        if len(fields) == 2:
            got = np.stack(O.filter_func_vec(spec, grid, *fields, gv))
        else:
            got = O.filter_func(spec, grid, fields[0], gv)
    assert np.array_equal(got, want_f)
