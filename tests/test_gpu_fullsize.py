"""BASELINE-size checks (2400 x 3600) through size-independent properties, plus a bounded oracle comparison.
The oracle needs ~1 s per Laplacian step at this size, so the direct comparison uses a short polynomial; the
full benchmark polynomial (n_steps 63, 8 steps per HBM pass) is pinned to the single-step kernel bit for bit."""
import numpy as np
import pytest

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu
SHAPE = (2400, 3600)


@pytest.fixture(scope="module")
def irregular():
    gv = T.scalar_grid_vars("IRREGULAR_WITH_LAND", SHAPE)
    gv["kappa_w"] = T.smooth_kappa(SHAPE, 11)
    f = T.random_field(SHAPE, 100)
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    return f, gv, dx


def test_bench_workload_blocked_equals_single_steps(irregular):
    f, gv, dx = irregular
    flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER,
                 grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    assert flt.n_steps == 63
    plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, SHAPE)
    try:
        plan.set_tuning(multi_s=1)
        ref = flt.apply(f)
        plan.set_tuning(multi_s=8)
        got = flt.apply(f)
        plan.set_tuning(multi_s=4)
        got4 = flt.apply(f)
    finally:
        plan.set_tuning(multi_s=8)
    assert np.array_equal(ref, got) and np.array_equal(ref, got4)
    assert np.isfinite(got).all() and got.min() > -0.2 and got.max() < 1.2   # stable: dx_min is the true minimum
    area, m = gv["area"], gv["wet_mask"]
    np.testing.assert_allclose((got * area * m).sum(), (f * area * m).sum(), rtol=1e-10)   # integral over the ocean
    assert (got[m == 1] ** 2).sum() < (f[m == 1] ** 2).sum()


def test_linearity_and_constants(irregular):
    f, gv, dx = irregular
    flt = Filter(filter_scale=8 * dx, dx_min=dx, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    g = T.random_field(SHAPE, 7)
    a = flt.apply(np.stack([f, g, 2.5 * f - 0.75 * g, np.full(SHAPE, 3.25)]))
    scale = np.abs(a[2]).max()
    assert np.abs(a[2] - (2.5 * a[0] - 0.75 * a[1])).max() <= 1e-12 * scale
    wet = gv["wet_mask"] == 1
    np.testing.assert_allclose(a[3][wet], 3.25, rtol=1e-12)   # L(const) = 0 on the ocean, p(-1) = 1


def test_vs_oracle_short_polynomial(irregular):
    f, gv, dx = irregular
    f = np.where(gv["wet_mask"] == 0, np.nan, f)
    with pytest.warns(UserWarning):
        flt = Filter(filter_scale=16 * dx, dx_min=dx, n_steps=8, filter_shape=FilterShape.TAPER,
                     grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    got = flt.apply(f)
    spec = O.FilterSpec(8, flt.filter_spec.s_max, np.asarray(flt.filter_spec.p), flt.filter_spec.dx_min_sq)
    want = O.filter_func(spec, "IRREGULAR_WITH_LAND", f, gv)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).sum() == (gv["wet_mask"] == 0).sum()
    ok = ~np.isnan(want)
    assert np.abs(got[ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()


def test_tripolar_pop_fullsize_fold_band():
    gv = T.scalar_grid_vars("TRIPOLAR_POP_WITH_LAND", SHAPE)
    f = T.random_field(SHAPE, 100)
    dx = T.grid_dx_min("TRIPOLAR_POP_WITH_LAND", gv)
    flt = Filter(filter_scale=50 * dx, dx_min=dx, grid_type=GridType.TRIPOLAR_POP_WITH_LAND, grid_vars=gv)
    assert flt.n_steps == 56
    plan = ALL_KERNELS[GridType.TRIPOLAR_POP_WITH_LAND](**gv)._plan(_lib.F64, SHAPE)
    try:
        plan.set_tuning(multi_s=1)
        ref = flt.apply(f)
        plan.set_tuning(multi_s=8)
        got = flt.apply(f)
    finally:
        plan.set_tuning(multi_s=8)
    assert np.array_equal(ref, got)
    np.testing.assert_allclose((got * gv["tarea"] * gv["wet_mask"]).sum(), (f * gv["tarea"] * gv["wet_mask"]).sum(), rtol=1e-10)
