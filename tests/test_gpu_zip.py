"""k_ringcz (round 6, csrc/gcmf_ringc_impl.hpp: ringc_march<ZIP>): short strips of the f64 flux kinds marched in PAIRS away from a shared seam.
The two waves of a pair trade one row per level through LDS instead of warming their levels up over S ghost rows each: the same operands
and operations per cell (reference stencil kernels.py:297-315 / 321-432, recurrence filter.py:162-212 as restated in the oracle), so the
SAME BITS as the plain strips -- on closed and periodic grids, with land, NaN / inf in wet cells (the nan_to_num redo of a whole
workgroup), however the levels are cut, in batches, and on row slabs."""
import warnings

import numpy as np
import pytest

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


def _case(grid, shape, n_steps, nanland=False, nanwet=None, nb=1):
    f, gv = T.scalar_case(grid, shape)
    if nb > 1:
        f = np.stack([f + 0.1 * i for i in range(nb)])
    land = gv["wet_mask"] == 0 if "wet_mask" in gv else np.zeros(shape, bool)
    if nanland:
        f = np.where(land, np.nan, f)
    if nanwet is not None:
        f = f.copy()
        wet = np.argwhere(~land)
        for q, val in enumerate(nanwet):
            j, i = wet[(len(wet) * (q + 1)) // (len(nanwet) + 1)]
            f[..., j, i] = val
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    return flt, plan, f, want


def _both(flt, plan, f):
    outs, kernels = [], []
    try:
        for z in (0, 1):
            plan.set_option("ringc_zip", z)
            plan.last_kernel()
            with np.errstate(all="ignore"):
                outs.append(flt.apply(f))
            kernels.append((plan.last_kernel(), plan.last_kernel_geometry()))
    finally:
        plan.set_option("ringc_zip", 1)
    return outs, kernels


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5U", "MOM5T"])
@pytest.mark.parametrize("shape", [(200, 392), (64, 200), (130, 1100), (721, 300), (37, 128)])
@pytest.mark.parametrize("n_steps,kwargs", [(5, {}), (9, dict(nanland=True)), (17, dict(nanwet=[np.nan, np.inf])), (29, dict(nanland=True, nanwet=[np.nan])),
                                            (63, dict(nanland=True)), (21, dict(nanwet=[-np.inf, np.nan, np.nan]))])
def test_zipped_strips_give_the_same_bits(grid, shape, n_steps, kwargs):
    flt, plan, f, want = _case(grid, shape, n_steps, **kwargs)
    outs, kernels = _both(flt, plan, f)
    if "k_ringc" not in kernels[0][0]:
        pytest.skip("nine levels on a grid of fewer than 64 rows: no backward cut (forward recurrence)")
    assert "k_ringcz" not in kernels[0][0] and "k_ringcz<double" in kernels[1][0], kernels
    assert kernels[1][1]["nstrips"] % 2 == 0, kernels
    assert np.array_equal(outs[0], outs[1], equal_nan=True), kernels
    if not any(isinstance(v, list) and np.isinf(v).any() for v in kwargs.values()):   # (an inf in a wet cell: NaN patterns as the reference's, values not compared)
        ok = ~np.isnan(want)
        assert np.array_equal(np.isnan(outs[1]), np.isnan(want))
        assert np.abs(outs[1][ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()


@pytest.mark.parametrize("smax,n", [(5, 60), (6, 63), (7, 63), (8, 63), (0, 63)])
def test_zipped_strips_at_every_depth(smax, n):
    """Launch depths 5 .. 9 (option "ringc_smax": 60 levels as 12 x 5, 63 as 11 launches of <= 6, 9 x 7, 8 launches of <= 8, 7 x 9): the same
    bits whatever the cut."""
    flt, plan, f, want = _case("IRREGULAR_WITH_LAND", (300, 520), n, nanland=True, nanwet=[np.nan])
    try:
        plan.set_option("ringc_smax", 0)
        plan.set_option("ringc_zip", 0)
        ref = flt.apply(f)
        plan.set_option("ringc_zip", 1)
        plan.set_option("ringc_smax", smax)
        cut = plan.clenshaw_cut(n)
        assert sum(cut) == n and max(cut) == (smax or 9) and min(cut) >= 5, cut
        plan.last_kernel()
        got = flt.apply(f)
        assert "k_ringcz<double" in plan.last_kernel(), plan.last_kernel()
    finally:
        plan.set_option("ringc_smax", 0)
        plan.set_option("ringc_zip", 1)
    assert np.array_equal(ref, got, equal_nan=True), cut
    ok = ~np.isnan(want)
    assert np.abs(got[ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()


@pytest.mark.parametrize("nb", [2, 5])
def test_zipped_strips_in_a_batch(nb):
    """A batch whose fields are not packed into one column (option "pack_batch" 0) runs the zipped pairs of every field side by side
    (gridDim.y = the batch); with NaN in one field only, whose workgroups alone take the redo."""
    flt, plan, f, want = _case("IRREGULAR_WITH_LAND", (260, 520), 27, nanland=True, nb=nb)
    f = f.copy()
    wet = np.argwhere(~np.isnan(f[0]))
    j, i = wet[len(wet) // 2]
    f[1, j, i] = np.nan
    try:
        plan.set_option("pack_batch", 0)
        outs, kernels = _both(flt, plan, f)
    finally:
        plan.set_option("pack_batch", 1)
    assert "k_ringcz<double" in kernels[1][0] and kernels[1][1]["grid"].endswith(f"x{nb}"), kernels
    assert np.array_equal(outs[0], outs[1], equal_nan=True)
    clean = [k for k in range(nb) if k != 1]
    ok = ~np.isnan(want[clean])
    assert np.abs(outs[1][clean][ok] - want[clean][ok]).max() <= 1e-12 * np.abs(want[clean][ok]).max()


def test_depth_of_a_lone_field_on_a_cache_resident_grid():
    """gcmf_api_blocks.hip: clenshaw_cut -- 1080 x 1440 (the reference's tutorial grid) n 63 runs 8 launches of <= 8 levels (13 windows of 112
    columns, strips of 14 rows marching 24) instead of 7 x 9 (14 windows of 108, 15 rows marching 28); 720 x 1440 stays at 7 x 9; batches and
    grids beyond the caches keep the fewest launches."""
    for shape, n, want in (((1080, 1440), 63, [8] * 7 + [7]), ((720, 1440), 63, [9] * 7), ((2400, 3600), 63, [9] * 7)):
        f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
        plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, shape)
        assert plan.clenshaw_cut(n) == want, (shape, plan.clenshaw_cut(n))


def test_zipped_strips_are_the_default_where_they_march_fewer_rows():
    """Policy (gcmf_api.hip: launch_ringc): k_ringcz where it marches at least 10 % fewer rows than the plain / early-exit strips -- 1/4-degree
    grids, the slab of one of eight ranks, and (92 rows against 108) BASELINE config 3; on tripolar plans only where the seam's band runs
    AFTER the launch (short launches: nothing has to fit beside the marching waves), never a packed batch."""
    for grid, shape, zipped in (("IRREGULAR_WITH_LAND", (300, 3600), True), ("IRREGULAR_WITH_LAND", (720, 1440), True),
                                ("TRIPOLAR_POP_WITH_LAND", (300, 520), True), ("TRIPOLAR_POP_WITH_LAND", (2400, 3600), True),   # (the seam inside the launch)
                                ("IRREGULAR_WITH_LAND", (2400, 3600), True)):
        f, gv = T.scalar_case(grid, shape)
        dx = T.grid_dx_min(grid, gv)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=18, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
        plan.last_kernel()
        flt.apply(f)
        assert ("k_ringcz" in plan.last_kernel()) == zipped, (grid, shape, plan.last_kernel())


@pytest.mark.parametrize("grid", ["TRIPOLAR_POP_WITH_LAND", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"])
@pytest.mark.parametrize("backward", [True, False])
@pytest.mark.parametrize("n_steps,kwargs", [(16, {}), (21, dict(nb=3, nanland=True)), (29, dict(nanwet=[np.nan, np.nan]))])
def test_the_seam_band_after_or_beside_the_launch(grid, backward, n_steps, kwargs):
    """Round 6 (csrc/gcmf_api.hip advance_multi, csrc/gcmf_foldband.hip): on short launches the tripole seam's k_fold_band runs AFTER the blocked
    launch in its stream with 1024 threads per tile, on long ones beside it on a side stream with 256 -- the same tile arithmetic: the same
    bits (reference fold: kernels.py:33-40, 517-585), forward and backward evaluation, batches, NaN in wet cells; and with the band out of
    the way the flux kind's strips may be zipped."""
    flt, plan, f, want = _case(grid, (180, 392), n_steps, **kwargs)
    outs, kernels = [], []
    try:
        plan.set_option("zip_fold", 0)           # (the f64 flux kind evaluated backwards would advance the seam inside its launches)
        plan.set_tuning(multi_s=8, clenshaw=2 if backward else 0)
        for cells in (0, 3000000):
            plan.set_option("band_seq_cells", cells)
            plan.last_kernel()
            with np.errstate(all="ignore"):
                outs.append(flt.apply(f))
            kernels.append(plan.last_kernel())
    finally:
        plan.set_option("band_seq_cells", 3000000)
        plan.set_option("zip_fold", 1)
        plan.set_tuning(multi_s=8, clenshaw=2)      # (the plan's default)
    assert np.array_equal(outs[0], outs[1], equal_nan=True), kernels
    if backward and grid == "TRIPOLAR_POP_WITH_LAND" and kwargs.get("nb", 1) == 1:
        assert "k_ringcz" not in kernels[0] and "k_ringcz" in kernels[1], kernels
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(outs[1]), np.isnan(want))
    assert np.abs(outs[1][ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()


@pytest.mark.parametrize("shape", [(180, 392), (300, 520), (97, 256), (260, 1100)])
@pytest.mark.parametrize("n_steps,kwargs", [(8, {}), (16, dict(nanland=True)), (21, dict(nb=3, nanland=True)), (29, dict(nanwet=[np.nan, np.nan])),
                                            (56, dict(nanland=True, nanwet=[np.inf]))])
def test_the_tripole_seam_inside_the_launch(shape, n_steps, kwargs):
    """Round 6 (csrc/gcmf_ringc_impl.hpp: ringc_march<ZIP>, fold): on the f64 flux kind of a tripolar plan (TRIPOLAR_POP_WITH_LAND: reference
    kernels.py:517-585, the fold of :33-40) the top rows of a backward launch are strips that START at the seam, each zipped with the strip of
    its MIRROR window -- the northern neighbour of (ny - 1, i) is (ny - 1, nx - 1 - i) -- so k_fold_band is not launched at all.  Same bits as
    the launches with the band (option "zip_fold" 0), the oracle within 1e-12."""
    flt, plan, f, want = _case("TRIPOLAR_POP_WITH_LAND", shape, n_steps, **kwargs)
    outs, kernels = [], []
    try:
        if kwargs.get("nb", 1) > 1:
            plan.set_option("pack_batch", 0)
        for zf in (0, 1):
            plan.set_option("zip_fold", zf)
            plan.last_kernel()
            with np.errstate(all="ignore"):
                outs.append(flt.apply(f))
            kernels.append((plan.last_kernel(), plan.last_kernel_geometry()))
    finally:
        plan.set_option("zip_fold", 1)
        plan.set_option("pack_batch", 1)
    assert "k_ringcz<double" in kernels[1][0], kernels
    assert np.array_equal(outs[0], outs[1], equal_nan=True), kernels
    if not any(isinstance(v, list) and np.isinf(v).any() for v in kwargs.values()):
        ok = ~np.isnan(want)
        assert np.array_equal(np.isnan(outs[1]), np.isnan(want))
        assert np.abs(outs[1][ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()


@pytest.mark.parametrize("n_steps,kwargs", [(9, {}), (27, dict(nanland=True)), (63, dict(nanland=True, nanwet=[np.nan])), (17, dict(nb=2))])
def test_nine_levels_per_launch_on_tripolar_grids(n_steps, kwargs):
    """With the seam inside the launch no k_fold_band (<= 8 levels) is involved, so whole f64 TRIPOLAR_POP grids take nine levels per launch
    where that saves one (63 = 7 x 9), like the flux grids without a seam.  Same bits as the
    cut into launches of <= 8 (n = 9 cannot be cut otherwise: the forward recurrence, within tolerance)."""
    flt, plan, f, want = _case("TRIPOLAR_POP_WITH_LAND", (200, 392), n_steps, **kwargs)
    nb = kwargs.get("nb", 1)
    try:
        cut9 = plan.clenshaw_cut(n_steps)
        assert 9 in cut9 and len(cut9) == -(-n_steps // 9), cut9
        plan.last_kernel()
        with np.errstate(all="ignore"):
            got9 = flt.apply(f)
        k9 = plan.last_kernel()
        plan.set_option("ringc9", 0)
        with np.errstate(all="ignore"):
            got8 = flt.apply(f)
    finally:
        plan.set_option("ringc9", 1)
    # (gcmf_apply cuts with the batch in hand: a batch that the launcher would rather pack keeps the band and the eights -- 16 fields on a
    # 1/4-degree grid; this small one has its fields zipped side by side)
    assert "k_ringcz<double, 9" in k9, k9
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(got9), np.isnan(want))
    assert np.abs(got9[ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()
    if n_steps != 9:
        assert np.array_equal(got9, got8, equal_nan=True)


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND"])
@pytest.mark.parametrize("shape,nb", [((300, 520), 3), ((260, 1100), 5), ((720, 1440), 6), ((200, 392), 16)])
def test_batches_zipped_or_packed_give_the_same_bits(grid, shape, nb):
    """The launcher weighs the zipped strips of a batch (every field's pairs side by side, gridDim.y = the batch; on tripolar plans with the
    seam inside the launch) against whole strips / the packed column (+ k_fold_band) in rows marched: small batches on 1/4-degree grids take
    the zipped strips (720 x 1440 POP, 6 fields: 1141 -> 611 us), big ones stay packed.  Whatever it picks: the bits of the plain strips."""
    flt, plan, f, want = _case(grid, shape, 21, nanland=True, nb=nb)
    f = f.copy()
    wet = np.argwhere(~np.isnan(f[0]))
    j, i = wet[len(wet) // 3]
    f[nb - 1, j, i] = np.nan
    outs, kernels = [], []
    try:
        for z in (0, 1):
            plan.set_option("ringc_zip", z)
            plan.last_kernel()
            with np.errstate(all="ignore"):
                outs.append(flt.apply(f))
            kernels.append(plan.last_kernel())
    finally:
        plan.set_option("ringc_zip", 1)
    assert "k_ringcz" not in kernels[0] and "k_ringc" in kernels[1], kernels
    assert np.array_equal(outs[0], outs[1], equal_nan=True), kernels
    clean = list(range(nb - 1))
    ok = ~np.isnan(want[clean])
    assert np.abs(outs[1][clean][ok] - want[clean][ok]).max() <= 1e-12 * np.abs(want[clean][ok]).max()
