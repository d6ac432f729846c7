"""The backward (Clenshaw) evaluation of the filter polynomial (csrc/gcmf_ringc_impl.hpp; the default for the f64 flux-form grid
types): same polynomial as the reference's forward recurrence (filter.py:162-212), two state planes instead of three.  Against
the oracle (the reference's recurrence restated): <= 1e-12 relative, identical NaN pattern -- incl. NaN on land, NaN / inf in
wet cells (the in-kernel redo with nan_to_num), batches, area weighting; identical bits however the levels are cut into
launches; and the forward path for everything it does not cover."""
import re

import numpy as np
import pytest

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu

SCALAR_F64 = ["REGULAR", "REGULAR_AREA_WEIGHTED", "REGULAR_WITH_LAND", "REGULAR_WITH_LAND_AREA_WEIGHTED", "IRREGULAR_WITH_LAND",
              "MOM5U", "MOM5T", "TRIPOLAR_POP_WITH_LAND", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"]   # tripolar: + k_fold_band on the seam rows


def _case(grid, shape, n_steps, nanland=False, nanwet=None, nb=1, fshape="TAPER"):
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
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape[fshape], grid_type=GridType[grid],
                     grid_vars=gv)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    return flt, plan, f, want


@pytest.mark.parametrize("grid", SCALAR_F64)
@pytest.mark.parametrize("n_steps,kwargs", [
    (5, {}), (8, {}), (11, {}), (16, {}), (24, dict(fshape="GAUSSIAN")), (63, {}), (21, dict(nanland=True)),
    (21, dict(nanwet=[np.nan])), (17, dict(nanwet=[np.nan, np.nan])), (15, dict(nb=3, nanland=True)),
])
def test_backward_evaluation_matches_the_reference_recurrence(grid, n_steps, kwargs):
    # (+-inf in a wet cell is not a case here: nan_to_num turns it into +-DBL_MAX in the stencil and what overflows where then
    # depends on the order of the operations -- garbage around the cell in the reference too; the forward path keeps the
    # reference's order for the regular / land-mask kinds, which stay on it by default, tests/test_gpu_parity.py)
    flt, plan, f, want = _case(grid, (150, 384), n_steps, **kwargs)
    try:
        plan.set_tuning(multi_s=8, clenshaw=2)          # every scalar kind (the default covers the flux kinds only)
        assert plan.clenshaw_cut(n_steps) and sum(plan.clenshaw_cut(n_steps)) == n_steps
        plan.ring_fallbacks()
        got = flt.apply(f)
        assert "k_ringc" in plan.last_kernel()
        nfb = plan.ring_fallbacks()
    finally:
        plan.set_tuning(multi_s=8, clenshaw=2)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-12 * np.nanmax(np.abs(want))
    if "nanwet" in kwargs and grid not in ("REGULAR", "REGULAR_AREA_WEIGHTED"):
        # the strips around the cells were redone inside the kernel (nan_to_num on the stencil operands): a few of the ~76 strips x 3 launches
        # (k_ringcz, round 6: 2-row strips in pairs, a whole workgroup of four redoes together -- 120 of 296 waves x 3 launches)
        assert 0 < nfb < 200
    elif not (grid.startswith("MOM5") and kwargs.get("nanland")):   # MOM5: land cells with an open face keep their NaN in the state
        assert nfb == 0


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "REGULAR_WITH_LAND", "REGULAR", "TRIPOLAR_POP_WITH_LAND"])
def test_same_bits_however_the_levels_are_cut(grid):
    """strip height and workgroup order change which wave computes what, not the arithmetic of a cell"""
    flt, plan, f, want = _case(grid, (260, 520), 29, nanland=True)
    outs = []
    try:
        for strip, xcd in ((0, 1), (24, 0), (40, 1)):
            plan.set_tuning(multi_s=8, strip_rows=strip, xcd_remap=xcd, clenshaw=2)
            outs.append(flt.apply(f))
            assert "k_ringc" in plan.last_kernel()
    finally:
        plan.set_tuning(multi_s=8, strip_rows=0, xcd_remap=1, clenshaw=2)
    assert np.array_equal(outs[0], outs[1], equal_nan=True) and np.array_equal(outs[0], outs[2], equal_nan=True)


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5U", "MOM5T"])
@pytest.mark.parametrize("n_steps", [9, 17, 27, 63])
@pytest.mark.parametrize("kwargs", [dict(nanland=True), dict(nanwet=[np.nan, np.nan]), dict(nb=2)])
def test_nine_levels_per_launch_give_the_same_bits(grid, n_steps, kwargs):
    """Round 5: whole f64 flux-form grids without a tripole seam run up to NINE levels per k_ringc launch when that saves a launch
    (63 = 7 x 9, 27 = 3 x 9, 17 = 9 + 8, 9 = one launch; csrc/gcmf_ringc_flux9.hip).  Same arithmetic per level: the same bits as the cut
    into launches of 5..8 (n = 9: as the forward recurrence within tolerance -- 9 cannot be cut otherwise), and the oracle."""
    flt, plan, f, want = _case(grid, (200, 392), n_steps, **kwargs)
    try:
        plan.set_option("ringc9", 1)
        cut9 = plan.clenshaw_cut(n_steps)
        assert 9 in cut9 and len(cut9) == -(-n_steps // 9), cut9
        plan.last_kernel()
        got9 = flt.apply(f)
        kern9 = plan.last_kernel()
        assert "k_ringc<double, 2, 9" in kern9 or "k_ringcz<double, 9" in kern9, kern9   # (short strips: zipped pairs, round 6)
        plan.set_option("ringc9", 0)
        cut8 = plan.clenshaw_cut(n_steps)
        assert 9 not in cut8
        plan.last_kernel()
        got8 = flt.apply(f)
    finally:
        plan.set_option("ringc9", 1)
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(got9), np.isnan(want))
    assert np.abs(got9[ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()
    if n_steps != 9:
        assert "k_ringc" in plan.last_kernel() and len(cut8) == len(cut9) + 1
        assert np.array_equal(got9, got8, equal_nan=True)


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5U", "MOM5T", "TRIPOLAR_POP_WITH_LAND"])
@pytest.mark.parametrize("dt", ["f8", "f4"])
@pytest.mark.parametrize("kwargs", [dict(nanland=True), dict(nanwet=[np.nan, np.nan]), dict(nb=3)])
def test_opposite_marches_give_the_same_bits(grid, dt, kwargs):
    """Odd strips of the backward flux kernels march upwards (csrc/gcmf_ringc_impl.hpp: neighbouring strips meet at their common
    boundary, so the ghost rows they re-read are still in the caches): the same instruction stream on mirrored rows, and a flux
    sum (fe - fw) + (fn - fs) that is symmetric under the exchange -- every cell gets the same bits whichever way its strip marched,
    on closed, periodic and tripolar grids, through the nan_to_num redo of a strip, in batches, in f64 and f32."""
    flt, plan, f, want = _case(grid, (260, 520), 29, **kwargs)
    if dt == "f4":    # (an f32 plan is keyed by f32 grid variables)
        import warnings
        f = f.astype(np.float32)
        gv4 = {k: np.asarray(v).astype("f4") for k, v in flt.grid_vars.items()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flt = Filter(filter_scale=flt.filter_scale, dx_min=flt.dx_min, n_steps=29, filter_shape=FilterShape.TAPER, grid_type=GridType[grid],
                         grid_vars=gv4, evaluation="backward")   # (f32 scalar fields: backward only when asked for, round 5)
        plan = ALL_KERNELS[GridType[grid]](**gv4)._plan(_lib.F32, (260, 520))
    outs = []
    try:
        for strip, zz in ((0, 0), (0, 1), (24, 1), (24, 0), (37, 1), (70, 1), (70, 0)):   # (70: whole-period strips, k_ringc; shorter: early exits, k_ringcs)
            plan.set_tuning(multi_s=8, strip_rows=strip, xcd_remap=1, clenshaw=2, zigzag=zz)
            outs.append(flt.apply(f))
            assert "k_ringc" in plan.last_kernel()
    finally:
        plan.set_tuning(multi_s=8, strip_rows=0, xcd_remap=1, clenshaw=2, zigzag=1)
    for o in outs[1:]:
        assert np.array_equal(outs[0], o, equal_nan=True)
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(outs[0]), np.isnan(want))
    assert np.abs(outs[0][ok] - want[ok]).max() <= (1e-12 if dt == "f8" else 1e-4) * np.abs(want[ok]).max()


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5U", "TRIPOLAR_POP_WITH_LAND", "REGULAR_WITH_LAND", "REGULAR", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"])
@pytest.mark.parametrize("shape,nb", [((300, 520), 16), ((260, 520), 5), ((96, 1100), 7), ((1200, 260), 3), ((64, 300), 41)])
def test_packed_batches_give_the_same_bits(grid, shape, nb):
    """Round 6 (csrc/gcmf_ringc_impl.hpp: ringc_walk): the fields of a batch are cut as ONE column of nb x rows rows per window, a wave
    walks a run that may cross from one field into the next -- two marches with their own warm-up rows.  The same rows, the same
    arithmetic: bit-identical with whole strips per field (option "pack_batch" 0), with NaN on land, NaN / inf in wet cells of some fields
    (the redo of one segment), on periodic, closed and tripolar grids, and at forced strip heights that put the cuts everywhere."""
    flt, plan, f, want = _case(grid, shape, 21, nanland=True, nb=nb)
    f = f.copy()
    land = np.isnan(f[0])
    wet = np.argwhere(~land)
    j, i = wet[len(wet) // 3]
    f[1, j, i] = np.nan
    j, i = wet[(2 * len(wet)) // 3]
    f[nb - 1, j, i] = np.inf
    outs, kernels = [], []
    try:
        for pack in (0, 1):
            plan.set_option("pack_batch", pack)
            plan.last_kernel()
            with np.errstate(all="ignore"):
                outs.append(flt.apply(f))
            kernels.append((plan.last_kernel(), plan.last_kernel_geometry()))
    finally:
        plan.set_option("pack_batch", 1)
    assert "k_ringc" in kernels[0][0] and "k_ringc" in kernels[1][0], kernels
    assert kernels[0][1]["grid"].endswith(f"x{nb}"), kernels                        # one grid row per field ...
    assert np.array_equal(outs[0], outs[1], equal_nan=True), kernels
    clean = [k for k in range(nb) if k not in (1, nb - 1)]
    ok = ~np.isnan(want[clean])
    assert np.array_equal(np.isnan(outs[1][clean]), np.isnan(want[clean]))
    assert np.abs(outs[1][clean][ok] - want[clean][ok]).max() <= 1e-12 * np.abs(want[clean][ok]).max()


def test_packed_batches_are_what_a_slab_batch_runs():
    """... and the packed form is what the launcher picks where whole strips tile the wave slots badly: 16 fields on a 300-row grid."""
    flt, plan, f, want = _case("IRREGULAR_WITH_LAND", (300, 3600), 24, nb=16)
    plan.last_kernel()
    got = flt.apply(f)
    geom = plan.last_kernel_geometry()
    assert "k_ringc" in plan.last_kernel() and geom["grid"].endswith("x1"), (plan.last_kernel(), geom)   # ... or ONE for the packed column
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def test_default_is_backward_for_flux_kinds_and_forward_for_the_rest():
    """(The name is round 3's.)  Round 4: the default evaluates EVERY f64 scalar kind backwards -- the land-mask / REGULAR kinds gain
    12-20 % from the fused arithmetic and stay within 1e-14 of numpy; Filter(evaluation="reference") is the bit-exact escape."""
    for grid, backward in (("IRREGULAR_WITH_LAND", True), ("MOM5U", True), ("REGULAR_WITH_LAND", True), ("REGULAR", True),
                           ("TRIPOLAR_POP_WITH_LAND", True), ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", True)):
        flt, plan, f, want = _case(grid, (120, 256), 16)
        got = flt.apply(f)
        assert ("k_ringc" in plan.last_kernel()) == backward, (grid, plan.last_kernel())
        assert bool(plan.clenshaw_cut(16)) == backward
        assert np.nanmax(np.abs(got - want)) <= 1e-12 * np.nanmax(np.abs(want))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = Filter(filter_scale=flt.filter_scale, dx_min=flt.dx_min, n_steps=16, filter_shape=flt.filter_shape, grid_type=flt.grid_type,
                         grid_vars=flt.grid_vars, evaluation="reference").apply(f)
        assert "k_ringc" not in plan.last_kernel(), (grid, plan.last_kernel())
        assert np.nanmax(np.abs(ref - want)) <= 1e-12 * np.nanmax(np.abs(want))
    # polynomial lengths that cannot be cut into launches of 5..8, f32 state: the forward recurrence
    flt, plan, f, want = _case("IRREGULAR_WITH_LAND", (120, 256), 9)
    assert plan.clenshaw_cut(9) == [9] and plan.clenshaw_cut(63) == [9] * 7 and plan.clenshaw_cut(65) == [9] + [8] * 7   # (round 5: nine levels where that saves a launch)
    try:
        plan.set_option("ringc9", 0)
        assert plan.clenshaw_cut(9) == [] and plan.clenshaw_cut(4) == [] and plan.clenshaw_cut(10) == [5, 5] and plan.clenshaw_cut(63) == [8] * 7 + [7]
        got = flt.apply(f)
        assert "k_ringc" not in plan.last_kernel()
    finally:
        plan.set_option("ringc9", 1)
    assert np.nanmax(np.abs(got - want)) <= 1e-12 * np.nanmax(np.abs(want))
    # f32 state (round 5): forward by default -- the reference's own scheme (f32 T_k, f64 running sum), bit for bit on the REGULAR /
    # land-mask kinds; backward (k_ringc<float>, four cells per lane: all f32, 15-45 x less accurate) only when asked for
    for grid in ("IRREGULAR_WITH_LAND", "REGULAR_WITH_LAND", "REGULAR"):
        f32, gv = T.scalar_case(grid, (120, 256))
        gv = {k: v.astype("f4") for k, v in gv.items()}
        dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F32, (120, 256))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flts = {ev: Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=16, grid_type=GridType[grid], grid_vars=gv, evaluation=ev)
                    for ev in ("auto", "reference", "backward")}
        outs = {}
        for ev, backward in (("auto", False), ("reference", False), ("backward", True)):
            outs[ev] = flts[ev].apply(f32.astype("f4"))
            assert (re.search(r"k_ringcs?<float", plan.last_kernel()) is not None) == backward, (grid, ev, plan.last_kernel())
        assert plan.clenshaw_cut(16) == []
        assert np.array_equal(outs["auto"], outs["reference"], equal_nan=True)
        assert np.nanmax(np.abs(outs["backward"] - outs["auto"])) <= 1e-5 * np.nanmax(np.abs(outs["auto"]))
        try:
            plan.set_option("clenshaw_f32", 1)     # the plan-wide switch (GCMF_CLENSHAW_F32=1): what the slab drivers and the row blocks follow
            assert plan.clenshaw_cut(16) != []
            got = flts["auto"].apply(f32.astype("f4"))
            assert re.search(r"k_ringcs?<float", plan.last_kernel()) and np.array_equal(got, outs["backward"], equal_nan=True)
        finally:
            plan.set_option("clenshaw_f32", 0)


@pytest.mark.parametrize("dt,nlev,n_steps", [("f4", 1, 9), ("f4", 5, 16), ("f4", 7, 23), ("f8", 1, 11), ("f8", 4, 16), ("f4", 50, 44)])
def test_cgrid_backward_evaluation(dt, nlev, n_steps):
    """VECTOR_C_GRID (the default): against the oracle in f64 (f32 state: the SURVEY 8d gate 1e-4, measured <= 3e-6), NaN pattern,
    f64 result dtype, and a level filtered alone or in a batch gives the same bits."""
    import torch
    import warnings
    shape = (96, 160) if nlev < 50 else (64, 128)
    gv = {k: v.astype(dt) for k, v in T.vector_grid_vars("VECTOR_C_GRID", shape).items()}
    u = np.stack([T.random_field(shape, 42 + 2 * l).astype(dt) for l in range(nlev)])
    v = np.stack([T.random_field(shape, 43 + 2 * l).astype(dt) for l in range(nlev)])
    u[nlev // 2, 5, 7] = np.nan
    dx = T.grid_dx_min("VECTOR_C_GRID", gv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=10 * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_C_GRID, grid_vars=gv)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        wu, wv = O.filter_func_vec(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "VECTOR_C_GRID",
                                   u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
    plan = ALL_KERNELS[GridType.VECTOR_C_GRID](**gv)._plan(_lib.dtype_code(dt), shape)
    gu, gw = flt.apply_to_vector(u, v)
    # (batched f32 levels: k_cgrid_ring, round 5 -- the same bits; single levels and f64 state: k_cgrid_stream2c)
    assert ("k_cgrid_ring<" if (dt == "f4" and nlev > 1 and n_steps >= 4) else "k_cgrid_stream2c<") in plan.last_kernel(), plan.last_kernel()
    assert gu.dtype == np.float64 and gw.dtype == np.float64
    tol = 1e-4 if dt == "f4" else 1e-12
    for g, w in ((gu, wu), (gw, wv)):
        assert np.array_equal(np.isnan(g), np.isnan(w))
        assert np.nanmax(np.abs(g - w)) <= tol * np.nanmax(np.abs(w))
    if nlev > 1:
        l = nlev - 1
        a1, b1 = flt.apply_to_vector(u[l:l + 1], v[l:l + 1])
        assert np.array_equal(a1[0], gu[l], equal_nan=True) and np.array_equal(b1[0], gw[l], equal_nan=True)


@pytest.mark.parametrize("grid", ["TRIPOLAR_POP_WITH_LAND", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"])
@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("n_steps,nb", [(16, 1), (21, 3)])
def test_tripole_seam_rows_with_nan_in_wet_cells(grid, backward, n_steps, nb):
    """k_fold_band (the top S rows of every blocked launch on a tripolar plan, reference kernels.py:33-40): NaN in wet cells ON the
    seam row, on its mirror partner's side and in the ghost rows below the band -- the stencil sees nan_to_num, the cell keeps its
    NaN (filter.py:166-175) -- in the forward (bit-identical with single steps) and the backward form."""
    shape = (70, 384)
    flt, plan, f, want = _case(grid, shape, n_steps, nanland=True, nb=nb)
    f = f.copy()
    _, gv = T.scalar_case(grid, shape)
    wet = gv["wet_mask"] != 0
    spots = [(shape[0] - 1, 37), (shape[0] - 1, shape[1] - 1 - 36), (shape[0] - 2, 200), (shape[0] - 9, 120), (shape[0] - 14, 300)]
    spots = [(j, i) for j, i in spots if wet[j, i]]
    assert len(spots) >= 3
    for j, i in spots:
        f[..., j, i] = np.nan
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
    try:
        plan.set_tuning(multi_s=1)
        single = flt.apply(f)
        plan.set_tuning(multi_s=8, clenshaw=2 if backward else 0)
        got = flt.apply(f)
        assert ("k_ringc" in plan.last_kernel()) == backward
    finally:
        plan.set_tuning(multi_s=8, clenshaw=2)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got[..., spots[0][0], spots[0][1]]).all()
    assert np.nanmax(np.abs(got - want)) <= 1e-12 * np.nanmax(np.abs(want))
    if not backward:
        assert np.array_equal(got, single, equal_nan=True)


def _cgrid_f32_case(shape, nlev, n_steps, scale):
    import warnings
    gv = {k: v.astype("f4") for k, v in T.vector_grid_vars("VECTOR_C_GRID", shape).items()}
    u = np.stack([T.random_field(shape, 42 + 2 * l).astype("f4") for l in range(nlev)])
    v = np.stack([T.random_field(shape, 43 + 2 * l).astype("f4") for l in range(nlev)])
    dx = T.grid_dx_min("VECTOR_C_GRID", gv)
    flts = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ev in ("auto", "reference"):
            flts[ev] = Filter(filter_scale=scale * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_C_GRID, grid_vars=gv,
                              evaluation=ev)
    fs = flts["auto"].filter_spec
    spec = O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq)
    with np.errstate(all="ignore"):
        ref = O.filter_func_vec(spec, "VECTOR_C_GRID", u, v, gv)       # the reference's own f32 path: f32 T_k, f64 fbar
        truth = O.filter_func_vec(spec, "VECTOR_C_GRID", u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
    plan = ALL_KERNELS[GridType.VECTOR_C_GRID](**gv)._plan(_lib.F32, shape)
    return flts, plan, u, v, ref, truth


def _rel2(a, b):
    return max(float(np.abs(x - y).max() / np.abs(y).max()) for x, y in zip(a, b))


@pytest.mark.parametrize("n_steps,scale", [(44, 40), (63, 57), (98, 90), (125, 114)])
def test_cgrid_f32_precision_policy(n_steps, scale):
    """float32 C-grid fields.  The reference carries T_k in f32 and the running sum in f64; the default here
    (evaluation="auto": Clenshaw's recurrence in Reinsch's form, everything in f32) must be AT LEAST as close to f64 arithmetic as the
    reference's OWN f32 path is (measured 0.55-0.6 x its error; DESIGN.md 3.4 has the table), growing linearly with n_steps like it -- and
    Filter(evaluation="reference") must run the reference's scheme (forward kernel, f64 running sum) without any env var."""
    flts, plan, u, v, ref, truth = _cgrid_f32_case((96, 160), 8, n_steps, scale)
    e_ref = _rel2(ref, truth)                       # what f32 state costs the reference itself: 3e-6 (n 44) ... 1.3e-5 (n 125)
    got = flts["auto"].apply_to_vector(u, v)
    assert "k_cgrid_ring<float" in plan.last_kernel()
    fwd = flts["reference"].apply_to_vector(u, v)
    assert "k_cgrid_ringf<float" in plan.last_kernel()       # (round 6: the forward scheme's own static-ring kernel; until then k_cgrid_stream2<float, double>)
    assert got[0].dtype == np.float64 and fwd[0].dtype == np.float64
    e_auto, e_fwd = _rel2(got, truth), _rel2(fwd, truth)
    print(f"n_steps {n_steps}: error against f64 arithmetic -- reference's f32 path {e_ref:.2e}, evaluation='reference' {e_fwd:.2e}, "
          f"'auto' (backward, f32) {e_auto:.2e}; 'auto' against the reference's f32 result {_rel2(got, ref):.2e}")
    assert e_fwd <= 1.5 * e_ref + 1e-6              # the same scheme as the reference: the same error
    assert e_auto <= 0.8 * e_ref                    # the default: closer to f64 arithmetic than the reference's f32 path (measured 0.55-0.6 x)
    assert e_auto <= 1e-5 and _rel2(got, ref) <= 2e-5   # absolute ceilings at the longest polynomial of the tutorials (SURVEY 8d gate: 1e-4)


@pytest.mark.parametrize("n_steps,scale", [(44, 40), (98, 90)])
def test_bgrid_f32_precision_policy(n_steps, scale):
    """float32 B-grid fields: the default is the reference's own scheme (forward recurrence, f64 running sum), reproduced BIT FOR BIT;
    Filter(evaluation="backward") (Reinsch's form, all f32) is 10 % faster and stays within 4 x of the error the reference's f32 path has
    against f64 arithmetic (measured 2.1-3.4 x; plain Clenshaw was 5.6-7.3 x)."""
    import warnings
    shape, nlev = (96, 160), 6
    gv = {k: v.astype("f4") for k, v in T.vector_grid_vars("VECTOR_B_GRID", shape).items()}
    u = np.stack([T.random_field(shape, 42 + 2 * l).astype("f4") for l in range(nlev)])
    v = np.stack([T.random_field(shape, 43 + 2 * l).astype("f4") for l in range(nlev)])
    dx = T.grid_dx_min("VECTOR_B_GRID", gv)
    flts = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ev in ("auto", "reference", "backward"):
            flts[ev] = Filter(filter_scale=scale * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_B_GRID, grid_vars=gv, evaluation=ev)
    fs = flts["auto"].filter_spec
    spec = O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq)
    with np.errstate(all="ignore"):
        ref = O.filter_func_vec(spec, "VECTOR_B_GRID", u, v, gv)
        truth = O.filter_func_vec(spec, "VECTOR_B_GRID", u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
    plan = ALL_KERNELS[GridType.VECTOR_B_GRID](**gv)._plan(_lib.F32, shape)
    for ev in ("auto", "reference"):
        fwd = flts[ev].apply_to_vector(u, v)
        assert "k_bgrid_stream2<float" in plan.last_kernel(), (ev, plan.last_kernel())
        assert np.array_equal(fwd[0], ref[0]) and np.array_equal(fwd[1], ref[1])
    got = flts["backward"].apply_to_vector(u, v)
    assert "k_bgrid_stream2c<float" in plan.last_kernel()
    e_ref, e_back = _rel2(ref, truth), _rel2(got, truth)
    print(f"n_steps {n_steps}: error against f64 arithmetic -- reference's f32 path = the default {e_ref:.2e}, evaluation='backward' {e_back:.2e}")
    assert e_back <= 4.0 * e_ref and e_back <= 1.5e-5


def test_evaluation_option_reaches_the_plan_for_scalar_grids_too():
    import warnings
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (120, 256))
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, (120, 256))
    outs = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ev, kern in (("auto", "k_ringc"), ("reference", "k_ring<")):
            flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=16, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv, evaluation=ev)
            outs[ev] = flt.apply(f)
            assert kern in plan.last_kernel(), (ev, plan.last_kernel())
        with pytest.raises(ValueError, match="evaluation must be one of"):
            Filter(filter_scale=4.0, dx_min=1.0, evaluation="fast")
    assert np.abs(outs["auto"] - outs["reference"]).max() <= 1e-13 * np.abs(outs["reference"]).max()


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5U", "MOM5T", "TRIPOLAR_POP_WITH_LAND", "REGULAR_WITH_LAND", "REGULAR",
                                  "REGULAR_WITH_LAND_AREA_WEIGHTED", "REGULAR_AREA_WEIGHTED", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"])
@pytest.mark.parametrize("n_steps,kwargs", [(16, {}), (24, dict(nanland=True)), (63, {}), (21, dict(nanwet=True)), (15, dict(nb=3, nanland=True))])
def test_f32_flux_kinds_backward_evaluation(grid, n_steps, kwargs):
    """f32 state on the flux-form grids can run k_ringc<float> (four cells per lane, no f64 running-sum ring;
    Filter(evaluation="backward") since round 5, when the default for f32 scalar fields went back to the reference's scheme).  Against the
    oracle's f64 arithmetic within the f32-state gate (1e-4; measured ~1e-6), f64 result dtype like the reference's
    promotion, NaN pattern incl. NaN in a wet cell (the in-kernel redo), and Filter(evaluation="reference") takes the forward kernels."""
    import warnings
    shape = (150, 512)
    f, gv = T.scalar_case(grid, shape)
    nb = kwargs.get("nb", 1)
    if nb > 1:
        f = np.stack([f + 0.1 * i for i in range(nb)])
    if "wet_mask" not in gv:       # (round 4: the REGULAR kinds run k_ringc<float> too; no land there, and NaN spreads -- kernels.py:113-121)
        if kwargs.get("nanland") or kwargs.get("nanwet"):
            pytest.skip("no land / no nan_to_num on REGULAR")
        land = np.zeros(shape, bool)
    else:
        land = gv["wet_mask"] == 0
    if kwargs.get("nanland"):
        f = np.where(land, np.nan, f)
    if kwargs.get("nanwet"):
        f = f.copy()
        j, i = np.argwhere(~land)[len(np.argwhere(~land)) // 2]
        f[..., j, i] = np.nan
    f4 = f.astype("f4")
    gv4 = {k: v.astype("f4") for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv4) if O.DIMENSIONAL[grid] else 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv4,
                     evaluation="backward")
        ref = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv4,
                     evaluation="reference")
        dflt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv4)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f4.astype("f8"),
                             {k: v.astype("f8") for k, v in gv4.items()})
    plan = ALL_KERNELS[GridType[grid]](**gv4)._plan(_lib.F32, shape)
    plan.ring_fallbacks()
    got = flt.apply(f4)
    assert re.search(r"k_ringcs?<float", plan.last_kernel()), plan.last_kernel()   # (short strips of the flux kinds: the early-exit form)
    nfb = plan.ring_fallbacks()
    fwd = ref.apply(f4)
    assert "k_ringc" not in plan.last_kernel()
    auto = dflt.apply(f4)     # round 5: f32 scalar fields take the reference's scheme by default (15-45 x closer to f64 arithmetic)
    assert "k_ringc" not in plan.last_kernel() and np.array_equal(auto, fwd, equal_nan=True)
    assert got.dtype == np.float64 and fwd.dtype == np.float64
    for g in (got, fwd):
        assert np.array_equal(np.isnan(g), np.isnan(want))
        assert np.nanmax(np.abs(g - want)) <= 1e-4 * np.nanmax(np.abs(want))
    assert np.nanmax(np.abs(got - want)) <= 2.5 * np.nanmax(np.abs(fwd - want)) + 2e-6 * np.nanmax(np.abs(want))   # same accuracy class as the reference's scheme
    if kwargs.get("nanwet"):
        assert nfb > 0
    elif not (grid.startswith("MOM5") and kwargs.get("nanland")):
        assert nfb == 0


@pytest.mark.parametrize("dt,nlev,n_steps", [("f8", 1, 11), ("f8", 4, 16), ("f8", 7, 23), ("f4", 1, 9), ("f4", 5, 16), ("f4", 12, 21)])
def test_bgrid_backward_evaluation_is_an_option(dt, nlev, n_steps):
    """VECTOR_B_GRID (reference kernels.py:702-840) is bit-exact with numpy on the forward recurrence, which therefore stays its default;
    GCMF_CLENSHAW=2 / set_tuning(clenshaw=2) evaluates it backwards too (k_bgrid_stream2c: no fbar planes, fused multiply-adds) --
    against the oracle in f64 arithmetic <= 1e-12 (f32 state: the 1e-4 gate), f64 result dtype, NaN pattern, batch == single level."""
    import warnings
    shape = (96, 160)
    gv = {k: v.astype(dt) for k, v in T.vector_grid_vars("VECTOR_B_GRID", shape).items()}
    u = np.stack([T.random_field(shape, 42 + 2 * l).astype(dt) for l in range(nlev)])
    v = np.stack([T.random_field(shape, 43 + 2 * l).astype(dt) for l in range(nlev)])
    u[nlev // 2, 5, 7] = np.nan
    dx = T.grid_dx_min("VECTOR_B_GRID", gv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=10 * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_B_GRID, grid_vars=gv,
                     evaluation="auto" if dt == "f8" else "backward")   # (f32 B-grid fields: backward only when asked for, round 5)
    fs = flt.filter_spec
    spec = O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq)
    with np.errstate(all="ignore"):
        wu, wv = O.filter_func_vec(spec, "VECTOR_B_GRID", u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
        ru, rv = O.filter_func_vec(spec, "VECTOR_B_GRID", u, v, gv)          # the reference's own path for this dtype
    plan = ALL_KERNELS[GridType.VECTOR_B_GRID](**gv)._plan(_lib.dtype_code(dt), shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fref = Filter(filter_scale=10 * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_B_GRID, grid_vars=gv, evaluation="reference")
    fu, fv = fref.apply_to_vector(u, v)                                       # evaluation="reference": forward, bit-exact with numpy
    assert "k_bgrid_stream2<" in plan.last_kernel()
    assert np.array_equal(fu, ru, equal_nan=True) and np.array_equal(fv, rv, equal_nan=True)
    try:
        gu, gw = flt.apply_to_vector(u, v)                                    # f64: the default since round 4: backward
        assert "k_bgrid_stream2c<" in plan.last_kernel(), plan.last_kernel()
        if nlev > 1:
            l = nlev - 1
            a1, b1 = flt.apply_to_vector(u[l:l + 1], v[l:l + 1])
            assert np.array_equal(a1[0], gu[l], equal_nan=True) and np.array_equal(b1[0], gw[l], equal_nan=True)
    finally:
        plan.set_tuning(multi_s=8)
    assert gu.dtype == np.float64 and gw.dtype == np.float64
    tol = 1e-4 if dt == "f4" else 1e-12
    for g, w in ((gu, wu), (gw, wv)):
        assert np.array_equal(np.isnan(g), np.isnan(w))
        assert np.nanmax(np.abs(g - w)) <= tol * np.nanmax(np.abs(w))


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5T", "TRIPOLAR_POP_WITH_LAND"])
@pytest.mark.parametrize("nb,n_steps,kw", [(1, 16, {}), (3, 29, dict(nanland=True)), (2, 21, dict(nanwet=True)), (1, 63, dict(f32out=True))])
def test_f32_flux_forward_ring_kernel(grid, nb, n_steps, kw):
    """Round 5: f32 scalar fields run the reference's forward scheme by default, and on the flux kinds that is now k_ring<float, double,
    K_FLUX, S, FIRST, 2> (csrc/gcmf_ring_flux_f32.hip: two cells per lane; 509 -> 815 G on IRREGULAR 2400 x 3600).  Same bits as the general
    kernel it replaces (k_flux_multi2, plan option ring_flux_f32 = 0), through NaN on land and NaN in a wet cell (the strip is redone by
    the general march with the ring kernel's window geometry), and against the oracle's f32 path."""
    import warnings
    shape = (150, 512)
    f, gv = T.scalar_case(grid, shape)
    land = gv["wet_mask"] == 0
    if nb > 1:
        f = np.stack([f + 0.1 * i for i in range(nb)])
    if kw.get("nanland"):
        f = np.where(land, np.nan, f)
    if kw.get("nanwet"):
        f = f.copy()
        j, i = np.argwhere(~land)[len(np.argwhere(~land)) // 3]
        f[..., j, i] = np.nan
    f4 = f.astype("f4")
    gv4 = {k: v.astype("f4") for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv4)
    plan = ALL_KERNELS[GridType[grid]](**gv4)._plan(_lib.F32, shape)
    lap = ALL_KERNELS[GridType[grid]](**gv4)
    run = (lambda: lap._run([f4], spec=flt.filter_spec, out_f32=True)[0]) if kw.get("f32out") else (lambda: flt.apply(f4))
    try:
        plan.set_option("ring_flux_f32", 0)
        want = run()
        assert "k_ring<" not in plan.last_kernel()
        plan.set_option("ring_flux_f32", 1)
        got = run()
        kern = plan.last_kernel()
        assert re.search(r"k_ring<float, (double|float), 2, \d, (true|false), 2>", kern) or "k_fold_band" in kern or "k_land_fix" in kern, kern
    finally:
        plan.set_option("ring_flux_f32", 1)
    assert got.dtype == (np.float32 if kw.get("f32out") else np.float64)
    assert np.array_equal(got, want, equal_nan=True)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        ref = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f4, gv4)    # the reference's own f32 path
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) <= (2e-5 if kw.get("f32out") else 2e-6) * np.nanmax(np.abs(ref))
