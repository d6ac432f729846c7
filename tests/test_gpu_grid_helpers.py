"""SURVEY 8f-4 on the HIP path: the tutorial conversions (gcm_filters_amd/grid_helpers.py; reference
docs/examples/example_tripole_grid.ipynb, example_vector_laplacian.ipynb) feed Filter on the MI355X and the result is
the oracle's -- POP file variables -> TRIPOLAR_POP / IRREGULAR / fixed-factor filters, a symmetric MOM6 C-grid -> VECTOR_C_GRID
with the tutorial's fixed-factor kappas."""
import numpy as np
import pytest

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, grid_helpers as H, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


def _pop_file(shape=(120, 256)):
    gv = T.scalar_grid_vars("TRIPOLAR_POP_WITH_LAND", shape)
    kmt = (gv["wet_mask"] * 7).astype(np.int32)
    return kmt, gv["dxe"] * 100, gv["dye"] * 100, gv["dxn"] * 100, gv["dyn"] * 100, gv["tarea"] * 1e4


def _check(grid, gv, f, scale_factor=6.0, fixed=False):
    dx = 1.0 if fixed else H.dx_min_over_ocean(gv["wet_mask"], *[gv[k] for k in gv if k[:2] in ("dx", "dy")])
    flt = Filter(filter_scale=scale_factor * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType[grid], grid_vars=gv)
    got = flt.apply(f)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-11 * np.nanmax(np.abs(want)), grid
    return got


def test_pop_file_variables_through_the_filter():
    kmt, hus, hte, htn, huw, tarea = _pop_file()
    f = T.random_field(kmt.shape, 11)
    f = np.where(kmt > 0, f, np.nan)                         # land is NaN in model output
    pop = H.pop_tripolar_grid_vars(kmt, hus, hte, htn, huw, tarea)
    irr = H.pop_irregular_grid_vars(kmt, hus, hte, htn, huw, tarea)
    a = _check("TRIPOLAR_POP_WITH_LAND", pop, f)
    b = _check("IRREGULAR_WITH_LAND", irr, f)
    # the two operators agree away from the seam (example_tripole_grid.ipynb compares exactly these two filters)
    lo = slice(0, kmt.shape[0] // 2)
    assert np.nanmax(np.abs(a[lo] - b[lo])) <= 1e-6 * np.nanmax(np.abs(b[lo]))
    assert np.nanmax(np.abs(a[-3:] - b[-3:])) > 1e-6
    ff = H.fixed_factor_grid_vars(pop["tarea"], pop["wet_mask"], tripolar=True)
    _check("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", ff, f, fixed=True)
    ffr = H.fixed_factor_grid_vars(pop["tarea"], pop["wet_mask"], tripolar=False)
    _check("REGULAR_WITH_LAND_AREA_WEIGHTED", ffr, f, fixed=True)


def test_symmetric_mom6_cgrid_through_the_vector_filter():
    shape = (96, 160)
    gv = T.vector_grid_vars("VECTOR_C_GRID", shape)
    pad = lambda a, y, x: np.pad(a, ((1 if y else 0, 0), (1 if x else 0, 0)), constant_values=-1.0)
    sym = dict(wet=gv["wet_mask_t"], wet_c=pad(gv["wet_mask_q"], 1, 1), dxT=gv["dxT"], dyT=gv["dyT"],
               dxCu=pad(gv["dxCu"], 0, 1), dyCu=pad(gv["dyCu"], 0, 1), dxCv=pad(gv["dxCv"], 1, 0), dyCv=pad(gv["dyCv"], 1, 0),
               dxBu=pad(gv["dxBu"], 1, 1), dyBu=pad(gv["dyBu"], 1, 1))
    ki, ka, dx_max = H.fixed_factor_kappas(gv["dxCu"], gv["dyCv"])        # the tutorial's "fixed factor" vector filter
    out = H.mom6_cgrid_grid_vars(**sym, symmetric=True, kappa_iso=ki, kappa_aniso=ka)
    out = {k: np.asarray(v) for k, v in out.items()}
    (u, v), _ = T.vector_case("VECTOR_C_GRID", shape)
    flt = Filter(filter_scale=5.0 * dx_max, dx_min=dx_max, grid_type=GridType.VECTOR_C_GRID, grid_vars=out)
    gu, gw = flt.apply_to_vector(u, v)
    fs = flt.filter_spec
    wu, ww = O.filter_func_vec(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "VECTOR_C_GRID", u, v, out)
    for g, w in ((gu, wu), (gw, ww)):
        assert np.abs(g - w).max() <= 1e-11 * np.abs(w).max()


def test_a_view_made_before_caching_can_edit_a_protected_plane_and_full_verification_catches_it():
    """kernels.py (plan cache): planes are write-protected while their plan is cached, but a WRITABLE VIEW created before can still
    edit the buffer (numpy cannot revoke it).  The documented remedy, GCMF_PLAN_CACHE_VERIFY=full, hashes whole planes per call;
    this pins both halves of that statement."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from gcm_filters_amd import Filter, GridType, testing as T
from oracle import gcmf_oracle as O
f, gv = T.scalar_case("REGULAR_WITH_LAND", (64, 96))
backdoor = gv["wet_mask"][:]                       # a writable view, made BEFORE the plan exists
flt = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gv)
a = flt.apply(f)
import os
if os.environ.get("GCMF_PLAN_CACHE", "1") != "0":
    assert not gv["wet_mask"].flags.writeable      # the plane itself is protected while its plan is cached ...
    try:
        gv["wet_mask"][10:30, 40:70] = 0
        raise SystemExit("edit of a protected plane did not raise")
    except ValueError:
        pass
else:
    assert gv["wet_mask"].flags.writeable          # no cache, no protection: every call folds a fresh plan
backdoor[10:30, 40:70] = 0                         # ... the old view is not: this edit lands far from the 256 sampled values or on them
b = flt.apply(f)
want = O.filter_func(O.make_spec(4.0, 1.0), "REGULAR_WITH_LAND", f, {"wet_mask": np.asarray(gv["wet_mask"])})
fresh = np.abs(b - want).max() <= 1e-12 * np.abs(want).max()
print("FRESH" if fresh else "STALE")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = lambda env: subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
    full = run({"GCMF_PLAN_CACHE_VERIFY": "full"})
    assert full.returncode == 0 and "FRESH" in full.stdout, full.stdout + full.stderr      # whole-plane hashing sees the edit
    off = run({"GCMF_PLAN_CACHE": "0"})
    assert off.returncode == 0 and "FRESH" in off.stdout, off.stdout + off.stderr          # so does rebuilding per call (the reference's way)
    dflt = run({})
    assert dflt.returncode == 0 and ("FRESH" in dflt.stdout or "STALE" in dflt.stdout), dflt.stdout + dflt.stderr
    # (sampled verification may or may not notice -- that is the documented hole; the two remedies above must)


@pytest.mark.parametrize("mode", ["verify", "off"])
def test_plan_cache_keyword_lets_a_drop_in_user_edit_the_mask_in_place(mode):
    """the reference folds a fresh Laplacian per call (filter.py:183), so a user may edit wet_mask in place between two
    calls.  Under the default the cached plan write-protects the planes (loud); Filter(plan_cache="verify" | "off") keeps them writable
    and still filters with what the arrays hold NOW."""
    f, gv = T.scalar_case("REGULAR_WITH_LAND", (64, 96))
    flt = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gv, plan_cache=mode)
    a = flt.apply(f)
    assert gv["wet_mask"].flags.writeable
    gv["wet_mask"][10:30, 40:70] = 0
    b = flt.apply(f)
    want = O.filter_func(O.make_spec(4.0, 1.0), "REGULAR_WITH_LAND", f, {"wet_mask": np.asarray(gv["wet_mask"])})
    assert np.abs(b - want).max() <= 1e-12 * np.abs(want).max() and np.abs(a - b).max() > 1e-3
    again = flt.apply(f)
    assert np.array_equal(b, again)


def test_plan_cache_protect_is_the_default_and_says_so():
    from gcm_filters_amd.kernels import clear_plan_cache
    f, gv = T.scalar_case("REGULAR_WITH_LAND", (64, 96))
    flt = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gv)
    assert flt.plan_cache is None
    flt.apply(f)
    assert not gv["wet_mask"].flags.writeable
    with pytest.raises(ValueError, match="read-only"):
        gv["wet_mask"][10, 40] = 0
    clear_plan_cache()
    assert gv["wet_mask"].flags.writeable
    with pytest.raises(ValueError, match="plan_cache"):
        Filter(filter_scale=4.0, dx_min=1.0, plan_cache="sometimes")
