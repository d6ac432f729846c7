"""One large host field: the row-block pipeline (gcm_filters_amd/host_blocks.py) must return exactly what the one-plan
path returns (same kernels on row ranges of slab plans), for every scalar kind incl. the tripole fold, NaN on land, f32."""
import numpy as np
import pytest

from gcm_filters_amd import Filter, FilterShape, GridType, host_blocks, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS, clear_plan_cache
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def eager_blocks(monkeypatch):
    monkeypatch.setattr(host_blocks, "MIN_CELLS", 1)
    monkeypatch.setattr(host_blocks, "BUILD_AFTER_CALLS", 0)
    clear_plan_cache()
    yield
    clear_plan_cache()


@pytest.mark.parametrize("grid,shape,dt,n_steps,nblocks", [
    ("IRREGULAR_WITH_LAND", (520, 384), "f8", 19, 3), ("REGULAR_WITH_LAND", (400, 512), "f8", 24, 2),
    ("REGULAR", (384, 260), "f8", 16, 3), ("REGULAR_AREA_WEIGHTED", (300, 256), "f8", 11, 2),
    ("REGULAR_WITH_LAND_AREA_WEIGHTED", (512, 256), "f4", 17, 4), ("TRIPOLAR_POP_WITH_LAND", (420, 256), "f8", 18, 3),
    ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (390, 264), "f8", 9, 2), ("MOM5T", (300, 256), "f8", 13, 2),
    ("IRREGULAR_WITH_LAND", (512, 512), "f4", 21, 4),
])
def test_row_blocks_equal_one_plan(grid, shape, dt, n_steps, nblocks, eager_blocks, monkeypatch):
    f, gv = T.scalar_case(grid, shape)
    land = gv["wet_mask"] == 0 if "wet_mask" in gv else np.zeros(shape, bool)
    f = np.where(land, np.nan, f).astype(dt)
    gv = {k: v.astype(dt) for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    flt = Filter(filter_scale=3.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER,
                 grid_type=GridType[grid], grid_vars=gv)
    monkeypatch.setenv("GCMF_HOST_BLOCKS", "0")
    want = flt.apply(f)
    monkeypatch.setenv("GCMF_HOST_BLOCKS", str(nblocks))
    clear_plan_cache()
    got = flt.apply(f)
    got2 = flt.apply(f[None])[0]     # a leading dimension of one takes the same route
    assert got.dtype == want.dtype == np.float64
    assert np.array_equal(got, want, equal_nan=True), float(np.nanmax(np.abs(got - want)))
    assert np.array_equal(got2, want, equal_nan=True)
    with np.errstate(all="ignore"):
        ref = O.filter_func(O.make_spec(3.0 * dx, dx, "TAPER", n_steps=n_steps), grid, f, gv)
    err = np.nanmax(np.abs(got - ref)) / np.nanmax(np.abs(ref))
    assert err <= (1e-4 if dt == "f4" else 1e-11)


def test_row_blocks_are_built_lazily_and_really_used(monkeypatch):
    """The pipeline costs K extra plans: it appears with the third single-field host call on a plan, and from then on the
    blocked launches run on the block plans (the parent plan sees no launch)."""
    from gcm_filters_amd import _lib
    monkeypatch.setattr(host_blocks, "MIN_CELLS", 1)
    monkeypatch.delenv("GCMF_HOST_BLOCKS", raising=False)
    clear_plan_cache()
    shape = (600, 512)
    grid = "IRREGULAR_WITH_LAND"
    f, gv = T.scalar_case(grid, shape)
    dx = T.grid_dx_min(grid, gv)
    flt = Filter(filter_scale=4.0 * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType[grid], grid_vars=gv)
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    outs = []
    for call in range(4):
        plan.last_kernel()
        outs.append(flt.apply(f))
        ran_on_parent = plan.last_kernel() != ""
        assert ran_on_parent == (call < host_blocks.BUILD_AFTER_CALLS), call
    pipes = plan.__dict__["_host_blocks"]["pipes"]
    assert list(pipes) == [flt.n_steps] and pipes[flt.n_steps].nblocks == host_blocks.choose_blocks(shape[0], flt.n_steps) >= 2
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    # a batch, a device tensor and a vector field never take this route
    import torch
    assert np.array_equal(flt.apply(np.stack([f, f]))[1], outs[0])
    assert np.array_equal(flt.apply(torch.from_numpy(f).cuda()).cpu().numpy(), outs[0])
    clear_plan_cache()
    assert "_host_blocks" not in plan.__dict__     # closing the plan closed the block plans
