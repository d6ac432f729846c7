"""k_cgrid_ring (csrc/gcmf_cgrid_ring.hip): the static-ring, packed-f32 form of the backward C-grid kernel for batched f32 levels
(BASELINE config 5; reference stencil gcm_filters/kernels.py:630-696 inside the recurrence of filter.py:217-291).  It must give the
SAME BITS as k_cgrid_stream2c (same operation order per level), which the other C-grid tests pin to the oracle and the reference
vectors -- with NaN in wet cells (the per-row NaN masks), +-inf (the in-kernel redo with the full nan_to_num), ragged windows and
strips, batches that do not fill their last workgroup, slabs -- and match the oracle itself."""
import warnings

import numpy as np
import pytest

from gcm_filters_amd import Filter, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


def _case(shape, nlev, n_steps, scale=10.0, seed=0):
    gv = {k: v.astype("f4") for k, v in T.vector_grid_vars("VECTOR_C_GRID", shape).items()}
    u = np.stack([T.random_field(shape, 42 + 2 * l + seed).astype("f4") for l in range(nlev)])
    v = np.stack([T.random_field(shape, 43 + 2 * l + seed).astype("f4") for l in range(nlev)])
    dx = T.grid_dx_min("VECTOR_C_GRID", gv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=scale * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_C_GRID, grid_vars=gv)
    plan = ALL_KERNELS[GridType.VECTOR_C_GRID](**gv)._plan(_lib.F32, shape)
    return flt, plan, u, v, gv


def _both(flt, plan, u, v, smax=4, strip_rows=0, ncarry=0):
    try:
        plan.set_option("cgrid_ring", 0)
        plan.last_kernel()
        ref = flt.apply_to_vector(u, v)
        assert "k_cgrid_stream2c<" in plan.last_kernel()
        plan.set_option("cgrid_ring", 1)
        plan.set_option("cgrid_ring_smax", smax)
        plan.set_option("cgrid_ring_ncarry", ncarry)
        if strip_rows:
            plan.set_tuning(multi_s=8, strip_rows=strip_rows)
        got = flt.apply_to_vector(u, v)
        assert "k_cgrid_ring<" in plan.last_kernel(), plan.last_kernel()
    finally:
        plan.set_option("cgrid_ring", 1)
        plan.set_option("cgrid_ring_smax", 6)
        plan.set_option("cgrid_ring_ncarry", 0)
        plan.set_tuning(multi_s=8, strip_rows=0)
    return ref, got


@pytest.mark.parametrize("shape,nlev", [((96, 160), 8), ((64, 256), 12), ((33, 132), 2), ((120, 124), 5), ((48, 64), 50), ((25, 520), 4),
                                        ((7, 8), 3)])
@pytest.mark.parametrize("n_steps", [8, 13, 44])
@pytest.mark.parametrize("smax", [4, 5, 6])
def test_same_bits_as_stream2c(shape, nlev, n_steps, smax):
    flt, plan, u, v, gv = _case(shape, nlev, n_steps)
    u[nlev // 2, 5 % shape[0], 7 % shape[1]] = np.nan     # NaN in wet cells: the stencil sees 0, the cell keeps its NaN
    v[0, shape[0] - 1, shape[1] - 1] = np.nan
    v[nlev - 1, 0, 0] = np.nan
    if shape[0] < smax + 2:
        pytest.skip("fewer rows than a launch is deep")
    ref, got = _both(flt, plan, u, v, smax)
    for r, g in zip(ref, got):
        assert g.dtype == np.float64
        assert np.array_equal(r, g, equal_nan=True), (shape, nlev, n_steps, np.nanmax(np.abs(r - g)))
    assert np.isnan(got[0][nlev // 2, 5 % shape[0], 7 % shape[1]])


@pytest.mark.parametrize("smax", [4, 5, 6])
def test_round5_form_only_the_last_level_carries(smax):
    """"cgrid_ring_ncarry" = 1: the levels below the last REBUILD their previous row's three scaled copies from the raw row, its NaN
    masks and the coefficient slot of the row before (round 5's register budget); the default since round 6 carries them.  Same operands,
    same bits."""
    flt, plan, u, v, gv = _case((96, 160), 8, 23)
    u[3, 11, 17] = np.nan
    ref, got = _both(flt, plan, u, v, smax, ncarry=1)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g, equal_nan=True)


@pytest.mark.parametrize("smax", [5, 6])
@pytest.mark.parametrize("shape,nlev,n_steps", [((96, 160), 8, 13), ((40, 300), 5, 9)])
def test_inf_takes_the_redo_pass(shape, nlev, n_steps, smax):
    """+-inf in a wet cell: nan_to_num clamps it to +-FLT_MAX in the stencil (kernels.py:651-652); the workgroups that meet one redo
    their strip with the full nan_to_num at every level and give what k_cgrid_stream2c gives."""
    flt, plan, u, v, gv = _case(shape, nlev, n_steps)
    u[1, 20, 33] = np.inf
    v[nlev - 1, 3, 150 % shape[1]] = -np.inf
    u[0, 9, 9] = np.nan
    with np.errstate(all="ignore"):
        ref, got = _both(flt, plan, u, v, smax)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g, equal_nan=True)


def test_inf_that_appears_inside_a_launch_takes_the_redo_pass_too():
    """A field near FLT_MAX overflows to +-inf INSIDE a launch (nothing non-finite is delivered): the watch on the last level's row sends
    the strip to the redo pass, where every level clamps like k_cgrid_stream2c's nan_to_num (advisor, round 5)."""
    flt, plan, u, v, gv = _case((96, 160), 8, 13)
    u[2, 40:44, 60:64] = 3.0e38
    v[5, 10, 100] = -3.2e38
    with np.errstate(all="ignore"):
        ref, got = _both(flt, plan, u, v, 6)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g, equal_nan=True)


@pytest.mark.parametrize("strip_rows", [16, 20, 31])
def test_same_bits_however_the_strips_are_cut(strip_rows):
    flt, plan, u, v, gv = _case((150, 260), 6, 21)
    ref, got = _both(flt, plan, u, v, 6, strip_rows)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g, equal_nan=True)


def test_against_the_oracle():
    shape, nlev, n_steps = (96, 160), 9, 44
    flt, plan, u, v, gv = _case(shape, nlev, n_steps)
    u[3, 50, 70] = np.nan
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        wu, wv = O.filter_func_vec(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "VECTOR_C_GRID",
                                   u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
    plan.last_kernel()
    gu, gw = flt.apply_to_vector(u, v)
    assert "k_cgrid_ring<" in plan.last_kernel()
    for g, w in ((gu, wu), (gw, wv)):
        assert np.array_equal(np.isnan(g), np.isnan(w))
        assert np.nanmax(np.abs(g - w)) <= 1e-5 * np.nanmax(np.abs(w))   # (f32 state: SURVEY 8d's gate is 1e-4)


# ---- k_cgrid_ringf (csrc/gcmf_cgrid_ringf.hip): the same structure for the reference's forward recurrence with an f64 running sum --------
def _both_forward(u, v, gv, n_steps, scale=10.0, smax=5, strip_rows=0):
    shape = u.shape[-2:]
    dx = T.grid_dx_min("VECTOR_C_GRID", gv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=scale * dx, dx_min=dx, n_steps=n_steps, grid_type=GridType.VECTOR_C_GRID, grid_vars=gv, evaluation="reference")
    plan = ALL_KERNELS[GridType.VECTOR_C_GRID](**gv)._plan(_lib.F32, shape)
    try:
        plan.set_option("cgrid_ring", 0)
        plan.set_tuning(multi_s=smax, strip_rows=strip_rows)
        plan.last_kernel()
        ref = flt.apply_to_vector(u, v)
        assert "k_cgrid_stream2<float, double" in plan.last_kernel(), plan.last_kernel()
        plan.set_option("cgrid_ring", 1)
        got = flt.apply_to_vector(u, v)
        assert "k_cgrid_ringf<" in plan.last_kernel(), plan.last_kernel()
    finally:
        plan.set_option("cgrid_ring", 1)
        plan.set_tuning(multi_s=8, strip_rows=0)
    return ref, got


@pytest.mark.parametrize("shape,nlev", [((96, 160), 8), ((64, 256), 12), ((33, 132), 2), ((120, 124), 5), ((48, 64), 50), ((25, 520), 4), ((7, 8), 3)])
@pytest.mark.parametrize("n_steps", [9, 13, 44])
@pytest.mark.parametrize("smax", [4, 5])
def test_forward_ring_same_bits_as_stream2(shape, nlev, n_steps, smax):
    """Filter(evaluation="reference") on batched f32 levels: k_cgrid_ringf against k_cgrid_stream2<float, double> (which the other tests
    pin to single steps and to the reference's vectors), NaN in wet cells included."""
    if shape[0] < smax + 2:
        pytest.skip("fewer rows than a launch is deep")
    flt, plan, u, v, gv = _case(shape, nlev, n_steps)
    u[nlev // 2, 5 % shape[0], 7 % shape[1]] = np.nan
    v[0, shape[0] - 1, shape[1] - 1] = np.nan
    ref, got = _both_forward(u, v, gv, n_steps, smax=smax)
    for r, g in zip(ref, got):
        assert g.dtype == np.float64 and np.array_equal(r, g, equal_nan=True), (shape, nlev, n_steps, np.nanmax(np.abs(r - g)))
    assert np.isnan(got[0][nlev // 2, 5 % shape[0], 7 % shape[1]])


@pytest.mark.parametrize("strip_rows", [0, 16, 31])
def test_forward_ring_redo_pass_with_the_running_sum_updated_in_place(strip_rows):
    """+-inf delivered, and values near FLT_MAX that overflow INSIDE a launch: the fast pass stops storing where it sees the first
    non-finite value and the redo pass (full nan_to_num) stores from there on -- gcmf_apply updates the running sum in place, so rows the
    fast pass has already accumulated must not be accumulated again."""
    flt, plan, u, v, gv = _case((150, 260), 6, 21)
    u[1, 20, 33] = np.inf
    v[5, 100, 150] = -np.inf
    u[2, 70:74, 60:64] = 3.0e38
    u[0, 9, 9] = np.nan
    with np.errstate(all="ignore"):
        ref, got = _both_forward(u, v, gv, 21, strip_rows=strip_rows)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g, equal_nan=True)


def test_forward_ring_against_the_oracle():
    shape, nlev, n_steps = (96, 160), 9, 44
    flt, plan, u, v, gv = _case(shape, nlev, n_steps)
    u[3, 50, 70] = np.nan
    ref, got = _both_forward(u, v, gv, n_steps)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        wu, wv = O.filter_func_vec(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "VECTOR_C_GRID",
                                   u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
    for g, w in zip(got, (wu, wv)):
        assert np.array_equal(np.isnan(g), np.isnan(w))
        assert np.nanmax(np.abs(g - w)) <= 1e-5 * np.nanmax(np.abs(w))


def test_unaligned_device_views_take_the_general_kernel():
    """the caller's planes come straight from GCMF_DEVICE_PTRS; k_cgrid_ring / k_cgrid_ringf need them on 16-byte boundaries
    (LDS-direct loads, 16-byte stores).  A contiguous view that starts 4 bytes into a buffer runs k_cgrid_stream2[c] instead -- at most
    five levels per launch -- and gives the same bits."""
    import torch
    shape, nlev = (96, 160), 8
    flt, plan, u, v, gv = _case(shape, nlev, 23)
    n = u.size
    bu, bv = torch.zeros(n + 4, dtype=torch.float32, device="cuda"), torch.zeros(n + 4, dtype=torch.float32, device="cuda")
    du, dv = bu[1: 1 + n].view(u.shape), bv[1: 1 + n].view(v.shape)
    du.copy_(torch.from_numpy(u));  dv.copy_(torch.from_numpy(v))
    assert du.data_ptr() % 16 == 4 and du.is_contiguous()
    au, av = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    # (backward: every launch re-reads the caller's input; forward: only the first launch reads it -- that one takes k_cgrid_stream2, the
    # later ones work on the plan's own aligned planes and stay on k_cgrid_ringf)
    for ev, ring, general in (("auto", "k_cgrid_ring<", "k_cgrid_stream2c<"), ("reference", "k_cgrid_ringf<", "k_cgrid_ringf<")):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f2 = Filter(filter_scale=flt.filter_scale, dx_min=flt.dx_min, n_steps=23, grid_type=GridType.VECTOR_C_GRID, grid_vars=gv, evaluation=ev)
        plan.last_kernel()
        want = f2.apply_to_vector(au, av)
        assert ring in plan.last_kernel(), plan.last_kernel()
        plan.last_kernel()
        got = f2.apply_to_vector(du, dv)
        assert general in plan.last_kernel(), plan.last_kernel()
        for w, g in zip(want, got):
            assert torch.equal(w, g)
