"""GPU parity: the HIP path (through the C ABI) against the reference's goldens and the CPU oracle.

Tolerances (BASELINE.json north_star / SURVEY 8d):
  * fp64:  ||gpu - ref||_inf / ||ref||_inf <= 1e-6, identical NaN pattern (observed ~1e-15);
  * REGULAR / land-mask / B-grid kernels are written in the reference's operation order: bit-exact;
  * fp32 state: <= 1e-4 relative (the reference's own f32 path differs from its f64 path by ~1e-6).
"""
import re

import numpy as np
import pytest

import make_golden as MG
from gcm_filters_amd import Filter, FilterShape, GridType, required_grid_vars, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu

RTOL_F64 = 1e-6
RTOL_F32 = 1e-4
BIT_EXACT = {"REGULAR", "REGULAR_AREA_WEIGHTED", "REGULAR_WITH_LAND", "REGULAR_WITH_LAND_AREA_WEIGHTED",
             "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"}


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN pattern differs"
    fin = ~np.isnan(want)
    scale = np.abs(want[fin]).max() if fin.any() else 1.0
    return 0.0 if scale == 0 else float(np.abs(got[fin] - want[fin]).max() / scale)


def gpu_laplacian(grid, fields, gv):
    lap = ALL_KERNELS[GridType[grid]](**gv)
    res = lap(*fields)
    return np.stack(res) if isinstance(res, tuple) else res


def gpu_filter(grid, fields, gv, fk):
    flt = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]],
                 n_steps=fk.get("n_steps", 0), grid_type=GridType[grid], grid_vars=gv)
    if len(fields) == 2:
        return np.stack(flt.apply_to_vector(*fields))
    return flt.apply(fields[0])


# ---------------------------------------------------------------------------------------------------
# the reference's own known-answer tests (upstream tests/test_kernels_validation.py, test_filter_validation.py)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid", T.REFERENCE_TESTED_GRIDS)
def test_kernel_vs_reference_zarr(grid, golden_zarr):
    if grid in T.VECTOR_GRIDS:
        fields, gv = T.vector_case(grid)
    else:
        f, gv = T.scalar_case(grid)
        fields = (f,)
    res = gpu_laplacian(grid, fields, gv)
    want = golden_zarr[f"test_data_kernels/{grid}"]
    np.testing.assert_allclose(want, res.astype("f4"), rtol=2e-7)
    assert rel_err(res, O.make_laplacian(grid, gv)(*fields) if len(fields) == 1
                   else np.stack(O.make_laplacian(grid, gv)(*fields))) <= 1e-12


@pytest.mark.parametrize("grid", T.REFERENCE_TESTED_GRIDS)
def test_filter_vs_reference_zarr(grid, golden_zarr):
    if grid in T.VECTOR_GRIDS:
        fields, gv = T.vector_case(grid)
    else:
        f, gv = T.scalar_case(grid)
        fields = (f,)
    res = gpu_filter(grid, fields, gv, dict(filter_scale=8.0, dx_min=1.0, filter_shape="GAUSSIAN"))
    want = golden_zarr[f"test_data_filter/{grid}"]
    np.testing.assert_allclose(want, res.astype("f4"), rtol=2e-7, atol=1e-30)


# ---------------------------------------------------------------------------------------------------
# fp64 vectors captured from the imported reference (all 11 grid types x variants)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [n for n in MG.case_names() if n != "REGULAR/config1"])
def test_vs_imported_reference(name, golden_generated):
    grid, fields, gv, fk = MG.build_case(name)
    want = golden_generated[name]
    res = gpu_laplacian(grid, fields, gv) if fk is None else gpu_filter(grid, fields, gv, fk)
    assert res.shape == want.shape
    assert res.dtype == want.dtype, (res.dtype, want.dtype)
    all_f32 = all(f.dtype == np.float32 for f in fields) and all(v.dtype == np.float32 for v in gv.values())
    # f32 field on f64 grids: the reference takes its FIRST differences in f32, we convert the field up front
    mixed = name.endswith("/f32") and not all_f32
    tol = RTOL_F32 if all_f32 else RTOL_F64
    err = rel_err(res, want)
    assert err <= tol, (name, err)
    if not (all_f32 or mixed):
        assert err <= 1e-11, (name, err)  # what fp64 actually delivers
    if grid in BIT_EXACT and not mixed and fk is None:
        assert np.array_equal(res, want, equal_nan=True), name


@pytest.mark.parametrize("grid", sorted(BIT_EXACT))
@pytest.mark.parametrize("kind", ["gauss", "taper", "gauss/nanland", "gauss/batched"])
def test_bit_exact_recurrence(grid, kind, golden_generated):
    """With the reference's own polynomial coefficients the regular-grid filters reproduce the reference
    bit for bit under evaluation="reference" (the forward recurrence: same operation order, no FMA contraction, fused
    prepare/finalize included).  The default (round 4: the polynomial evaluated backwards with fused multiply-adds wherever a
    backward kernel exists) is held to 1e-13 of the same vectors and to the identical NaN pattern."""
    from gcm_filters_amd.filter import FilterSpec, _create_filter_func
    name = f"{grid}/{kind}"
    if name not in golden_generated:
        pytest.skip("no such case")
    g, fields, gv, fk = MG.build_case(name)
    o = O.make_spec(fk["filter_scale"], fk["dx_min"], fk["filter_shape"])  # == reference p (test_oracle_golden)
    cls = ALL_KERNELS[GridType[g]]
    want = golden_generated[name]
    func = _create_filter_func(FilterSpec(o.n_steps, o.s_max, o.p, o.dx_min_sq), cls, evaluation="reference")
    res = func(fields[0], *[gv[k] for k in cls.required_grid_args()])
    assert np.array_equal(res, want, equal_nan=True)
    auto = _create_filter_func(FilterSpec(o.n_steps, o.s_max, o.p, o.dx_min_sq), cls)(fields[0], *[gv[k] for k in cls.required_grid_args()])
    assert np.array_equal(np.isnan(auto), np.isnan(want))
    assert np.nanmax(np.abs(auto - want)) <= 1e-13 * np.nanmax(np.abs(want))


def test_config1_regular_512(golden_generated):
    """BASELINE config 1: REGULAR 512x512 f64, Gaussian filter_scale 4, n_steps 16."""
    f = T.random_field((512, 512), 100)
    flt = Filter(filter_scale=4.0, dx_min=1.0, n_steps=16, grid_type=GridType.REGULAR)
    res = flt.apply(f)
    np.testing.assert_allclose(res[::8, ::8], golden_generated["REGULAR/config1/probe"], rtol=1e-13)
    np.testing.assert_allclose([res.sum(), (res * res).sum(), np.abs(res).max()],
                               golden_generated["REGULAR/config1/sums"], rtol=1e-13)


# ---------------------------------------------------------------------------------------------------
# ragged / tiny / odd shapes against the oracle (scalar path falls back from 16-byte to scalar accesses)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 4), (5, 7), (37, 53), (64, 130), (33, 258), (2, 2)])
@pytest.mark.parametrize("grid", ["REGULAR", "REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "MOM5T", "VECTOR_C_GRID",
                                  "VECTOR_B_GRID", "TRIPOLAR_POP_WITH_LAND", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"])
def test_ragged_shapes(grid, shape):
    if grid.startswith("TRIPOLAR") and shape[1] % 2:
        pytest.skip("tripolar grids need an even nx (the reference raises for odd nx)")
    if shape[0] < 3 and grid.startswith("TRIPOLAR"):
        pytest.skip("degenerate")
    vec = grid in T.VECTOR_GRIDS
    if vec:
        fields, gv = T.vector_case(grid, shape)
    else:
        f, gv = T.scalar_case(grid, shape)
        fields = (f,)
    spec = O.make_spec(4.0, 1.0, "GAUSSIAN")
    with np.errstate(all="ignore"):
        want_l = O.make_laplacian(grid, gv)(*fields)
        want_f = O.filter_func_vec(spec, grid, *fields, gv) if vec else O.filter_func(spec, grid, fields[0], gv)
    want_l = np.stack(want_l) if vec else want_l
    want_f = np.stack(want_f) if vec else want_f
    assert rel_err(gpu_laplacian(grid, fields, gv), want_l) <= 1e-11
    got = gpu_filter(grid, fields, gv, dict(filter_scale=4.0, dx_min=1.0, filter_shape="GAUSSIAN"))
    assert rel_err(got, want_f) <= 1e-9


def test_infinities_follow_nan_to_num():
    """+-inf / NaN inputs go through numpy.nan_to_num (0, +-max) like the reference.  On the land-mask kernel
    the operation order is the reference's, so even the overflow pattern next to an inf is identical."""
    f, gv = T.scalar_case("REGULAR_WITH_LAND", (24, 32))
    f = f.copy()
    f[14, 20] = np.inf
    f[15, 3 + 16] = -np.inf
    f[20, 20] = np.nan
    with np.errstate(all="ignore"):
        want = O.make_laplacian("REGULAR_WITH_LAND", gv)(f)
    got = gpu_laplacian("REGULAR_WITH_LAND", (f,), gv)
    assert np.array_equal(got, want, equal_nan=True)
    # flux-form kernel: metric ratios are pre-folded, so compare away from the overflowing cells
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (24, 32))
    f = f.copy()
    f[20, 20] = np.nan
    f[14, 25] = np.inf
    with np.errstate(all="ignore"):
        want = O.make_laplacian("IRREGULAR_WITH_LAND", gv)(f)
    got = gpu_laplacian("IRREGULAR_WITH_LAND", (f,), gv)
    far = np.ones_like(f, dtype=bool)
    far[13:16, 24:27] = False
    assert np.isfinite(got[far]).all()
    np.testing.assert_allclose(got[far], want[far], rtol=1e-12, atol=1e-300)
    assert not np.isnan(got).any()


# ---------------------------------------------------------------------------------------------------
# restatement of upstream tests/test_kernels.py on the GPU kernels
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid", [g for g in T.SCALAR_GRIDS])
def test_conservation(grid):
    f, gv = T.scalar_case(grid)
    res = gpu_laplacian(grid, (f,), gv)
    area = 1
    if grid not in O.AREA_WEIGHTED:
        area = gv.get("area", gv.get("tarea", gv.get("area_u", gv.get("area_t", 1))))
    np.testing.assert_allclose((area * res).sum(), 0.0, atol=1e-12 if not grid.startswith("MOM5") else 1e-10)


@pytest.mark.parametrize("grid", T.ALL_GRIDS)
def test_required_grid_vars_and_dimensionality(grid):
    assert set(required_grid_vars(GridType[grid])) == set(T.FIXTURE_ARG_ORDER[grid])
    assert list(required_grid_vars(GridType[grid])) == list(O.GRID_ARGS[grid])
    assert ALL_KERNELS[GridType[grid]].is_dimensional == O.DIMENSIONAL[grid]


def test_kappa_validation():
    _, gv = T.scalar_case("IRREGULAR_WITH_LAND")
    cls = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND]
    bad = {k: v.copy() for k, v in gv.items()}
    bad["kappa_w"][99, 225] = 2.0
    with pytest.raises(ValueError, match=r"There are kappa_.*"):
        cls(**bad)
    bad["kappa_w"][99, 225] = 1.0
    bad["kappa_s"][99, 225] = 2.0
    with pytest.raises(ValueError, match=r"There are kappa_.*"):
        cls(**bad)
    bad = {k: v.copy() for k, v in gv.items()}
    bad["kappa_w"][:, :] = 0.5
    bad["kappa_s"][:, :] = 0.5
    with pytest.raises(ValueError, match=r"At least one place*"):
        cls(**bad)


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND"])
@pytest.mark.parametrize("direction", ["X", "Y"])
def test_flux(grid, direction):
    """Delta function + metric outliers placed so that a wrong shift direction breaks isotropy
    (upstream tests/test_kernels.py:109-183)."""
    f, gv = T.scalar_case(grid)
    jy, ix = 99, 225
    delta = np.zeros_like(f)
    delta[jy, ix] = 1
    kw = {k: (v if k == "wet_mask" else np.ones_like(f)) for k, v in gv.items()}
    spots = {
        "IRREGULAR_WITH_LAND": {"Y": ("dxs", (jy - 1, slice(None)), (jy + 2, slice(None))),
                                "X": ("dyw", (slice(None), ix - 1), (slice(None), ix + 2))},
        "TRIPOLAR_POP_WITH_LAND": {"Y": ("dxn", (jy - 2, slice(None)), (jy + 1, slice(None))),
                                   "X": ("dye", (slice(None), ix - 2), (slice(None), ix + 1))},
    }
    var, lo, hi = spots[grid][direction]
    pert = np.ones_like(f)
    pert[lo] = 1000
    pert[hi] = 2000
    kw[var] = pert
    d = gpu_laplacian(grid, (delta,), kw)
    np.testing.assert_allclose(d[jy - 1, ix], d[jy + 1, ix], atol=1e-12)
    np.testing.assert_allclose(d[jy, ix - 1], d[jy, ix + 1], atol=1e-12)


@pytest.mark.parametrize("grid", ["TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", "TRIPOLAR_POP_WITH_LAND"])
def test_tripolar_contract(grid):
    f, gv = T.tripolar_unit_case(grid)
    cls = ALL_KERNELS[GridType[grid]]
    bad = {k: v.copy() for k, v in gv.items()}
    bad["wet_mask"][0, 10] = 1
    with pytest.raises(AssertionError, match=r"Wet mask requires .*"):
        cls(**bad)
    if grid == "TRIPOLAR_POP_WITH_LAND":
        bad = {k: v.copy() for k, v in gv.items()}
        bad["dxn"][-1, 3] = 10
        with pytest.raises(AssertionError, match=r"Northernmost row of dxn .*"):
            cls(**bad)
        bad["dxn"][-1, 3] = 1
        bad["dyn"][-1, 3] = 10
        with pytest.raises(AssertionError, match=r"Northernmost row of dyn .*"):
            cls(**bad)
    # exchange across the seam (upstream tests/test_kernels.py:224-245)
    delta = np.zeros_like(f)
    nx = f.shape[1]
    delta[-1, 10] = 1
    d = cls(**gv)(delta)
    np.testing.assert_allclose(d[-2, 10], d[-1, nx - 10 - 1], atol=1e-12)


@pytest.mark.parametrize("grid", T.VECTOR_GRIDS)
def test_solid_body_rotation(grid):
    _, gv = T.vector_case(grid)
    u, v = T.solid_body_rotation()
    lu, lv = ALL_KERNELS[GridType[grid]](**gv)(u, v)
    np.testing.assert_allclose(lu, 0.0, atol=1e-12)
    np.testing.assert_allclose(lv, 0.0, atol=1e-12)
    flt = Filter(filter_scale=5.0, dx_min=1.0, n_steps=10, filter_shape=FilterShape.TAPER,
                 grid_type=GridType[grid], grid_vars=gv)
    fu, fv = flt.apply_to_vector(u, v)
    np.testing.assert_allclose(fu, u, atol=1e-12)
    np.testing.assert_allclose(fv, v, atol=1e-12)


@pytest.mark.parametrize("grid", [g for g in T.SCALAR_GRIDS if not g.startswith("MOM5")])
def test_diffusion_filter_properties(grid):
    """Conservation of the area integral and variance reduction (upstream tests/test_filter.py:114-138)."""
    f, gv = T.scalar_case(grid)
    flt = Filter(filter_scale=3.0, dx_min=1.0, grid_type=GridType[grid], grid_vars=gv)
    out = flt.apply(f)
    area = 1
    for k, v in gv.items():
        if "area" in k:
            area = v
            break
    np.testing.assert_allclose((f * area).sum(), (out * area).sum(), rtol=1e-5)
    assert (out ** 2).sum() < (f ** 2).sum()


def test_nondimensional_invariance():
    f = np.random.default_rng(0).normal(size=(100, 100))
    a = Filter(filter_scale=4, dx_min=1, grid_type=GridType.REGULAR).apply(f)
    b = Filter(filter_scale=8, dx_min=2, grid_type=GridType.REGULAR).apply(f)
    np.testing.assert_allclose(a, b, rtol=1e-5)


def test_batches_are_independent():
    f3 = np.stack([T.random_field((40, 64), s) for s in range(6)]).reshape(2, 3, 40, 64)
    _, gv = T.scalar_case("IRREGULAR_WITH_LAND", (40, 64))
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=6 * dx, dx_min=dx, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    whole = flt.apply(f3)
    for a in range(2):
        for b in range(3):
            assert np.array_equal(whole[a, b], flt.apply(f3[a, b]))
    assert np.array_equal(f3, np.stack([T.random_field((40, 64), s) for s in range(6)]).reshape(2, 3, 40, 64))


def test_device_resident_tensors():
    """torch tensors already in HBM are filtered without a host round trip and come back as tensors."""
    import torch
    f, gv = T.scalar_case("TRIPOLAR_POP_WITH_LAND", (40, 64))
    dx = T.grid_dx_min("TRIPOLAR_POP_WITH_LAND", gv)
    host = Filter(filter_scale=6 * dx, dx_min=dx, grid_type=GridType.TRIPOLAR_POP_WITH_LAND, grid_vars=gv).apply(f)
    tgv = {k: torch.from_numpy(v).cuda() for k, v in gv.items()}
    flt = Filter(filter_scale=6 * dx, dx_min=dx, grid_type=GridType.TRIPOLAR_POP_WITH_LAND, grid_vars=tgv)
    out = flt.apply(torch.from_numpy(f).cuda())
    assert isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float64
    assert np.array_equal(out.cpu().numpy(), host)


class _DLPackArray:
    """A device-array library that is NOT torch (stand-in for cupy, which this image lacks): a DLPack producer over a device buffer,
    with a module-level `from_dlpack` consumer like cupy's."""

    def __init__(self, tensor):
        self._t = tensor

    def __dlpack__(self, stream=None):
        return self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return self._t.dtype


def from_dlpack(obj):   # (what `cupy.from_dlpack` is to cupy: found by gcm_filters_amd through the producer's module)
    import torch
    return _DLPackArray(torch.from_dlpack(obj))


class _CAIArray:
    """A `__cuda_array_interface__` producer without DLPack (numba-style), whose class knows how to consume DLPack."""

    def __init__(self, tensor):
        self._t = tensor

    @property
    def __cuda_array_interface__(self):
        return self._t.__cuda_array_interface__

    @classmethod
    def from_dlpack(cls, obj):
        import torch
        return cls(torch.from_dlpack(obj))


class _OpaqueCAIArray:
    """... and one that offers no way back: the result is the torch tensor (documented in README)."""

    def __init__(self, tensor):
        self._t = tensor

    @property
    def __cuda_array_interface__(self):
        return self._t.__cuda_array_interface__


_OpaqueCAIArray.__module__ = "a_library_without_a_dlpack_consumer"   # (this test module HAS a `from_dlpack`, as cupy's has)


@pytest.mark.parametrize("wrap", [_DLPackArray, _CAIArray, _OpaqueCAIArray])
def test_foreign_device_arrays_come_back_as_their_own_kind(wrap):
    """SURVEY 8 (f)2 / VERDICT r4 item 5: the reference returns cupy arrays for cupy input (gpu_compat.py:5-10).  A DLPack or
    `__cuda_array_interface__` producer that is not torch is filtered in HBM (zero copy in) and handed back through ITS OWN DLPack
    consumer -- module-level `from_dlpack` (cupy's), a `from_dlpack` classmethod, `__array_namespace__().from_dlpack` -- and as a torch
    tensor only when it has none.  Scalar and vector Laplacians, and the bare Laplacian call."""
    import torch
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (48, 64))
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=6 * dx, dx_min=dx, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    want = flt.apply(torch.from_numpy(f).cuda())
    out = flt.apply(wrap(torch.from_numpy(f).cuda()))
    if wrap is _OpaqueCAIArray:
        assert isinstance(out, torch.Tensor) and out.is_cuda
        assert torch.equal(out, want)
    else:
        assert isinstance(out, wrap) and not isinstance(out, torch.Tensor)
        assert torch.equal(out._t, want) and out._t.is_cuda
    lap = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)(wrap(torch.from_numpy(f).cuda()))
    assert isinstance(lap, torch.Tensor if wrap is _OpaqueCAIArray else wrap)
    (u, v), gvv = T.vector_case("VECTOR_C_GRID", (40, 64))
    dxv = T.grid_dx_min("VECTOR_C_GRID", gvv)
    fv = Filter(filter_scale=4 * dxv, dx_min=dxv, grid_type=GridType.VECTOR_C_GRID, grid_vars=gvv)
    wu, wv = fv.apply_to_vector(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())
    gu, gw = fv.apply_to_vector(wrap(torch.from_numpy(u).cuda()), wrap(torch.from_numpy(v).cuda()))
    for g, w in ((gu, wu), (gw, wv)):
        assert isinstance(g, torch.Tensor if wrap is _OpaqueCAIArray else wrap)
        assert torch.equal(g if wrap is _OpaqueCAIArray else g._t, w)


# ---------------------------------------------------------------------------------------------------
# temporal blocking (S recurrence steps per HBM pass) must be bit-identical to S single steps
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid,shape,dt", [
    ("IRREGULAR_WITH_LAND", (96, 160), "f8"), ("IRREGULAR_WITH_LAND", (61, 520), "f8"), ("MOM5U", (40, 64), "f8"),
    ("REGULAR", (50, 258), "f8"), ("REGULAR_WITH_LAND_AREA_WEIGHTED", (70, 300), "f8"),
    ("REGULAR_AREA_WEIGHTED", (33, 64), "f8"), ("IRREGULAR_WITH_LAND", (64, 256), "f4"), ("REGULAR_WITH_LAND", (48, 1032), "f4"),
    ("TRIPOLAR_POP_WITH_LAND", (60, 160), "f8"), ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (41, 264), "f8"),
    ("TRIPOLAR_POP_WITH_LAND", (30, 128), "f4"),
])
@pytest.mark.parametrize("S,strip", [(2, 0), (3, 0), (4, 0), (4, 7), (5, 0), (6, 0), (7, 0), (8, 0), (8, 5)])
def test_temporal_blocking_bit_identical(grid, shape, dt, S, strip):
    from gcm_filters_amd import _lib
    f, gv = T.scalar_case(grid, shape)
    f = np.where(gv.get("wet_mask", np.ones(shape)) == 0, np.nan, f).astype(dt)
    gv = {k: v.astype(dt) for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    cls = ALL_KERNELS[GridType[grid]]
    lap = cls(**gv)
    plan = lap._plan(_lib.dtype_code(dt), shape)
    outs = {}
    for n_steps in (11, 16):
        flt = Filter(filter_scale=2.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER,
                     grid_type=GridType[grid], grid_vars=gv)
        try:
            plan.set_tuning(multi_s=1)
            plan.set_timing(True)
            ref = flt.apply(f)
            n_single = plan.last_timing()[1]
            plan.set_tuning(multi_s=S, strip_rows=strip, clenshaw=0)   # the forward recurrence: bit-identity with single steps
            got = flt.apply(f)
            n_multi = plan.last_timing()[1]
        finally:
            plan.set_tuning(multi_s=8, strip_rows=0, clenshaw=2)
            plan.set_timing(False)
        # the blocked path really ran (tripolar: + one k_fold_band per blocked launch, so S = 2 launches as often as single steps)
        assert n_multi < n_single or (grid.startswith("TRIPOLAR") and S == 2 and n_multi <= n_single), (n_multi, n_single)
        assert np.array_equal(ref, got, equal_nan=True), (grid, S, strip, n_steps, rel_err(got, ref))
        outs[n_steps] = got
    spec = O.make_spec(2.0 * dx, dx, "TAPER", n_steps=16)
    with np.errstate(all="ignore"):
        want = O.filter_func(spec, grid, f, gv)
    assert rel_err(outs[16], want) <= (1e-4 if dt == "f4" else 1e-11)


def test_concurrent_calls_from_threads_and_streams():
    """dask="parallelized" calls filter_func from worker threads: the plan (shared through the cache) must serialise
    them, also when device-resident inputs arrive on different HIP streams."""
    import threading
    import torch
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", (96, 160))
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=8 * dx, dx_min=dx, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    fields = [T.random_field((3, 96, 160), 50 + k) for k in range(8)]
    want = [flt.apply(x) for x in fields]
    got = [None] * len(fields)

    def host_worker(k):
        got[k] = flt.apply(fields[k])

    th = [threading.Thread(target=host_worker, args=(k,)) for k in range(len(fields))]
    [t.start() for t in th]
    [t.join() for t in th]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)

    dev = [torch.from_numpy(x).cuda() for x in fields]
    outs = [None] * len(fields)

    def stream_worker(k):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            outs[k] = flt.apply(dev[k])
        st.synchronize()

    th = [threading.Thread(target=stream_worker, args=(k,)) for k in range(len(fields))]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    for g, w in zip(outs, want):
        assert np.array_equal(g.cpu().numpy(), w)


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "REGULAR_WITH_LAND", "VECTOR_C_GRID", "VECTOR_B_GRID"])
def test_out_f32_option(grid):
    """GCMF_OUT_F32: f32 plans may keep fbar / the output in f32 (opt-in; the default follows NumPy >= 2 and returns
    f64).  Must agree with the f64-accumulated result to f32 accuracy."""
    shape = (64, 256)
    vec = grid in T.VECTOR_GRIDS
    if vec:
        fields, gv = T.vector_case(grid, shape)
    else:
        f, gv = T.scalar_case(grid, shape)
        fields = (f,)
    fields = [x.astype("f4") for x in fields]
    gv = {k: v.astype("f4") for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    flt = Filter(filter_scale=6.0 * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gv)
    lap = ALL_KERNELS[GridType[grid]](*[gv[k] for k in ALL_KERNELS[GridType[grid]].required_grid_args()])
    ref = lap._run(fields, spec=flt.filter_spec)
    got = lap._run(fields, spec=flt.filter_spec, out_f32=True)
    for r, g in zip(ref, got):
        assert r.dtype == np.float64 and g.dtype == np.float32
        assert np.abs(g - r).max() <= 2e-5 * np.abs(r).max()


@pytest.mark.parametrize("shape,nlev,dt", [((40, 64), 4, "f8"), ((33, 130), 8, "f8"), ((64, 256), 12, "f4"), ((25, 520), 4, "f4"),
                                           ((96, 160), 50, "f4"), ((6, 8), 4, "f8"), ((6, 8), 4, "f4"), ((48, 64), 7, "f8"),
                                           ((120, 124), 4, "f4"), ((40, 64), 1, "f8"), ((64, 256), 1, "f4"),
                                           ((33, 132), 2, "f4"), ((7, 8), 1, "f4")])
@pytest.mark.parametrize("grid", ["VECTOR_C_GRID", "VECTOR_B_GRID"])
def test_vector_temporal_blocking_bit_identical(grid, shape, nlev, dt):
    """The vector Laplacians advance S = 2..4 recurrence steps per HBM pass (k_cgrid_stream2 / k_bgrid_stream2);
    results must be bit-identical to single steps, for every split of the step count and every batch size (lock-step
    workgroups of 4 levels, padded), with NaN / inf in the input, and match the oracle."""
    from gcm_filters_amd import _lib
    (u0, v0), gv = T.vector_case(grid, shape)
    rng = np.random.default_rng(5)
    u = np.stack([u0 * (1 + 0.1 * k) + rng.standard_normal(shape) for k in range(nlev)]).astype(dt)
    v = np.stack([v0 * (1 - 0.05 * k) + rng.standard_normal(shape) for k in range(nlev)]).astype(dt)
    u[0, 3, 5] = np.nan
    v[nlev - 1, 1, 2] = np.inf
    gv = {k: x.astype(dt) for k, x in gv.items()}
    dx = T.grid_dx_min(grid, gv)
    lap = ALL_KERNELS[GridType[grid]](**gv)
    plan = lap._plan(_lib.dtype_code(dt), shape)
    blocked = True   # any batch size: workgroups of 4 levels, padded with shadow waves
    for n_steps in (3, 4, 8, 9, 13):
        flt = Filter(filter_scale=2.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER,
                     grid_type=GridType[grid], grid_vars=gv)
        try:
            plan.set_tuning(multi_s=1)
            plan.set_timing(True)
            ref = flt.apply_to_vector(u, v)
            assert plan.last_timing()[1] == n_steps
            for S in (2, 3, 4, 5, 6):
                plan.set_tuning(multi_s=S, clenshaw=0)   # the forward recurrence: bit-identity with single steps
                got = flt.apply_to_vector(u, v)
                n_launch = plan.last_timing()[1]
                if blocked:
                    assert n_launch < n_steps, (n_launch, n_steps, S)
                    if S == 4 and dt == "f4" and n_steps == 8:
                        assert n_launch == 2
                    if S == 2 and n_steps == 8:
                        assert n_launch == 4
                else:
                    assert n_launch == n_steps
                for r, g in zip(ref, got):
                    assert np.array_equal(r, g, equal_nan=True), (shape, nlev, dt, n_steps, S, rel_err(g, r))
            if grid == "VECTOR_C_GRID":   # the default: backward evaluation (k_cgrid_stream2c), same polynomial, other rounding
                plan.set_tuning(multi_s=8, clenshaw=2)
                plan.last_kernel()   # (reading resets: it reports the deepest kernel since the last read)
                gotc = flt.apply_to_vector(u, v)
                assert re.search(r"k_cgrid_(stream2c|ring)<", plan.last_kernel()), plan.last_kernel()
                if dt == "f8":   # (f32: the +-inf cell above becomes +-FLT_MAX in the stencil and what overflows where depends on
                    #               the order of the operations; the f32 backward path is checked on finite / NaN input in
                    #               tests/test_gpu_clenshaw.py and by tools/fuzz_gpu.py --cgrid)
                    for r, g in zip(ref, gotc):
                        assert np.array_equal(np.isnan(r), np.isnan(g))
                        assert rel_err(g, r) <= 1e-13, (shape, nlev, dt, n_steps, rel_err(g, r))
        finally:
            plan.set_tuning(multi_s=8, clenshaw=2)
            plan.set_timing(False)
    spec = O.make_spec(2.0 * dx, dx, "TAPER", n_steps=13)
    with np.errstate(all="ignore"):
        want = O.filter_func_vec(spec, grid, u, v, gv)
    for g, w in zip(got, want):
        assert rel_err(g, w) <= (1e-4 if dt == "f4" else 1e-11)


@pytest.mark.parametrize("grid,dt", [("IRREGULAR_WITH_LAND", "f8"), ("REGULAR_WITH_LAND_AREA_WEIGHTED", "f8"),
                                     ("VECTOR_C_GRID", "f4"), ("VECTOR_B_GRID", "f8"), ("TRIPOLAR_POP_WITH_LAND", "f4")])
def test_pipelined_host_batches(grid, dt, monkeypatch):
    """Host arrays with a batch: chunks stream through two staging slots (upload / filter / download overlapped).
    Forced here with a tiny chunk size; results must equal the one-shot path bit for bit, chunk remainders included."""
    from gcm_filters_amd.kernels import clear_plan_cache
    shape, nb = (40, 64), 7
    vec = grid in T.VECTOR_GRIDS
    gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
    gv = {k: v.astype(dt) for k, v in gv.items()}
    fields = [np.stack([T.random_field(shape, 11 + 7 * c + b) for b in range(nb)]).astype(dt) for c in range(2 if vec else 1)]
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    kw = dict(filter_scale=6.0 * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gv)
    run = (lambda f: f.apply_to_vector(*fields)) if vec else (lambda f: (f.apply(fields[0]),))
    lap_run = lambda: ALL_KERNELS[GridType[grid]](**gv)._run(fields)
    clear_plan_cache()
    monkeypatch.setenv("GCMF_HOST_CHUNK_MB", "0")          # one shot
    want, want_l = run(Filter(**kw)), lap_run()
    clear_plan_cache()
    monkeypatch.setenv("GCMF_HOST_CHUNK_MB", str(2.5 * shape[0] * shape[1] * np.dtype(dt).itemsize / 2**20))  # 2 entries
    got, got_l = run(Filter(**kw)), lap_run()
    clear_plan_cache()
    for g, w in zip(list(got) + list(got_l), list(want) + list(want_l)):
        assert g.dtype == w.dtype and np.array_equal(g, w, equal_nan=True)


@pytest.mark.parametrize("grid", ["REGULAR_WITH_LAND", "REGULAR_WITH_LAND_AREA_WEIGHTED",
                                  "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", "IRREGULAR_WITH_LAND"])
@pytest.mark.parametrize("dt", ["f8", "f4"])
def test_blocked_kernel_nan_and_inf_modes(grid, dt):
    """The blocked kernel runs a wave in one of three modes per row (all finite / NaN flags only / an inf around).
    NaN on land, NaN and +-inf on wet cells, far enough apart that different waves take different modes and switch
    between them along their march: every depth must reproduce the single-step kernel bit for bit (NaN pattern incl.)."""
    from gcm_filters_amd import _lib
    shape = (96, 520)
    f, gv = T.scalar_case(grid, shape)
    f = np.where(gv["wet_mask"] == 0, np.nan, f)
    wet = np.argwhere(gv["wet_mask"] == 1)
    pick = lambda i: tuple(wet[(len(wet) * i) // 7])
    f[pick(1)] = np.nan
    f[pick(3)] = np.inf
    f[pick(5)] = -np.inf
    f = f.astype(dt)
    gv = {k: v.astype(dt) for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.dtype_code(dt), shape)
    flt = Filter(filter_scale=2.0 * dx, dx_min=dx, n_steps=19, filter_shape=FilterShape.TAPER, grid_type=GridType[grid],
                 grid_vars=gv)
    try:
        plan.set_tuning(multi_s=1)
        with np.errstate(all="ignore"):
            ref = flt.apply(f)
        for S in (2, 4, 7, 8):
            plan.set_tuning(multi_s=S, clenshaw=0)
            got = flt.apply(f)
            assert np.array_equal(np.isnan(ref), np.isnan(got)), (grid, dt, S)
            assert np.array_equal(ref, got, equal_nan=True), (grid, dt, S)
    finally:
        plan.set_tuning(multi_s=8, clenshaw=2)
    assert np.isnan(ref[gv["wet_mask"] == 0]).all() and np.isfinite(ref).sum() > ref.size // 3


def test_very_long_batch_of_small_fields():
    """More batch entries than one launch can index (gridDim.y): the batch is cut into launches of 32768."""
    import torch
    shape, nb = (4, 8), 40000
    _, gv = T.scalar_case("REGULAR_WITH_LAND", shape)
    f = np.random.default_rng(3).random((nb,) + shape)
    flt = Filter(filter_scale=4, dx_min=1, grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gv)
    got = flt.apply(torch.from_numpy(f).cuda()).cpu().numpy()
    spec = O.make_spec(4, 1, "GAUSSIAN")
    for b in (0, 32767, 32768, nb - 1):
        want = O.filter_func(spec, "REGULAR_WITH_LAND", f[b], gv)
        np.testing.assert_allclose(got[b], want, rtol=1e-12, atol=1e-15, err_msg=str(b))
    host = flt.apply(f)            # host path (pipelined chunks are capped at the same size)
    assert np.array_equal(host, got)


@pytest.mark.parametrize("grid", ["IRREGULAR_WITH_LAND", "MOM5U", "MOM5T", "TRIPOLAR_POP_WITH_LAND"])
@pytest.mark.parametrize("dt", ["f8", "f4"])
def test_land_kept_out_of_the_state(grid, dt, monkeypatch):
    """Flux kinds: cells whose four faces are closed are zeroed in the state after the first blocked launch and get their
    (neighbour-free) polynomial from k_land_fix at the end.  Must equal the plain path bit for bit -- finite, NaN and
    +-inf values on land, a batch, both step-count parities -- and the oracle within tolerance."""
    from gcm_filters_amd.kernels import clear_plan_cache
    shape = (64, 256)
    f0, gv = T.scalar_case(grid, shape)
    land = gv["wet_mask"] == 0
    rng = np.random.default_rng(9)
    f = np.stack([f0, np.where(land, np.nan, f0), np.where(land, 7.5 * rng.standard_normal(shape), f0)])
    lj, li = np.argwhere(land)[len(np.argwhere(land)) // 2]
    f[2, lj, li] = np.inf
    f = f.astype(dt)
    gv = {k: v.astype(dt) for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv)
    outs = {}
    monkeypatch.setenv("GCMF_CLENSHAW", "0")   # the forward recurrence with and without the land handling: bit for bit
    for z in ("1", "0"):
        monkeypatch.setenv("GCMF_ZERO_LAND", z)
        clear_plan_cache()
        res = []
        for n_steps in (8, 13):
            flt = Filter(filter_scale=2.0 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER,
                         grid_type=GridType[grid], grid_vars=gv)
            with np.errstate(all="ignore"):
                res.append(flt.apply(f))
        outs[z] = res
    clear_plan_cache()
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.isnan(outs["1"][0][1][land]).all() and np.isfinite(outs["1"][0][1][~land]).all()
    spec = O.make_spec(2.0 * dx, dx, "TAPER", n_steps=13)
    with np.errstate(all="ignore"):
        want = O.filter_func(spec, grid, f[:2], gv)
    assert np.array_equal(np.isnan(outs["1"][1][:2]), np.isnan(want))
    assert rel_err(outs["1"][1][:2], want) <= (1e-4 if dt == "f4" else 1e-11)


def test_two_plans_stream_the_same_host_array(monkeypatch):
    """Two threads (dask workers) filter the SAME batched host array with different plans at once: the page-locking of
    the caller's input is reference counted, so the first call to finish does not unlock it under the other's uploads."""
    import threading
    from gcm_filters_amd.kernels import clear_plan_cache
    shape, nb = (96, 256), 24
    monkeypatch.setenv("GCMF_HOST_CHUNK_MB", str(2.0 * shape[0] * shape[1] * 8 / 2**20))  # chunks of two fields
    clear_plan_cache()
    f = np.stack([T.random_field(shape, 300 + b) for b in range(nb)])
    flts = []
    for grid in ("REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND"):
        _, gv = T.scalar_case(grid, shape)
        dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
        flts.append(Filter(filter_scale=6 * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gv))
    want = [flt.apply(f) for flt in flts]
    for _ in range(3):
        got = [None, None]

        def work(i):
            got[i] = flts[i].apply(f)

        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    clear_plan_cache()


def test_in_place_is_rejected():
    """gcmf_apply reads the input again at the end (land cells) and across strips: out == in is an argument error."""
    import torch
    from gcm_filters_amd import _lib
    shape = (40, 64)
    _, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
    lap = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)
    plan = lap._plan(_lib.F64, shape, 0)
    d = torch.zeros(shape, dtype=torch.float64, device="cuda")
    p = np.array([0.5, 0.3, 0.2])
    with pytest.raises(_lib.GcmfError) as e:
        plan.apply(p, 0.25, [d.data_ptr()], [d.data_ptr()], 1, device_ptrs=True)
    assert "alias" in str(e.value)


# ---------------------------------------------------------------------------------------------------
# grid variables with leading (level) dims: one device plan per level (upstream kernels.py:163-187 roll along the last two
# axes only; filter.py:478-486 broadcasts the other dims)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_gridbatched():
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_gridbatched.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("grid", MG.GRIDBATCHED_GRIDS)
@pytest.mark.parametrize("where", ["host", "device"])
def test_grid_variables_with_leading_dims(grid, where, golden_gridbatched):
    import torch
    fields, gv, fk = MG.build_gridbatched_case(grid)
    vec = grid in T.VECTOR_GRIDS
    if where == "device":
        gv = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in gv.items()}
        fields = tuple(torch.from_numpy(f).cuda() for f in fields)
    to_np = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else x
    lap = ALL_KERNELS[GridType[grid]](**gv)
    L = lap(*fields)
    L = np.stack([to_np(x) for x in L]) if vec else to_np(L)
    assert rel_err(L, golden_gridbatched[f"{grid}/lap/gridbatched"]) <= 1e-11
    flt = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], grid_type=GridType[grid], grid_vars=gv)
    got = flt.apply_to_vector(*fields) if vec else flt.apply(fields[0])
    got = np.stack([to_np(x) for x in got]) if vec else to_np(got)
    want = golden_gridbatched[f"{grid}/gauss/gridbatched"]
    assert got.shape == want.shape
    assert rel_err(got, want) <= 1e-11
    # a field WITHOUT the extra leading dim broadcasts the same way: (z, y, x) field, (z, y, x) grid variables
    one = flt.apply_to_vector(*(f[1] for f in fields)) if vec else flt.apply(fields[0][1])
    one = np.stack([to_np(x) for x in one]) if vec else to_np(one)
    assert rel_err(one, want[:, 1] if vec else want[1]) <= 1e-11


def test_kappa_check_spans_all_levels():
    """kernels.py:274-281 tests the whole kappa arrays: a level whose kappa stays below 1 is fine as long as another level
    reaches 1; nothing reaching 1 anywhere, or anything above 1 on any level, is an error."""
    fields, gv, fk = MG.build_gridbatched_case("IRREGULAR_WITH_LAND")
    assert gv["kappa_w"][1].max() < 0.9 and gv["kappa_w"][0].max() == 1.0
    ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)
    bad = dict(gv, kappa_w=0.8 * gv["kappa_w"])
    with pytest.raises(ValueError, match=r"At least one place in the domain must have either kappa_w = 1 or kappa_s = 1.*"):
        ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**bad)
    worse = dict(gv, kappa_w=gv["kappa_w"].copy())
    worse["kappa_w"][2, 5, 5] = 1.5
    with pytest.raises(ValueError, match=r"There are kappa_w values > 1.*"):
        ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**worse)


# ---------------------------------------------------------------------------------------------------
# plan cache: a cached device plan must never serve stale coefficients (the reference rebuilds its Laplacian per call,
# upstream filter.py:183)
# ---------------------------------------------------------------------------------------------------
def test_plan_cache_never_serves_stale_coefficients(monkeypatch):
    from gcm_filters_amd import kernels as K
    shape = (600, 700)                                    # 420 000 cells: the 256-value sample sees 1 cell in 1640
    f, gv = T.scalar_case("REGULAR_WITH_LAND", shape)
    mask = gv["wet_mask"]
    flt = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND, grid_vars=gv)
    before = flt.apply(f)
    j, i = 411, 333                                       # an unsampled, wet cell
    assert mask[j, i] == 1 and (j * shape[1] + i) % max(1, mask.size // 256) != 0
    # 1. an in-place edit of a plane a cached plan was folded from is refused
    assert not mask.flags.writeable
    with pytest.raises(ValueError, match="read-only"):
        mask[j, i] = 0
    # 2. once the plan leaves the cache the array is writable again, and the edit changes the result
    K.clear_plan_cache()
    assert mask.flags.writeable
    mask[j, i] = 0
    after = flt.apply(f)
    assert after[j, i] != before[j, i] and not np.array_equal(after, before)
    want = O.filter_func(O.make_spec(4.0, 1.0), "REGULAR_WITH_LAND", f, gv)
    assert rel_err(after, want) <= 1e-12
    # 3. a writable view that existed before the plan was cached can still reach the buffer: whole-plane hashing catches it
    K.clear_plan_cache()
    base = np.ones(shape)
    base[0, :] = 0
    alias = base[:]                                       # created before any plan exists
    monkeypatch.setattr(K, "_VERIFY_FULL", True)
    flt2 = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR_WITH_LAND, grid_vars={"wet_mask": base})
    r1 = flt2.apply(f)
    alias[j, i] = 0                                       # `alias` kept its flag; `base` is protected
    r2 = flt2.apply(f)
    assert not np.array_equal(r1, r2)
    assert rel_err(r2, O.filter_func(O.make_spec(4.0, 1.0), "REGULAR_WITH_LAND", f, {"wet_mask": base})) <= 1e-12
    K.clear_plan_cache()
    assert base.flags.writeable


def test_regular_spreads_nan_like_the_reference():
    """RegularLaplacian has no nan_to_num (upstream kernels.py:113-121): a NaN in the input spreads one cell per step in
    the 5-point pattern (181 cells after 9 steps).  Blocked (k_ring / general) and single-step schedules."""
    from gcm_filters_amd import _lib
    shape = (96, 160)
    f = T.random_field(shape, 3)
    f[40, 70] = np.nan
    flt9 = Filter(filter_scale=8.0, dx_min=1.0, grid_type=GridType.REGULAR, evaluation="reference")
    assert flt9.n_steps == 9
    fs = flt9.filter_spec
    want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "REGULAR", f, {})
    assert np.isnan(want).sum() == 181
    plan = ALL_KERNELS[GridType.REGULAR]()._plan(_lib.F64, shape)
    try:
        for ms in (1, 4, 8):
            plan.set_tuning(multi_s=ms)
            got = flt9.apply(f)
            assert np.array_equal(np.isnan(got), np.isnan(want)), ms
            assert np.array_equal(got, want, equal_nan=True), ms                  # REGULAR is bit-exact
    finally:
        plan.set_tuning(multi_s=8)
    flt = Filter(filter_scale=8.0, dx_min=1.0, n_steps=21, grid_type=GridType.REGULAR, evaluation="reference")   # 8 + 8 (k_ring) + 5
    fs = flt.filter_spec
    want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "REGULAR", f, {})
    got = flt.apply(f)
    assert "k_ring<double, double, 0, 8, " in plan.last_kernel()
    assert np.array_equal(got, want, equal_nan=True) and np.isnan(got).sum() == 2 * 21 * 22 + 1
    # the default (backward evaluation, k_ringc): the NaN spreads through the same stencil, one cell per level -- same pattern
    got = Filter(filter_scale=8.0, dx_min=1.0, n_steps=21, grid_type=GridType.REGULAR).apply(f)
    assert "k_ringc<double, 0, " in plan.last_kernel()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(np.abs(got - want)) <= 1e-13 * np.nanmax(np.abs(want))


@pytest.mark.parametrize("grid,clenshaw", [("IRREGULAR_WITH_LAND", 1), ("IRREGULAR_WITH_LAND", 0), ("REGULAR_WITH_LAND", 1),
                                           ("REGULAR_WITH_LAND", 2), ("REGULAR", 1)])
def test_ring_kernel_redoes_only_strips_with_non_finite_values(grid, clenshaw):
    """k_ring hands a wave strip to the general kernel when its NaN watch fires.  NaN on land is masked on load and must not
    fire it -- nor may stale register contents: the levels of the first rows run on ring slots no load has filled yet, and a
    kernel with NaNs in flight (k_land_fix over NaN land) precedes every first launch of a repeated filter."""
    from gcm_filters_amd import _lib
    shape = (400, 1200)
    gv = T.scalar_grid_vars(grid, shape)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    flt = Filter(filter_scale=8.0 * dx, dx_min=dx, n_steps=24, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    f = T.random_field(shape, 5)
    land = gv["wet_mask"] == 0 if "wet_mask" in gv else np.zeros(shape, bool)
    plan.set_tuning(multi_s=8, clenshaw=clenshaw)   # 1 (default): backward evaluation for the flux kinds; 2: for every kind
    request_restore = lambda: plan.set_tuning(multi_s=8, clenshaw=2)
    plan.ring_fallbacks()
    clean = flt.apply(f)
    backward = clenshaw == 2 or (clenshaw == 1 and grid == "IRREGULAR_WITH_LAND")
    assert ("k_ringc" if backward else "k_ring<") in plan.last_kernel()
    assert plan.ring_fallbacks() == 0
    if land.any():
        for _ in range(3):  # repeated: the second call's first launch follows the first call's k_land_fix
            got = flt.apply(np.where(land, np.nan, f))
        assert plan.ring_fallbacks() == 0
        assert np.array_equal(got[~land], clean[~land])
    g = f.copy()
    g[200, 600] = np.nan   # a wet cell
    assert not land[200, 600]
    got = flt.apply(g)
    with np.errstate(all="ignore"):
        fs = flt.filter_spec
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, g, gv)
    assert np.array_equal(np.isnan(got), np.isnan(want))   # the redone strips are the RIGHT strips (workgroup order != strip order)
    assert np.nanmax(np.abs(got - want)) <= 1e-11 * np.nanmax(np.abs(want))
    n = plan.ring_fallbacks()
    if grid == "REGULAR":   # no nan_to_num in the reference's REGULAR kernel: NaN spreads by plain arithmetic, nothing to redo
        assert n == 0
    else:
        assert 0 < n < 40, n   # the strips around the cell, in each of the three launches -- not the whole grid
    assert plan.ring_fallbacks() == 0   # reading resets
    request_restore()
