"""Grid-variable helpers (tutorial conversions) against the oracle: the west / south-face variables rolled out
of POP's east / north-face metrics make IRREGULAR_WITH_LAND agree with TRIPOLAR_POP_WITH_LAND away from the fold."""
import sys

import numpy as np
import pytest

from gcm_filters_amd import GridType, grid_helpers as H, required_grid_vars, testing as T
from oracle import gcmf_oracle as O


def _pop_file(shape=(40, 64)):
    gv = T.scalar_grid_vars("TRIPOLAR_POP_WITH_LAND", shape)
    kmt = (gv["wet_mask"] * 7).astype(np.int32)
    return kmt, gv["dxe"] * 100, gv["dye"] * 100, gv["dxn"] * 100, gv["dyn"] * 100, gv["tarea"] * 1e4, gv


def test_pop_conversions():
    kmt, hus, hte, htn, huw, tarea, gv = _pop_file()
    pop = H.pop_tripolar_grid_vars(kmt, hus, hte, htn, huw, tarea)
    assert list(pop) == required_grid_vars(GridType.TRIPOLAR_POP_WITH_LAND)
    for k in gv:
        np.testing.assert_allclose(pop[k], gv[k], rtol=1e-15)
    irr = H.pop_irregular_grid_vars(kmt, hus, hte, htn, huw, tarea)
    assert list(irr) == required_grid_vars(GridType.IRREGULAR_WITH_LAND)
    f = T.random_field(kmt.shape, 5) * pop["wet_mask"]
    a = O.make_laplacian("TRIPOLAR_POP_WITH_LAND", pop)(f)
    b = O.make_laplacian("IRREGULAR_WITH_LAND", irr)(f)
    np.testing.assert_allclose(a[1:-1], b[1:-1], rtol=1e-12, atol=1e-14)        # same operator away from the fold row
    assert not np.allclose(a[-1], b[-1])                                          # the tripole seam differs
    dx = H.dx_min_over_ocean(pop["wet_mask"], pop["dxe"], pop["dye"], pop["dxn"], pop["dyn"])
    wet = pop["wet_mask"] > 0
    assert dx == min(pop[k][wet].min() for k in ("dxe", "dye", "dxn", "dyn"))
    ff = H.fixed_factor_grid_vars(pop["tarea"], pop["wet_mask"], tripolar=True)
    assert list(ff) == required_grid_vars(GridType.TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED)


def test_mom6_cgrid_conversions():
    shape = (24, 32)
    gv = T.vector_grid_vars("VECTOR_C_GRID", shape)
    pad = lambda a, y, x: np.pad(a, ((1 if y else 0, 0), (1 if x else 0, 0)), constant_values=-1.0)
    sym = dict(wet=gv["wet_mask_t"], wet_c=pad(gv["wet_mask_q"], 1, 1), dxT=gv["dxT"], dyT=gv["dyT"],
               dxCu=pad(gv["dxCu"], 0, 1), dyCu=pad(gv["dyCu"], 0, 1), dxCv=pad(gv["dxCv"], 1, 0), dyCv=pad(gv["dyCv"], 1, 0),
               dxBu=pad(gv["dxBu"], 1, 1), dyBu=pad(gv["dyBu"], 1, 1))
    out = H.mom6_cgrid_grid_vars(**sym, symmetric=True)
    assert list(out) == required_grid_vars(GridType.VECTOR_C_GRID)
    for k in gv:
        if k != "kappa_aniso":
            np.testing.assert_array_equal(np.asarray(out[k]), gv[k], err_msg=k)
    assert not np.asarray(out["kappa_aniso"]).any()
    with pytest.raises(ValueError, match="symmetric"):
        H.mom6_cgrid_grid_vars(**sym)
    ki, ka, dx_max = H.fixed_factor_kappas(gv["dxCu"], gv["dyCv"])
    assert dx_max == max(gv["dxCu"].max(), gv["dyCv"].max()) and ki.max() <= 1.0 + 1e-15
    np.testing.assert_allclose(ki + ka, gv["dxCu"] ** 2 / dx_max ** 2)


def test_xarray_objects_come_back_as_xarray(monkeypatch):
    import fake_xarray as xr
    kmt, hus, hte, htn, huw, tarea, gv = _pop_file()
    da = lambda a: xr.DataArray(a, dims=["nlat", "nlon"])
    irr = H.pop_irregular_grid_vars(*(da(a) for a in (kmt, hus, hte, htn, huw, tarea)))
    assert all(isinstance(v, xr.DataArray) and v.dims == ("nlat", "nlon") for v in irr.values())
    np.testing.assert_allclose(irr["dxw"].data, np.roll(gv["dxe"], 1, axis=-1), rtol=1e-15)
