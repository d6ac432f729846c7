"""Grid-variable helpers: the small conversions the reference's tutorials wrap around ``Filter``.

* ``docs/examples/example_tripole_grid.ipynb``: POP history-file metrics (KMT, HUS, HTE, HTN, HUW, TAREA; cgs units, east /
  north face convention) -> ``TRIPOLAR_POP_WITH_LAND`` and, rolled to the west / south face convention,
  ``IRREGULAR_WITH_LAND``; ``dx_min`` as the smallest spacing over wet cells.
* ``docs/examples/example_vector_laplacian.ipynb``: MOM6 static-file metrics -> ``VECTOR_C_GRID`` (symmetric-memory arrays
  lose their first row / column), the anisotropic ``kappa_iso`` / ``kappa_aniso`` pair for fixed-factor filtering.

Everything is plain array code on the last two axes (y, x): numpy arrays in -> numpy arrays out; xarray objects in -> objects
of the same class out (same dims; coordinates are not carried, as in the notebooks' ``roll(..., roll_coords=False)``).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .kernels import GridType, required_grid_vars


def _data(x):
    return np.asarray(getattr(x, "data", x))


def _like(template, data):
    """Wrap `data` like `template` when that is an xarray object (same dims), else return the array."""
    dims = getattr(template, "dims", None)
    if dims is not None and np.shape(data) == np.shape(_data(template)):
        try:
            return template.copy(data=data)
        except (TypeError, AttributeError):  # minimal stand-ins without DataArray.copy(data=...)
            return type(template)(data, dims=dims)
    return data


def _check(grid_type: GridType, gv: Dict) -> Dict:
    want = required_grid_vars(grid_type)
    assert list(gv) == want, (list(gv), want)
    return gv


def wet_mask_from_kmt(kmt):
    """POP: a T-cell is ocean where it has at least one active level (``KMT > 0``)."""
    return _like(kmt, (_data(kmt) > 0).astype(np.float64))


def dx_min_over_ocean(wet_mask, *spacings) -> float:
    """Smallest of the given grid spacings over wet cells: the ``dx_min`` argument of a dimensional ``Filter``."""
    wet = _data(wet_mask) > 0
    return float(min(np.min(_data(s)[np.broadcast_to(wet, _data(s).shape)]) for s in spacings))


def pop_tripolar_grid_vars(kmt, hus, hte, htn, huw, tarea, cgs: bool = True) -> Dict:
    """``TRIPOLAR_POP_WITH_LAND`` grid variables from POP metrics (HUS / HTE: x / y spacing at the eastern T-cell edge,
    HTN / HUW: at the northern edge); ``cgs``: the file is in cm / cm^2 (POP history files) -> m / m^2."""
    s, a = (0.01, 1e-4) if cgs else (1.0, 1.0)
    return _check(GridType.TRIPOLAR_POP_WITH_LAND, {
        "wet_mask": wet_mask_from_kmt(kmt), "dxe": _like(hus, _data(hus) * s), "dye": _like(hte, _data(hte) * s),
        "dxn": _like(htn, _data(htn) * s), "dyn": _like(huw, _data(huw) * s), "tarea": _like(tarea, _data(tarea) * a)})


def pop_irregular_grid_vars(kmt, hus, hte, htn, huw, tarea, cgs: bool = True, kappa_w=None, kappa_s=None) -> Dict:
    """``IRREGULAR_WITH_LAND`` grid variables from the same POP metrics: that Laplacian wants the spacings at the WESTERN and
    SOUTHERN cell edges, i.e. the eastern / northern ones of the neighbour (a periodic roll by one along x / y); kappas
    default to one (no spatially varying filter scale)."""
    s, a = (0.01, 1e-4) if cgs else (1.0, 1.0)
    rx = lambda v: _like(v, np.roll(_data(v), 1, axis=-1) * s)
    ry = lambda v: _like(v, np.roll(_data(v), 1, axis=-2) * s)
    ones = np.ones(_data(hus).shape[-2:])
    return _check(GridType.IRREGULAR_WITH_LAND, {
        "wet_mask": wet_mask_from_kmt(kmt), "dxw": rx(hus), "dyw": rx(hte), "dxs": ry(htn), "dys": ry(huw),
        "area": _like(tarea, _data(tarea) * a),
        "kappa_w": _like(hus, ones) if kappa_w is None else kappa_w, "kappa_s": _like(hus, ones) if kappa_s is None else kappa_s})


def fixed_factor_grid_vars(area, wet_mask, tripolar: bool = False) -> Dict:
    """``REGULAR_WITH_LAND_AREA_WEIGHTED`` / ``TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED`` (simple fixed-factor filtering:
    ``dx_min = 1``, ``filter_scale`` = the coarsening factor)."""
    gt = GridType.TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED if tripolar else GridType.REGULAR_WITH_LAND_AREA_WEIGHTED
    return _check(gt, {"area": area, "wet_mask": wet_mask})


def trim_symmetric(x, trim_y: bool, trim_x: bool):
    """MOM6 symmetric-memory output carries one extra row / column at vorticity and velocity points: drop the first."""
    d = _data(x)
    d = d[..., 1:, :] if trim_y else d
    d = d[..., :, 1:] if trim_x else d
    return np.ascontiguousarray(d)


def mom6_cgrid_grid_vars(wet, wet_c, dxT, dyT, dxCu, dyCu, dxCv, dyCv, dxBu, dyBu, kappa_iso=None, kappa_aniso=None,
                         symmetric: bool = False) -> Dict:
    """``VECTOR_C_GRID`` grid variables from a MOM6 static file.  Areas at u / v points are the products of their spacings;
    ``symmetric``: u-point arrays have nx + 1 columns, v-point arrays ny + 1 rows, vorticity-point arrays both."""
    if symmetric:
        dxCu, dyCu = trim_symmetric(dxCu, False, True), trim_symmetric(dyCu, False, True)
        dxCv, dyCv = trim_symmetric(dxCv, True, False), trim_symmetric(dyCv, True, False)
        wet_c, dxBu, dyBu = (trim_symmetric(v, True, True) for v in (wet_c, dxBu, dyBu))
    shape = _data(dxT).shape[-2:]
    for name, v in (("dxCu", dxCu), ("dyCv", dyCv), ("dxBu", dxBu), ("wet_c", wet_c)):
        if _data(v).shape[-2:] != shape:
            raise ValueError(f"{name} has shape {_data(v).shape[-2:]}, the tracer grid {shape}: symmetric-memory output? "
                             f"(pass symmetric=True)")
    gv = {"wet_mask_t": wet, "wet_mask_q": wet_c, "dxT": dxT, "dyT": dyT, "dxCu": dxCu, "dyCu": dyCu, "dxCv": dxCv,
          "dyCv": dyCv, "dxBu": dxBu, "dyBu": dyBu, "area_u": _like(dxCu, _data(dxCu) * _data(dyCu)),
          "area_v": _like(dxCv, _data(dxCv) * _data(dyCv)),
          "kappa_iso": _like(dxT, np.ones(shape)) if kappa_iso is None else kappa_iso,
          "kappa_aniso": _like(dxT, np.zeros(shape)) if kappa_aniso is None else kappa_aniso}
    return _check(GridType.VECTOR_C_GRID, gv)


def fixed_factor_kappas(dxCu, dyCv, dx_max: Optional[float] = None):
    """Viscosity factors that make the C-grid vector filter a fixed-factor filter (scale = factor * local grid spacing):
    ``kappa_iso = dy^2 / dx_max^2``, ``kappa_aniso = (dx^2 - dy^2) / dx_max^2``; use with ``filter_scale = factor * dx_max``,
    ``dx_min`` as usual.  Returns (kappa_iso, kappa_aniso, dx_max)."""
    dx, dy = _data(dxCu), _data(dyCv)
    dx_max = float(max(dx.max(), dy.max())) if dx_max is None else float(dx_max)
    return _like(dxCu, dy * dy / dx_max ** 2), _like(dxCu, (dx * dx - dy * dy) / dx_max ** 2), dx_max
