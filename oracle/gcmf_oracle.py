"""CPU oracle for the gcm-filters iterated-Laplacian path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module -- and there only as the checker / the CPU number printed beside the GPU number.  Nothing
under ``gcm_filters_amd/`` imports it; the product path raises if the HIP library is missing.

What it is: a plain-numpy restatement (written from the algorithm, not copied) of

* the filter-polynomial fit            reference ``gcm_filters/filter.py:28-151``
* the Chebyshev-recurrence filter loop reference ``gcm_filters/filter.py:154-291``
* the eleven Laplacian stencils        reference ``gcm_filters/kernels.py:107-840``

in the same *array-at-a-time* style and the same floating-point operation order as the reference, so
that it reproduces the reference bit-for-bit on this numpy.  It is the "reference numpy path" that
``bench.py`` times on the host cores (``cpu_baseline.kind = "port"``).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this module against
  (1) all 18 of the reference's own zarr goldens (decoded to ``tests/golden/reference_zarr.npz``),
  (2) the reference's known-answer polynomial coefficients (``tests/test_filter.py:23-84`` upstream),
  (3) fp64 vectors captured by importing the reference in the build container
      (``tests/golden/make_golden.py`` -> ``tests/golden/reference_generated.npz``), including the
      MOM5U/MOM5T kernels that upstream never tests.

Conventions: arrays are C-order ``(..., ny, nx)``; E/W shift along the last axis, N/S along axis -2;
every shift is periodic (``np.roll``), exactly like the reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, NamedTuple, Sequence

import numpy as np

# ----------------------------------------------------------------------------------------------------
# periodic neighbour operators:  E(a)[j,i] = a[j,i+1], W(a)[j,i] = a[j,i-1], N(a)[j,i] = a[j+1,i] ...
# ----------------------------------------------------------------------------------------------------


def E(a):
    return np.roll(a, -1, axis=-1)


def W(a):
    return np.roll(a, 1, axis=-1)


def N(a):
    return np.roll(a, -1, axis=-2)


def S(a):
    return np.roll(a, 1, axis=-2)


def _fold_extend(a):
    """Append the tripole ghost row: a mirrored copy of the northernmost row (kernels.py:33-40)."""
    return np.concatenate((a, a[..., -1:, ::-1]), axis=-2)


# ----------------------------------------------------------------------------------------------------
# Laplacians.  Each factory validates + precomputes like the reference __post_init__ and returns an
# object with  .apply(f)  /  .apply(u, v),  .prepare, .finalize, .is_dimensional
# ----------------------------------------------------------------------------------------------------

GRID_ARGS: Dict[str, Sequence[str]] = {
    # order == dataclass field order in the reference == order of positional grid args of filter_func
    "REGULAR": (),
    "REGULAR_AREA_WEIGHTED": ("area",),
    "REGULAR_WITH_LAND": ("wet_mask",),
    "REGULAR_WITH_LAND_AREA_WEIGHTED": ("area", "wet_mask"),
    "IRREGULAR_WITH_LAND": ("wet_mask", "dxw", "dyw", "dxs", "dys", "area", "kappa_w", "kappa_s"),
    "MOM5U": ("wet_mask", "dxt", "dyt", "dxu", "dyu", "area_u"),
    "MOM5T": ("wet_mask", "dxt", "dyt", "dxu", "dyu", "area_t"),
    "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED": ("area", "wet_mask"),
    "TRIPOLAR_POP_WITH_LAND": ("wet_mask", "dxe", "dye", "dxn", "dyn", "tarea"),
    "VECTOR_C_GRID": (
        "wet_mask_t", "wet_mask_q", "dxT", "dyT", "dxCu", "dyCu", "dxCv", "dyCv",
        "dxBu", "dyBu", "area_u", "area_v", "kappa_iso", "kappa_aniso",
    ),
    "VECTOR_B_GRID": ("DXU", "DYU", "HUS", "HUW", "HTE", "HTN", "UAREA", "TAREA"),
}
GRID_TYPE_VALUES = {name: k + 1 for k, name in enumerate(GRID_ARGS)}  # enum values 1..11
DIMENSIONAL = {
    "REGULAR": False, "REGULAR_AREA_WEIGHTED": False, "REGULAR_WITH_LAND": False,
    "REGULAR_WITH_LAND_AREA_WEIGHTED": False, "IRREGULAR_WITH_LAND": True, "MOM5U": True, "MOM5T": True,
    "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED": False, "TRIPOLAR_POP_WITH_LAND": True,
    "VECTOR_C_GRID": True, "VECTOR_B_GRID": True,
}
AREA_WEIGHTED = {"REGULAR_AREA_WEIGHTED", "REGULAR_WITH_LAND_AREA_WEIGHTED",
                 "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"}
VECTOR = {"VECTOR_C_GRID", "VECTOR_B_GRID"}


class _Lap:
    is_vector = False

    def __init__(self, grid_type, apply, area=None):
        self.grid_type = grid_type
        self.is_dimensional = DIMENSIONAL[grid_type]
        self.apply = apply
        self._area = area

    def __call__(self, *f):
        return self.apply(*f)

    # AreaWeightedMixin (kernels.py:89-104): transform to / from the unit-spaced grid
    def prepare(self, *f):
        if self._area is None:
            return f[0] if len(f) == 1 else f
        return f[0] * self._area

    def finalize(self, *f):
        if self._area is None:
            return f[0] if len(f) == 1 else f
        return f[0] / self._area


def _regular(**gv):                                                       # kernels.py:107-124
    def L(f):
        return -4 * f + E(f) + W(f) + N(f) + S(f)
    return L


def _regular_land(wet_mask, **_):                                         # kernels.py:150-190
    m = wet_mask
    nwet = E(m) + W(m) + N(m) + S(m)

    def L(f):
        g = m * np.nan_to_num(f)
        g = -nwet * g + E(g) + W(g) + N(g) + S(g)
        return m * g
    return L


def _irregular_land(wet_mask, dxw, dyw, dxs, dys, area, kappa_w, kappa_s):  # kernels.py:222-318
    if np.any(kappa_w > 1.0):
        raise ValueError("There are kappa_w values > 1 and this can cause the filter to blow up."
                         "Please make sure all kappa_w are <=1.")
    if np.any(kappa_s > 1.0):
        raise ValueError("There are kappa_s values > 1 and this can cause the filter to blow up."
                         "Please make sure all kappa_s are <=1.")
    near1 = lambda k: np.any(np.isclose(k, 1.0, rtol=0, atol=1e-05))
    if not (near1(kappa_w) or near1(kappa_s)):
        raise ValueError("At least one place in the domain must have either kappa_w = 1 or kappa_s = 1. "
                         "Otherwise the filter's scale will not be equal to filter_scale anywhere in the domain.")
    face_w = wet_mask * W(wet_mask) * kappa_w
    face_s = wet_mask * S(wet_mask) * kappa_s

    def L(f):
        g = np.nan_to_num(f)
        fw = (g - W(g)) / dxw * dyw
        fs = (g - S(g)) / dys * dxs
        fw = fw * face_w
        fs = fs * face_s
        return (E(fw) - fw + N(fs) - fs) / area
    return L


def _mom5u(wet_mask, dxt, dyt, dxu, dyu, area_u):                          # kernels.py:321-375
    # NB: the reference names the axis -2 difference "fx" and masks it with the axis -1 face mask.
    mask_a = wet_mask * E(wet_mask)
    mask_b = wet_mask * N(wet_mask)

    def L(f):
        g = np.nan_to_num(f)
        fx = 2 * (N(g) - g)
        fx /= N(dxt) + N(E(dxt))
        fy = 2 * (E(g) - g)
        fy /= E(dyt) + N(E(dyt))
        fx *= mask_a
        fy *= mask_b
        o1 = 0.5 * fx * (dyu + N(dyu))
        o1 -= 0.5 * S(fx) * (dyu + S(dyu))
        o1 /= area_u
        o2 = 0.5 * fy * (dxu + E(dxu))
        o2 -= 0.5 * W(fy) * (dxu + W(dxu))
        o2 /= area_u
        return o1 + o2
    return L


def _mom5t(wet_mask, dxt, dyt, dxu, dyu, area_t):                          # kernels.py:378-432
    mask_a = wet_mask * E(wet_mask)
    mask_b = wet_mask * N(wet_mask)

    def L(f):
        g = np.nan_to_num(f)
        fx = 2 * (N(g) - g)
        fx /= dxu + W(dxu)
        fy = 2 * (E(g) - g)
        fy /= dyu + S(dyu)
        fx *= mask_a
        fy *= mask_b
        o1 = fx * 0.5 * (dyt + N(dyt))
        o1 -= S(fx) * 0.5 * (dyt + S(dyt))
        o1 /= area_t
        o2 = fy * 0.5 * (dxt + E(dxt))
        o2 -= W(fy) * 0.5 * (dxt + W(dxt))
        o2 /= area_t
        return o1 + o2
    return L


def _require_antarctica(wet_mask):
    if wet_mask[..., 0, :].any():
        raise AssertionError("Wet mask requires zeros in southernmost row")


def _tripolar_regular(area, wet_mask):                                     # kernels.py:435-492
    _require_antarctica(wet_mask)
    mx = _fold_extend(wet_mask)
    nwet = E(mx) + W(mx) + N(mx) + S(mx)

    def L(f):
        g = wet_mask * np.nan_to_num(f)
        g = _fold_extend(g)
        g = -nwet * g + E(g) + W(g) + N(g) + S(g)
        return wet_mask * g[..., :-1, :]
    return L


def _tripolar_pop(wet_mask, dxe, dye, dxn, dyn, tarea):                    # kernels.py:495-588
    _require_antarctica(wet_mask)
    mx = _fold_extend(wet_mask)
    dxe_x, dye_x, dxn_x, dyn_x = (_fold_extend(a) for a in (dxe, dye, dxn, dyn))
    face_e = mx * E(mx)
    face_n = mx * N(mx)
    nx = dxn_x.shape[-1]
    for name, arr, exact in (("dxn", dxn_x, True), ("dyn", dyn_x, False)):
        wet_only = np.where(face_n == 1, arr, 0)
        left = wet_only[..., -2, : nx // 2][..., ::-1]
        right = wet_only[..., -2, nx // 2:]
        ok = np.all(left == right) if exact else np.allclose(left, right)
        if not ok:
            raise AssertionError(f"Northernmost row of {name} does not fold onto itself. "
                                 "This is a requirement for using a tripole boundary condition.")

    def L(f):
        g = _fold_extend(np.nan_to_num(f))
        fe = (E(g) - g) / dxe_x * dye_x
        fn = (N(g) - g) / dyn_x * dxn_x
        fe = fe * face_e
        fn = fn * face_n
        out = fe - W(fe) + fn - S(fn)
        return out[..., :-1, :] / tarea
    return L


def _cgrid(wet_mask_t, wet_mask_q, dxT, dyT, dxCu, dyCu, dxCv, dyCv, dxBu, dyBu,
           area_u, area_v, kappa_iso, kappa_aniso):                         # kernels.py:591-699
    dx_dyT = dxT / dyT * wet_mask_t
    dy_dxT = dyT / dxT * wet_mask_t
    dx_dyBu = dxBu / dyBu * wet_mask_q
    dy_dxBu = dyBu / dxBu * wet_mask_q
    dx2h, dy2h = dxT * dxT, dyT * dyT
    dx2q, dy2q = dxBu * dxBu, dyBu * dyBu
    with np.errstate(divide="ignore"):
        rau = np.where(area_u > 0, 1 / area_u, 0)
        rav = np.where(area_v > 0, 1 / area_v, 0)

    def L(u, v):
        u = np.nan_to_num(u)
        v = np.nan_to_num(v)
        du_dx = dy_dxT * (u / dyCu - W(u / dyCu))
        dv_dy = dx_dyT * (v / dxCv - S(v / dxCv))
        sxx = du_dx - dv_dy
        sxx = -(kappa_iso + 0.5 * kappa_aniso) * sxx
        dv_dx = dy_dxBu * (E(v / dyCv) - v / dyCv)
        du_dy = dx_dyBu * (N(u / dxCu) - u / dxCu)
        sxy = dv_dx + du_dy
        sxy = -kappa_iso * sxy
        lu = 1 / dyCu * (dy2h * sxx - E(dy2h * sxx))
        lu += 1 / dxCu * (S(dx2q * sxy) - dx2q * sxy)
        lu *= rau
        lv = 1 / dyCv * (W(dy2q * sxy) - dy2q * sxy)
        lv -= 1 / dxCv * (dx2h * sxx - N(dx2h * sxx))
        lv *= rav
        return lu, lv
    return L


def bgrid_coefficients(DXU, DYU, HUS, HUW, HTE, HTN, UAREA, TAREA):
    """The ten stencil weights of the POP B-grid operator (kernels.py:746-809); field independent."""
    ra_u, ra_t = 1 / UAREA, 1 / TAREA
    rdx, rdy = 1 / DXU, 1 / DYU
    w = HUS / HTE
    dus = w * ra_u
    dun = W(w) * ra_u
    w = HUW / HTN
    duw = w * ra_u
    due = S(w) * ra_u
    kxu = (S(HUW) - HUW) * ra_u
    kyu = (W(HUS) - HUS) * ra_u
    kxt = (HTE - N(HTE)) * ra_t
    w2 = 0.5 * (kxt + W(kxt))
    dxkx = (S(w2) - w2) * rdx
    w2 = 0.5 * (kxt + S(kxt))
    dykx = (W(w2) - w2) * rdy
    kyt = (HTN - E(HTN)) * ra_t
    w2 = 0.5 * (kyt + S(kyt))
    dyky = (W(w2) - w2) * rdy
    w2 = 0.5 * (kyt + W(kyt))
    dxky = (S(w2) - w2) * rdx
    dum = -(dxkx + dyky + 2 * (kxu * kxu + kyu * kyu))
    dmc = dxky - dykx
    dme = (2 * kyu) / (HTN + S(HTN))
    dmn = -(2 * kxu) / (HTE + W(HTE))
    duc = -(dun + dus + due + duw)
    return dict(cc=duc + dum, dun=dun, dus=dus, due=due, duw=duw, dmc=dmc, dmn=dmn, dms=-dmn, dme=dme, dmw=-dme)


def _bgrid(**gv):                                                          # kernels.py:702-840
    def L(u, v):
        u = np.nan_to_num(u)
        v = np.nan_to_num(v)
        c = bgrid_coefficients(**gv)   # the reference rebuilds the weights on every call; so do we
        am = 1
        lu = am * (c["cc"] * u + c["dun"] * N(u) + c["dus"] * S(u) + c["due"] * E(u) + c["duw"] * W(u)
                   + c["dmc"] * v + c["dmn"] * N(v) + c["dms"] * S(v) + c["dme"] * E(v) + c["dmw"] * W(v))
        lv = am * (c["cc"] * v + c["dun"] * N(v) + c["dus"] * S(v) + c["due"] * E(v) + c["duw"] * W(v)
                   + c["dmc"] * u + c["dmn"] * N(u) + c["dms"] * S(u) + c["dme"] * E(u) + c["dmw"] * W(u))
        return lu, lv
    return L


_FACTORY: Dict[str, Callable] = {
    "REGULAR": _regular,
    "REGULAR_AREA_WEIGHTED": lambda area: _regular(),
    "REGULAR_WITH_LAND": _regular_land,
    "REGULAR_WITH_LAND_AREA_WEIGHTED": lambda area, wet_mask: _regular_land(wet_mask),
    "IRREGULAR_WITH_LAND": _irregular_land,
    "MOM5U": _mom5u,
    "MOM5T": _mom5t,
    "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED": _tripolar_regular,
    "TRIPOLAR_POP_WITH_LAND": _tripolar_pop,
    "VECTOR_C_GRID": _cgrid,
    "VECTOR_B_GRID": _bgrid,
}


def make_laplacian(grid_type: str, grid_vars: dict) -> _Lap:
    """Build the Laplacian of ``grid_type`` (name of the reference ``GridType`` member)."""
    need = set(GRID_ARGS[grid_type])
    if set(grid_vars) != need:
        raise ValueError(f"grid_vars {sorted(grid_vars)} do not match {sorted(need)}")
    lap = _Lap(grid_type, _FACTORY[grid_type](**grid_vars),
               area=grid_vars["area"] if grid_type in AREA_WEIGHTED else None)
    lap.is_vector = grid_type in VECTOR
    return lap


# ----------------------------------------------------------------------------------------------------
# filter polynomial (host side)
# ----------------------------------------------------------------------------------------------------

N_STEPS_PARAMS = {  # filter.py:28-37
    "GAUSSIAN": {1: (0.8, 0.0, 1), 2: (1.1, 0.0, 1)},
    "TAPER": {1: (2.2, 0.6, 2.5), 2: (3.2, 0.7, 2.7)},
}


class FilterSpec(NamedTuple):
    n_steps: int
    s_max: float
    p: np.ndarray
    dx_min_sq: float


def n_steps_default(ndim, filter_shape: str, filter_scale, dx_min, transition_width=np.pi):
    offset, factor, exponent = N_STEPS_PARAMS[filter_shape][ndim]                       # filter.py:74-89
    per_cell = offset + factor * ((np.pi / transition_width) ** exponent)
    return max(np.ceil(per_cell * (filter_scale / dx_min)).astype(int), 3)


def target_function(filter_shape: str, s_max, filter_scale, transition_width):
    """Target transfer function F(t), t in [-1, 1] <-> s = k^2 in [0, s_max] (filter.py:47-65)."""
    if filter_shape == "GAUSSIAN":
        return lambda t: np.exp(-(s_max * (t + 1) / 2) * filter_scale ** 2 / 24)
    from scipy.interpolate import PchipInterpolator
    knots = np.array([0, 2 * np.pi / (transition_width * filter_scale), 2 * np.pi / filter_scale,
                      8 * np.sqrt(s_max)])
    fk = PchipInterpolator(knots, np.array([1, 1, 0, 0]))
    return lambda t: fk(np.sqrt((t + 1) * (s_max / 2)))


def filter_spec(filter_scale, dx_min, filter_shape: str, transition_width=np.pi, ndim=2, n_steps=0) -> FilterSpec:
    """Galerkin fit of the target in the Shen basis phi_i = T_i - T_{i+2} (filter.py:99-151)."""
    cheb = np.polynomial.chebyshev
    n = n_steps
    mass = (np.pi / 2) * (2 * np.eye(n - 1) - np.diag(np.ones(n - 3), 2) - np.diag(np.ones(n - 3), -2))
    mass[0, 0] = 3 * np.pi / 2
    s_max = ndim * (2 / dx_min) ** 2
    F = target_function(filter_shape, s_max, filter_scale, transition_width)
    rhs = np.zeros(n - 1)
    x, w = cheb.chebgauss(n + 1)
    for i in range(n - 1):
        e = np.zeros(n + 1)
        e[i], e[i + 2] = 1, -1
        phi = cheb.chebval(x, e)
        rhs[i] = np.sum(w * phi * (F(x) - ((1 - x) / 2 + F(1) * (x + 1) / 2)))
    c_hat = np.linalg.solve(mass, rhs)
    p = np.zeros(n + 1)
    p[0] = c_hat[0] + (1 + F(1)) / 2
    p[1] = c_hat[1] - (1 - F(1)) / 2
    for i in range(2, n - 1):
        p[i] = c_hat[i] - c_hat[i - 2]
    p[n - 1] = -c_hat[n - 3]
    p[n] = -c_hat[n - 2]
    return FilterSpec(n, s_max, p, dx_min ** 2)


# ----------------------------------------------------------------------------------------------------
# the hot loop: Chebyshev three-term recurrence on the shifted Laplacian  A(x) = -x - c L(x)
# ----------------------------------------------------------------------------------------------------


def _shift_scale(spec: FilterSpec, lap: _Lap):
    return 2 / spec.s_max if lap.is_dimensional else 2 / (spec.s_max * spec.dx_min_sq)


def filter_func(spec: FilterSpec, grid_type: str, field, grid_vars: dict):
    """Scalar filter (filter.py:177-212): returns the filtered copy of ``field``."""
    lap = make_laplacian(grid_type, grid_vars)
    c = _shift_scale(spec, lap)
    A = lambda x: -x - c * lap(x)
    fbar = lap.prepare(field.copy())
    t2 = fbar.copy()
    t1 = A(fbar)
    fbar = spec.p[0] * t2 + spec.p[1] * t1
    for k in range(2, spec.n_steps + 1):
        t0 = 2 * A(t1) - t2
        fbar += spec.p[k] * t0
        t2 = t1.copy()
        t1 = t0.copy()
    return lap.finalize(fbar)


def filter_func_vec(spec: FilterSpec, grid_type: str, u, v, grid_vars: dict):
    """Vector filter (filter.py:242-289): the same recurrence on the coupled (u, v) pair."""
    lap = make_laplacian(grid_type, grid_vars)
    c = _shift_scale(spec, lap)

    def A(a, b):
        la, lb = lap(a, b)
        return -a - c * la, -b - c * lb

    ubar, vbar = u.copy(), v.copy()
    u2, v2 = ubar.copy(), vbar.copy()
    u1, v1 = A(ubar, vbar)
    ubar = spec.p[0] * u2 + spec.p[1] * u1
    vbar = spec.p[0] * v2 + spec.p[1] * v1
    for k in range(2, spec.n_steps + 1):
        a0, b0 = A(u1, v1)
        a0 = 2 * a0 - u2
        b0 = 2 * b0 - v2
        ubar += spec.p[k] * a0
        vbar += spec.p[k] * b0
        u2, u1 = u1.copy(), a0.copy()
        v2, v1 = v1.copy(), b0.copy()
    return ubar, vbar


def make_spec(filter_scale, dx_min, filter_shape="GAUSSIAN", transition_width=np.pi, ndim=2, n_steps=0):
    """``Filter.__post_init__`` n_steps logic (filter.py:352-384) without the grid checks."""
    if n_steps < 3:
        n_steps = int(n_steps_default(ndim, filter_shape, filter_scale, dx_min, transition_width))
    return filter_spec(filter_scale, dx_min, filter_shape, transition_width, ndim, n_steps)
