"""Seeded synthetic grids and fields for tests, golden-vector generation and ``bench.py``.

These generators restate the *recipes* (seeds, value ranges, mask geometry) of the reference's
test fixtures so that arrays produced here are bit-identical to what the reference's own tests
feed its kernels -- which is what makes the reference's zarr goldens usable as parity pins:

* random fields          -> reference ``tests/conftest.py:79-81``   (PCG64(seed).random)
* land mask              -> reference ``tests/conftest.py:84-89``   (row 0 + SW quadrant are land)
* irregular metrics      -> reference ``tests/conftest.py:92-97``   (0.9 + 0.2*random)
* tripole-folded metrics -> reference ``tests/conftest.py:100-109`` (last row: right half mirrors left)
* scalar grid dict       -> reference ``tests/conftest.py:112-133``
* spherical C/B grid     -> reference ``tests/conftest.py:180-270``

Nothing in this module touches the GPU; it is plain numpy.
"""
from __future__ import annotations

import numpy as np
from numpy.random import PCG64, Generator

# argument order used by the reference *fixtures* (seed = position in this list), conftest.py:12-60
FIXTURE_ARG_ORDER = {
    "REGULAR": [],
    "REGULAR_AREA_WEIGHTED": ["area"],
    "REGULAR_WITH_LAND": ["wet_mask"],
    "REGULAR_WITH_LAND_AREA_WEIGHTED": ["wet_mask", "area"],
    "IRREGULAR_WITH_LAND": ["wet_mask", "dxw", "dyw", "dxs", "dys", "area", "kappa_w", "kappa_s"],
    "MOM5U": ["wet_mask", "dxt", "dyt", "dxu", "dyu", "area_u"],
    "MOM5T": ["wet_mask", "dxt", "dyt", "dxu", "dyu", "area_t"],
    "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED": ["wet_mask", "area"],
    "TRIPOLAR_POP_WITH_LAND": ["wet_mask", "dxe", "dye", "dxn", "dyn", "tarea"],
    "VECTOR_C_GRID": [
        "wet_mask_t", "wet_mask_q", "dxT", "dyT", "dxCu", "dyCu", "dxCv", "dyCv",
        "dxBu", "dyBu", "area_u", "area_v", "kappa_iso", "kappa_aniso",
    ],
    "VECTOR_B_GRID": ["DXU", "DYU", "HUS", "HUW", "HTE", "HTN", "UAREA", "TAREA"],
}

SCALAR_GRIDS = [
    "REGULAR", "REGULAR_AREA_WEIGHTED", "REGULAR_WITH_LAND", "REGULAR_WITH_LAND_AREA_WEIGHTED",
    "IRREGULAR_WITH_LAND", "MOM5U", "MOM5T", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED",
    "TRIPOLAR_POP_WITH_LAND",
]
VECTOR_GRIDS = ["VECTOR_C_GRID", "VECTOR_B_GRID"]
ALL_GRIDS = SCALAR_GRIDS + VECTOR_GRIDS
# grid types the reference's own fixtures cover (MOM5U/MOM5T are untested upstream, conftest.py:62-70)
REFERENCE_TESTED_GRIDS = [g for g in ALL_GRIDS if not g.startswith("MOM5")]

EARTH_RADIUS = 6378000


def random_field(shape, seed):
    """Uniform [0,1) field from PCG64(seed)."""
    return Generator(PCG64(seed)).random(shape)


def land_mask(shape):
    """1 = ocean, 0 = land: southernmost row and the south-west quadrant are land."""
    ny, nx = shape[-2:]
    m = np.ones(shape)
    m[..., 0, :] = 0
    m[..., : ny // 2, : nx // 2] = 0
    return m


def island_mask(shape, seed, n_islands=40, max_extent=0.06):
    """``land_mask`` AND a seeded set of rectangular islands (bench-only variant)."""
    ny, nx = shape[-2:]
    m = land_mask(shape)
    rng = Generator(PCG64(seed))
    for _ in range(n_islands):
        j0 = int(rng.integers(1, ny))
        i0 = int(rng.integers(0, nx))
        hj = max(1, int(rng.random() * max_extent * ny))
        hi = max(1, int(rng.random() * max_extent * nx))
        m[..., j0 : j0 + hj, i0 : i0 + hi] = 0
    return m


def irregular_metric(shape, seed):
    """Positive metric with mean 1 and +-10 % noise."""
    return 0.9 + 0.2 * Generator(PCG64(seed)).random(shape)


def tripole_metric(shape, seed):
    """``irregular_metric`` whose northernmost row folds onto itself (right half = reversed left)."""
    g = irregular_metric(shape, seed)
    nx = shape[-1]
    g[-1, nx // 2 :] = g[-1, : nx // 2][::-1]
    return g


def scalar_grid_vars(grid_type: str, shape=(128, 256)):
    """Grid-variable dict for a scalar grid type, seeds exactly as the reference fixture assigns them."""
    names = FIXTURE_ARG_ORDER[grid_type]
    gv = {}
    seed = 0
    for seed, name in enumerate(names):
        if name == "wet_mask":
            gv[name] = land_mask(shape)
        elif "kappa" in name:
            gv[name] = np.ones(shape)
        else:
            gv[name] = irregular_metric(shape, seed)
    if grid_type == "TRIPOLAR_POP_WITH_LAND":
        for name in names:  # seeds continue after the enumerate (6 for dxn, 7 for dyn)
            if name in ("dxn", "dyn"):
                seed += 1
                gv[name] = tripole_metric(shape, seed)
    return gv


def scalar_case(grid_type: str, shape=(128, 256), field_seed=100):
    """(field, grid_vars) of the reference's scalar kernel/filter fixtures."""
    return random_field(shape, field_seed), scalar_grid_vars(grid_type, shape)


def tripolar_unit_case(grid_type: str, shape=(128, 256)):
    """All-ones metrics + land mask, field seed 30 (reference ``conftest.py:146-162``)."""
    gv = {}
    for name in FIXTURE_ARG_ORDER[grid_type]:
        gv[name] = land_mask(shape) if name == "wet_mask" else np.ones(shape)
    return random_field(shape, 30), gv


def spherical_geometry(shape=(128, 256)):
    """Lon/lat of C-grid u and v points: 70S..70N, 0..60E."""
    ny, nx = shape
    lat0, lat1 = -70, 70
    lon0, lon1 = 0, 60
    lat_u = np.linspace(lat0 + 0.5 * (lat1 - lat0) / ny, lat1 - 0.5 * (lat1 - lat0) / ny, ny)
    lat_v = np.linspace(lat0 + (lat1 - lat0) / ny, lat1, ny)
    lon_u = np.linspace(lon0 + (lon1 - lon0) / nx, lon1, nx)
    lon_v = np.linspace(lon0 + 0.5 * (lon1 - lon0) / nx, lon1 - 0.5 * (lon1 - lon0) / nx, nx)
    geolon_u, geolat_u = np.meshgrid(lon_u, lat_u)
    geolon_v, geolat_v = np.meshgrid(lon_v, lat_v)
    return geolon_u, geolat_u, geolon_v, geolat_v


def vector_grid_vars(grid_type: str, shape=(128, 256)):
    """Spherical-geometry metrics for VECTOR_C_GRID / VECTOR_B_GRID (island mask in the SW quadrant)."""
    _, geolat_u, _, geolat_v = spherical_geometry(shape)
    ny, nx = shape
    names = FIXTURE_ARG_ORDER[grid_type]
    gv = {}
    dx_u = EARTH_RADIUS * np.cos(geolat_u / 360 * 2 * np.pi)
    dx_v = EARTH_RADIUS * np.cos(geolat_v / 360 * 2 * np.pi)
    dy = np.max(dx_u) * np.ones((ny, nx))
    for name in names:
        if name in ("dxCu", "dxT", "HUS", "HTE"):
            gv[name] = dx_u.copy()
        if name in ("dxCv", "dxBu", "DXU", "HUW", "HTN"):
            gv[name] = dx_v.copy()
    for name in names:
        if name in ("dyCu", "dyCv", "dyBu", "dyT", "DYU"):
            gv[name] = dy
    for name in names:
        if name == "area_u":
            gv[name] = gv["dxCu"] * gv["dyCu"]
        elif name == "area_v":
            gv[name] = gv["dxCv"] * gv["dyCv"]
        elif name == "UAREA":
            gv[name] = gv["DXU"] * gv["DYU"]
        elif name == "TAREA":
            gv[name] = gv["HTE"] * gv["DYU"]
    for name in names:
        if name in ("kappa_iso", "kappa_aniso"):
            gv[name] = np.ones((ny, nx))
    mask = np.ones((ny, nx))
    mask[: ny // 2, : nx // 2] = 0
    for name in names:
        if name in ("wet_mask_t", "wet_mask_q"):
            gv[name] = mask
    return {k: gv[k] for k in names}


def vector_case(grid_type: str, shape=(128, 256)):
    """((u, v), grid_vars) of the reference's vector fixtures (u seed 42, v seed 43)."""
    return (random_field(shape, 42), random_field(shape, 43)), vector_grid_vars(grid_type, shape)


def solid_body_rotation(shape=(128, 256)):
    """u = cos(lat), v = 0 on the spherical grid: every vector Laplacian must annihilate it."""
    _, geolat_u, _, _ = spherical_geometry(shape)
    u = np.cos(geolat_u / 360 * 2 * np.pi)
    return u, np.zeros_like(u)


def smooth_kappa(shape, seed):
    """Smooth field in (0, 1] whose maximum is exactly 1 (variable-scale filtering variant)."""
    ny, nx = shape
    rng = Generator(PCG64(seed))
    ph = rng.random(4) * 2 * np.pi
    y = np.linspace(0, 2 * np.pi, ny, endpoint=False)[:, None]
    x = np.linspace(0, 2 * np.pi, nx, endpoint=False)[None, :]
    k = 0.55 + 0.2 * np.sin(y + ph[0]) * np.cos(2 * x + ph[1]) + 0.2 * np.cos(3 * y + ph[2]) * np.sin(x + ph[3])
    k = np.clip(k, 0.05, None)
    return k / k.max()


def grid_dx_min(grid_type: str, grid_vars) -> float:
    """Smallest grid spacing over all spacing planes of a dimensional grid."""
    spacing = [v for k, v in grid_vars.items() if k.lower().startswith(("dx", "dy", "hu", "ht"))]
    if not spacing:
        return 1.0
    return float(min(np.min(s) for s in spacing))


# ------------------------------------------------------------------------------------------------
# BASELINE.json workloads: one definition shared by bench.py, the full-size GPU tests and
# tests/golden/make_golden.py (which feeds the very same arrays to the imported reference)
# ------------------------------------------------------------------------------------------------
BASELINE_SHAPE = (2400, 3600)
BASELINE_GRID = {1: "REGULAR", 2: "REGULAR_WITH_LAND", 3: "IRREGULAR_WITH_LAND", 4: "TRIPOLAR_POP_WITH_LAND",
                 5: "VECTOR_C_GRID", 6: "VECTOR_B_GRID"}


def baseline_workload(cfg: int, shape=BASELINE_SHAPE, nlev: int = 0, f32: bool = False, f64: bool = False,
                      scale: float = 0.0, levels=None):
    """Synthetic inputs of BASELINE.json config `cfg` (6 = the B-grid extra) on `shape`.

    Returns dict(grid, fields, grid_vars, fk) with fk = dict(filter_scale, dx_min, filter_shape name).
    `scale` overrides the filter scale in units of dx_min (config 2: 10 => n_steps 11, 50 => 56).
    `levels` selects which vertical levels of configs 5 / 6 to build (default range(nlev)); a level's
    fields depend only on its own index, so a subset equals the same levels of the full workload."""
    shape = tuple(shape)
    if cfg in (1, 2):
        grid = BASELINE_GRID[cfg]
        gv = {} if cfg == 1 else {"wet_mask": land_mask(shape)}
        fields = [random_field(shape, 100)]
        fk = dict(filter_scale=float(scale or 50.0), dx_min=1.0, filter_shape="GAUSSIAN")
    elif cfg in (3, 4):
        grid = BASELINE_GRID[cfg]
        gv = scalar_grid_vars(grid, shape)
        fields = [random_field(shape, 100)]
        dx = grid_dx_min(grid, gv)
        if cfg == 3:
            fk = dict(filter_scale=(scale or 16) * dx, dx_min=dx, filter_shape="TAPER")
        else:
            fk = dict(filter_scale=(scale or 50) * dx, dx_min=dx, filter_shape="GAUSSIAN")
    elif cfg == 5:
        grid = "VECTOR_C_GRID"
        cdt = np.float64 if f64 else np.float32   # BASELINE config 5 is f32; f64 is an extra measurement
        gv = {k: v.astype(cdt) for k, v in vector_grid_vars(grid, shape).items()}
        gv["kappa_aniso"] = np.zeros(shape, dtype=cdt)
        lv = list(levels) if levels is not None else list(range(nlev or 50))
        fields = [np.stack([random_field(shape, 42 + c + 2 * l).astype(cdt) for l in lv]) for c in range(2)]
        dx = grid_dx_min(grid, gv)
        fk = dict(filter_scale=(scale or 40) * dx, dx_min=dx, filter_shape="GAUSSIAN")
    elif cfg == 6:  # not a BASELINE config: the POP B-grid vector Laplacian at the benchmark size
        grid = "VECTOR_B_GRID"
        gv = vector_grid_vars(grid, shape)
        lv = list(levels) if levels is not None else list(range(nlev or 1))
        if len(lv) <= 1 and levels is None:
            fields = [random_field(shape, 42), random_field(shape, 43)]
        else:
            fields = [np.stack([random_field(shape, 42 + c + 2 * l) for l in lv]) for c in range(2)]
        if f32:
            gv = {k: v.astype(np.float32) for k, v in gv.items()}
            fields = [f.astype(np.float32) for f in fields]
        dx = grid_dx_min(grid, gv)
        fk = dict(filter_scale=(scale or 40) * dx, dx_min=dx, filter_shape="GAUSSIAN")
    else:
        raise ValueError(f"unknown BASELINE config {cfg}")
    return dict(grid=grid, fields=fields, grid_vars=gv, fk=fk)


def probe_points(shape, n=300, seed=2024):
    """Seeded (j, i) sample positions of the full-size golden probes (tests/golden/reference_fullsize.npz)."""
    rng = Generator(PCG64(seed))
    return rng.integers(0, shape[-2], n), rng.integers(0, shape[-1], n)


def free_port(tries: int = 64) -> int:
    """A TCP port for a local rendezvous (torch.distributed on 127.0.0.1) that is free NOW and that the kernel will not hand out as the
    source port of somebody's outgoing connection a moment later: ports asked from the OS (bind to 0) come from the ephemeral range, where
    exactly that happens -- `EADDRINUSE` when the rendezvous server finally listens (seen once in ~50 multi-process test runs on the GPU
    boxes).  So: a random port BELOW the ephemeral range, checked by binding it."""
    import random
    import socket
    lo_eph = 32768
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo_eph = int(f.read().split()[0])
    except (OSError, ValueError):
        pass
    hi = max(min(lo_eph, 32768), 12000)
    rnd = random.Random()
    for _ in range(tries):
        port = rnd.randrange(10000, hi)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as sk:       # give up on the preference
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
