import sys, warnings
import numpy as np
sys.path.insert(0, "/root/repo")
warnings.simplefilter("ignore")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O
shape = (48, 64)
bad = 0
for grid in ("REGULAR", "REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND", "VECTOR_C_GRID", "VECTOR_B_GRID"):
    vec = grid in T.VECTOR_GRIDS
    gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
    fields = [T.random_field(shape, 5 + c) for c in range(2 if vec else 1)]
    if not vec and "wet_mask" in gv:
        fields[0] = np.where(gv["wet_mask"] == 0, np.nan, fields[0])
    lap = ALL_KERNELS[GridType[grid]](**gv)
    for n in (1, 2, 3, 4, 5, 9, 17, 64, 257, 1000):
        # a stable polynomial of degree n: Chebyshev coefficients of a smooth function, reference-style spec
        rng = np.random.default_rng(n)
        p = rng.standard_normal(n + 1) / (1 + np.arange(n + 1)) ** 2
        dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
        spec = O.FilterSpec(n, 8.0 / dx**2 if O.DIMENSIONAL[grid] else 8.0, p, dx * dx)
        from gcm_filters_amd.filter import FilterSpec
        got = lap._run(fields, spec=FilterSpec(n, spec.s_max, p, spec.dx_min_sq))
        with np.errstate(all="ignore"):
            want = O.filter_func_vec(spec, grid, *fields, gv) if vec else (O.filter_func(spec, grid, fields[0], gv),)
        for g, w in zip(got, want):
            nz = lambda a: np.nan_to_num(a, nan=0.0)
            e = float(np.abs(nz(g) - nz(w)).max() / max(np.abs(nz(w)).max(), 1e-300))
            if not np.array_equal(np.isnan(g), np.isnan(w)) or e > 1e-10:
                print("BAD", grid, n, e); bad += 1
print("n_steps sweep done,", bad, "bad")
