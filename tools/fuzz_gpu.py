"""Randomised GPU-vs-oracle sweep (not part of the test suite): grid types x shapes x dtypes x batches x n_steps."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
from oracle import gcmf_oracle as O

# --eval auto|reference|backward: the Filter's evaluation order (default auto; "backward" = the all-f32 backward kernels for f32 scalar / B-grid fields)
EVAL = sys.argv[sys.argv.index("--eval") + 1] if "--eval" in sys.argv else "auto"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
worst = {}
t0 = time.time()
for it in range(ncase):
    grid = T.ALL_GRIDS[rng.integers(len(T.ALL_GRIDS))]
    if "--cgrid" in sys.argv:  # batched vector fields: exercise the temporally blocked vector kernels
        grid = "VECTOR_C_GRID"
    if "--bgrid" in sys.argv:
        grid = "VECTOR_B_GRID"
    vec = grid in T.VECTOR_GRIDS
    if "--tiny" in sys.argv:
        ny = int(rng.integers(1, 14)); nx = int(rng.integers(1, 14))
    elif "--mid" in sys.argv:   # strips of 5..60 rows on whole grids: early-exit strips, both march directions, the level ramp
        ny = int(rng.integers(300, 1300)); nx = int(rng.integers(400, 1600))
    else:
        ny = int(rng.integers(3, 200)); nx = int(rng.integers(2, 700))
    if grid.startswith("TRIPOLAR"):
        nx += nx % 2; ny = max(ny, 4)
    if rng.random() < 0.5: nx = (nx // 4 + 1) * 4      # vector-width friendly half of the time
    if grid.startswith("TRIPOLAR"):
        nx += nx % 2; ny = max(ny, 2)
    shape = (ny, nx)
    dt = "f8" if rng.random() < 0.7 else "f4"
    nb = () if rng.random() < 0.6 else (int(rng.integers(1, 4)),) if rng.random() < 0.7 else (2, int(rng.integers(1, 3)))
    if "--cgrid" in sys.argv or "--bgrid" in sys.argv:
        nb = (int(rng.choice([1, 3, 4, 8, 12, 41])),)
    gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
    ncomp = 2 if vec else 1
    fields = [rng.random(nb + shape) for _ in range(ncomp)]
    if not vec and "wet_mask" in gv and rng.random() < 0.5:
        fields = [np.where(gv["wet_mask"] == 0, np.nan, f) for f in fields]
    if grid == "IRREGULAR_WITH_LAND" and rng.random() < 0.5:
        gv["kappa_w"] = T.smooth_kappa(shape, int(rng.integers(100))) if min(shape) > 1 else gv["kappa_w"]
    if ("--nanwet" in sys.argv or "--infwet" in sys.argv) and rng.random() < 0.6:
        # a few NaN (--nanwet) / +-inf (--infwet) anywhere: the static-ring kernels' per-strip fallback.  nan_to_num turns inf
        # into +-DBL_MAX and the stencil then overflows: where it does depends on the order of operations, so on the grid
        # types whose coefficients are folded at plan time only the land-mask / regular kinds can be held to the NaN pattern
        vals = [np.nan] if "--infwet" not in sys.argv else [np.nan, np.inf, -np.inf]
        for f in fields:
            flat = f.reshape(-1)
            for _ in range(int(rng.integers(1, 4))):
                flat[int(rng.integers(flat.size))] = rng.choice(vals)
    fields = [f.astype(dt) for f in fields]
    gv = {k: v.astype(dt) for k, v in gv.items()}
    dim = O.DIMENSIONAL[grid]
    dx = T.grid_dx_min(grid, gv) if dim else 1.0
    shp = "GAUSSIAN" if rng.random() < 0.5 else "TAPER"
    n_steps = int(rng.integers(3, 40))
    scale = float(rng.uniform(1.5, 6.0)) * dx
    if "-v" in sys.argv:
        print("CASE", it, grid, shape, dt, nb, shp, n_steps, round(scale / dx, 3), flush=True)
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flt = Filter(filter_scale=scale, dx_min=dx, filter_shape=FilterShape[shp], n_steps=n_steps, grid_type=GridType[grid], grid_vars=gv,
                         evaluation=EVAL)
            if "--tune" in sys.argv:  # random blocking depth / strip height / prefetch depth of the plan this case uses
                from gcm_filters_amd import _lib
                from gcm_filters_amd.kernels import ALL_KERNELS
                plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.dtype_code(dt), shape)
                plan.set_tuning(multi_s=int(rng.integers(1, 9)), strip_rows=int(rng.choice([0, 0, 3, 5, 9, 17, 40, 70, 100])),
                                prefetch_rows=int(rng.choice([0, 0, 1, 2])), xcd_remap=int(rng.integers(0, 2)), zigzag=int(rng.integers(0, 2)))
            got = flt.apply_to_vector(*fields) if vec else (flt.apply(fields[0]),)
        spec = O.FilterSpec(flt.n_steps, flt.filter_spec.s_max, np.asarray(flt.filter_spec.p), flt.filter_spec.dx_min_sq)
        with np.errstate(all="ignore"):
            want = O.filter_func_vec(spec, grid, *fields, gv) if vec else (O.filter_func(spec, grid, fields[0], gv),)
    except Exception as e:
        print("EXC", grid, shape, dt, nb, n_steps, repr(e)[:200]); continue
    err = 0.0
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and g.shape == w.shape, (grid, g.dtype, w.dtype)
        if not np.array_equal(np.isnan(g), np.isnan(w)):
            print("NANPATTERN", grid, shape, dt, nb, n_steps); err = np.inf; break
        ok = np.isfinite(w)
        sc = np.abs(w[ok]).max() if ok.any() else 1.0
        if sc > 0 and ok.any():
            err = max(err, float(np.abs(g[ok] - w[ok]).max() / sc))
    tol = 2e-4 if dt == "f4" else 1e-9
    key = (grid, dt)
    worst[key] = max(worst.get(key, 0.0), err)
    if not err <= tol:
        print("FAIL", grid, shape, dt, nb, shp, n_steps, err)
    if it % 50 == 49: clear_plan_cache()
print(f"{ncase} cases in {time.time()-t0:.1f} s; worst relative errors:")
for k in sorted(worst): print("  ", k, f"{worst[k]:.2e}")
