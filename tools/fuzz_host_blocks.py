"""Random scalar cases through the row-block host pipeline (forced on small grids) against the one-plan path (bit-identical)
and the oracle.   python tools/fuzz_host_blocks.py <seed> <ncases>"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from gcm_filters_amd import Filter, FilterShape, GridType, host_blocks, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
from oracle import gcmf_oracle as O
host_blocks.MIN_CELLS = 1
host_blocks.BUILD_AFTER_CALLS = 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(ncase):
    grid = T.SCALAR_GRIDS[rng.integers(len(T.SCALAR_GRIDS))]
    ny = int(rng.integers(260, 700)); nx = int(rng.integers(16, 160)) * 4
    if grid.startswith("TRIPOLAR"):
        nx += nx % 2
    n_steps = int(rng.integers(3, 40))
    K = int(rng.integers(2, 5))
    dt = "f8" if rng.random() < 0.7 else "f4"
    f, gv = T.scalar_case(grid, (ny, nx))
    mode = rng.integers(3)
    if mode == 1 and "wet_mask" in gv:
        f = np.where(gv["wet_mask"] == 0, np.nan, f)
    elif mode == 2:
        f = f.copy(); f[rng.integers(ny), rng.integers(nx)] = np.nan
    f = f.astype(dt); gv = {k: v.astype(dt) for k, v in gv.items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    shape = FilterShape.TAPER if rng.random() < 0.5 else FilterShape.GAUSSIAN
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=float(rng.uniform(2.5, 6)) * dx, dx_min=dx, n_steps=n_steps, filter_shape=shape,
                     grid_type=GridType[grid], grid_vars=gv)
    os.environ["GCMF_HOST_BLOCKS"] = "0"; clear_plan_cache()
    want = flt.apply(f)
    os.environ["GCMF_HOST_BLOCKS"] = str(K); clear_plan_cache()
    got = flt.apply(f)
    used = host_blocks.choose_blocks(ny, n_steps)
    ok = np.array_equal(got, want, equal_nan=True)
    if not ok:
        bad += 1
        print("MISMATCH", grid, (ny, nx), dt, n_steps, "K", K, "->", used, "mode", mode, float(np.nanmax(np.abs(got - want))), flush=True)
clear_plan_cache()
print(f"{ncase} cases, {bad} bad")
