import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
def run(grid, shape, dt, S, strip, n_steps, nan=False):
    f, gv = T.scalar_case(grid, shape)
    if nan: f = np.where(gv.get("wet_mask", np.ones(shape)) == 0, np.nan, f)
    f = f.astype(dt); gv = {k: v.astype(dt) for k, v in gv.items()}
    lap = ALL_KERNELS[GridType[grid]](**gv)
    plan = lap._plan(_lib.dtype_code(dt), shape)
    flt = Filter(filter_scale=2.0, dx_min=1.0, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    plan.set_tuning(multi_s=1); ref = flt.apply(f)
    plan.set_tuning(multi_s=S, strip_rows=strip); got = flt.apply(f)
    plan.set_tuning(multi_s=8, strip_rows=0)
    bad = ~((ref == got) | (np.isnan(ref) & np.isnan(got)))
    jj, ii = np.nonzero(bad)
    print(grid, shape, dt, "S", S, "n", n_steps, "mismatches", bad.sum(), "of", bad.size, "maxrel", np.nanmax(np.abs(ref-got))/np.nanmax(np.abs(ref)), "cols%4", sorted(set((ii%4).tolist()))[:8], "rows", sorted(set(jj.tolist()))[:8])
run("IRREGULAR_WITH_LAND", (96, 160), "f8", 2, 0, 2)
run("IRREGULAR_WITH_LAND", (96, 160), "f8", 4, 0, 4)
run("IRREGULAR_WITH_LAND", (96, 160), "f8", 2, 0, 2, nan=True)
