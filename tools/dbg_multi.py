import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
def run(grid, shape, dt, S, strip, n_steps, pf=0):
    f, gv = T.scalar_case(grid, shape)
    f = f.astype(dt); gv = {k: v.astype(dt) for k, v in gv.items()}
    lap = ALL_KERNELS[GridType[grid]](**gv)
    plan = lap._plan(_lib.dtype_code(dt), shape)
    flt = Filter(filter_scale=2.0, dx_min=1.0, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    plan.set_timing(True)
    plan.set_tuning(multi_s=1); ref = flt.apply(f); n1 = plan.last_timing()[1]
    plan.set_tuning(multi_s=S, strip_rows=strip, prefetch_rows=pf); got = flt.apply(f); n2 = plan.last_timing()[1]
    plan.set_tuning(multi_s=4, strip_rows=0)
    bad = ~((ref == got) | (np.isnan(ref) & np.isnan(got)))
    jj, ii = np.nonzero(bad)
    print(grid, shape, dt, "S", S, "strip", strip, "n", n_steps, "launches", n1, n2, "mismatches", bad.sum(), "rows", sorted(set(jj.tolist()))[:16], "maxdiff", np.nanmax(np.abs(ref-got)))
for S in (2, 3, 4, 8):
    for n in (S, 2*S, 11):
        run("TRIPOLAR_POP_WITH_LAND", (60, 160), "f8", S, 0, n)
run("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (41, 264), "f8", 4, 0, 8)
