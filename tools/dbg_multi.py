import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
def run(grid, shape, dt, S, strip, n_steps, skew=1, nan=True):
    f, gv = T.scalar_case(grid, shape)
    if nan: f = np.where(gv.get("wet_mask", np.ones(shape)) == 0, np.nan, f)
    f = f.astype(dt); gv = {k: v.astype(dt) for k, v in gv.items()}
    lap = ALL_KERNELS[GridType[grid]](**gv)
    plan = lap._plan(_lib.dtype_code(dt), shape)
    flt = Filter(filter_scale=2.0, dx_min=1.0, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    plan.set_timing(True)
    plan.set_tuning(multi_s=1); ref = flt.apply(f); n1 = plan.last_timing()[1]
    plan.set_tuning(multi_s=S, strip_rows=strip, skew=skew); got = flt.apply(f); n2 = plan.last_timing()[1]
    plan.set_tuning(multi_s=8, strip_rows=0)
    bad = ~((ref == got) | (np.isnan(ref) & np.isnan(got)))
    jj, ii = np.nonzero(bad)
    print(grid, shape, dt, "S", S, "strip", strip, "n", n_steps, "launches", n1, n2, "mismatches", bad.sum(), "rows", sorted(set(jj.tolist()))[:16])
for S in (4, 6):
    for grid, shape, dt in [("IRREGULAR_WITH_LAND", (96, 160), "f8"), ("REGULAR", (50, 258), "f8"), ("REGULAR_WITH_LAND_AREA_WEIGHTED", (70, 300), "f8"),
                            ("IRREGULAR_WITH_LAND", (64, 256), "f4"), ("REGULAR_WITH_LAND", (48, 1032), "f4"), ("TRIPOLAR_POP_WITH_LAND", (60, 160), "f8"), ("MOM5U", (40, 64), "f8")]:
        for n, strip in ((S, 0), (2 * S + 1, 0), (16, 7)):
            run(grid, shape, dt, S, strip, n)
