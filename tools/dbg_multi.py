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
    plan.set_tuning(multi_s=1); ref = flt.apply(f)
    plan.set_tuning(multi_s=S, strip_rows=strip, prefetch_rows=pf); got = flt.apply(f)
    plan.set_tuning(multi_s=4, strip_rows=0)
    bad = ~((ref == got) | (np.isnan(ref) & np.isnan(got)))
    jj, ii = np.nonzero(bad)
    print(grid, shape, dt, "S", S, "strip", strip, "n", n_steps, "pf", pf, "mismatches", bad.sum(), "rows", sorted(set(jj.tolist()))[:12], "cols%8", sorted(set((ii % 8).tolist())), "cols", ii[:12].tolist())
    if bad.any():
        j, i = jj[0], ii[0]
        print("   first", j, i, ref[j, i], got[j, i], "mask", gv["wet_mask"][j-1:j+2, i-1:i+2].tolist())
for strip in (0, 16, 24, 48):
    run("REGULAR_WITH_LAND", (48, 256), "f4", 8, strip, 8)
run("REGULAR_WITH_LAND", (48, 256), "f4", 8, 0, 8, pf=2)
run("REGULAR_WITH_LAND", (64, 256), "f4", 8, 0, 8)
run("REGULAR_WITH_LAND", (48, 256), "f4", 8, 0, 16)
run("REGULAR_WITH_LAND", (48, 256), "f4", 8, 0, 9)
