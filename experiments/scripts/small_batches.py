"""Small batches on a 1/4-degree grid: the packed kernel (+ the seam's band on tripolar plans) against the zipped strips of every field side by side."""
import os, sys, time, warnings
sys.path.insert(0, os.getcwd())
os.environ["GCMF_RESIDENT"] = "0"
import numpy as np, torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
warnings.simplefilter("ignore")
def timed(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best
for grid in ("IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND"):
    for shape in ((1080, 1440), (720, 1440)):
        f, gv = T.scalar_case(grid, shape)
        dx = T.grid_dx_min(grid, gv)
        flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, filter_scale=16.0 * dx, filter_shape=FilterShape.TAPER)
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
        for nb in (1, 2, 3, 4, 6, 8, 16):
            d = torch.from_numpy(np.stack([f + 0.1 * i for i in range(nb)])).cuda()
            row = []
            for pack in (1, 0):
                plan.set_option("pack_batch", pack)
                plan.last_kernel(); flt.apply(d); k = plan.last_kernel()
                t = timed(lambda: flt.apply(d))
                row.append(f"pack {pack}: {t*1e6:8.1f} us {nb*shape[0]*shape[1]*flt.n_steps/t/1e9:6.1f} G {k.split('::')[1][:28]}")
            plan.set_option("pack_batch", 1)
            print(grid[:9], shape, "nb", nb, " | ".join(row), flush=True)
