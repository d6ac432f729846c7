import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch, warnings
warnings.simplefilter("ignore")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
shape = (2400, 3600)
grid = "IRREGULAR_WITH_LAND"
for dt in ("f4", "f8"):
    gv = {k: v.astype(dt) for k, v in T.scalar_grid_vars(grid, shape).items()}
    dx = T.grid_dx_min(grid, gv)
    d = torch.from_numpy(T.random_field(shape, 100).astype(dt)).cuda()
    for n in (40, 42, 43, 44, 45, 46, 47, 48):
        flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, n_steps=n, grid_type=GridType[grid], grid_vars=gv)
        flt.apply(d); torch.cuda.synchronize()
        td = []
        for _ in range(5):
            t0 = time.perf_counter(); flt.apply(d); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
        print(f"{dt} n_steps={n}: {min(td)*1e3:.3f} ms  ({(min(td)*1e3)/n*1e3:.1f} us/step)", flush=True)
