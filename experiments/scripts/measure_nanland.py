"""Realistic ocean input: NaN on land.  Rate of the blocked kernels with and without NaNs in the field."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from oracle import gcmf_oracle as O
shape = (2400, 3600)
for grid in ("REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND"):
    gv = T.scalar_grid_vars(grid, shape)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    f = T.random_field(shape, 100)
    land = gv["wet_mask"] == 0
    for name, field in (("finite everywhere", f), ("NaN on land", np.where(land, np.nan, f)), ("zero on land", np.where(land, 0.0, f))):
        d = torch.from_numpy(field).cuda()
        flt.apply(d); torch.cuda.synchronize()
        td = []
        for _ in range(5):
            t0 = time.perf_counter(); flt.apply(d); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
        cells = shape[0] * shape[1] * flt.n_steps
        print(f"{grid:24s} {name:18s}: {min(td)*1e3:6.2f} ms  {cells/min(td)/1e9:6.1f} G cell-steps/s  (land fraction {land.mean():.2f})", flush=True)
