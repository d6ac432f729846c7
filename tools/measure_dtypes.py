"""HBM-resident rates of the scalar kinds across dtypes / batch sizes (not a bench.py config; quick survey)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from oracle import gcmf_oracle as O
shape = (2400, 3600)
for grid in ("REGULAR", "REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "REGULAR_WITH_LAND_AREA_WEIGHTED", "MOM5U", "TRIPOLAR_POP_WITH_LAND"):
    for dt in ("f8", "f4"):
        for nb in (1, 8):
            gv = {k: v.astype(dt) for k, v in T.scalar_grid_vars(grid, shape).items()}
            dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
            flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
            f = np.stack([T.random_field(shape, 100 + b) for b in range(nb)]).astype(dt)
            d = torch.from_numpy(f).cuda()
            flt.apply(d); torch.cuda.synchronize()
            td = []
            for _ in range(4):
                t0 = time.perf_counter(); flt.apply(d); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
            cells = nb * shape[0] * shape[1] * flt.n_steps
            print(f"{grid:42s} {dt} nb={nb}: {min(td)*1e3:8.2f} ms  {cells/min(td)/1e9:7.1f} G cell-steps/s (n_steps {flt.n_steps})", flush=True)
            del d
