"""A 1/4-degree tripolar grid through the default path, 30 applications (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["GCMF_RESIDENT"] = "0"
import numpy as np, torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
grid = sys.argv[1] if len(sys.argv) > 1 else "TRIPOLAR_POP_WITH_LAND"
shape = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1440)
f, gv = T.scalar_case(grid, shape)
dx = T.grid_dx_min(grid, gv)
flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, filter_scale=16.0 * dx, filter_shape=FilterShape.TAPER)
d = torch.from_numpy(f).cuda()
for _ in range(30):
    flt.apply(d)
torch.cuda.synchronize()
print(grid, shape, "n", flt.n_steps)
