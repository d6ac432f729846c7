import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch, warnings
warnings.simplefilter("ignore")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
shape = (2400, 3600); grid = "TRIPOLAR_POP_WITH_LAND"; dt = sys.argv[1] if len(sys.argv) > 1 else "f4"
gv = {k: v.astype(dt) for k, v in T.scalar_grid_vars(grid, shape).items()}
dx = T.grid_dx_min(grid, gv)
flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
d = torch.from_numpy(T.random_field(shape, 100).astype(dt)).cuda()
for _ in range(4): flt.apply(d)
torch.cuda.synchronize()
