"""cProfile of the host side of Filter.apply on a device-resident field (small grid: the GPU work is negligible)."""
import cProfile, pstats, sys
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
shape = (512, 512)
grid = sys.argv[1] if len(sys.argv) > 1 else "IRREGULAR_WITH_LAND"
gv = T.scalar_grid_vars(grid, shape)
dx = T.grid_dx_min(grid, gv) if grid != "REGULAR" else 1.0
flt = Filter(filter_scale=8 * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType[grid], grid_vars=gv)
d = torch.from_numpy(T.random_field(shape, 1)).cuda()
for _ in range(3):
    flt.apply(d)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    flt.apply(d)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
