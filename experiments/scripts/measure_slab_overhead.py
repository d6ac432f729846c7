"""Driver overhead of SlabFilter: one rank owning the whole grid (world=1: no exchange) against Filter.apply on the same
workload.  What is left is the Python loop, the input copy and the event bookkeeping of the slab driver."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.distributed import SlabFilter
shape = (2400, 3600)
grid = "IRREGULAR_WITH_LAND"
gv = T.scalar_grid_vars(grid, shape)
dx = T.grid_dx_min(grid, gv)
fk = dict(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER)
f = T.random_field(shape, 100)
flt = Filter(grid_type=GridType[grid], grid_vars=gv, **fk)
d = torch.from_numpy(f).cuda()
flt.apply(d); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): flt.apply(d)
torch.cuda.synchronize(); t_f = (time.perf_counter() - t0) / 10
for tk in (False, True):
    sf = SlabFilter(grid, gv, fk, shape[0], shape[1], rank=0, world=1, device=0)
    sf.time_kernels = tk
    local = sf.scatter_from_global([f[None]])
    sf.apply_local(local); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): sf.apply_local(local)
    torch.cuda.synchronize(); t_s = (time.perf_counter() - t0) / 10
    sf.collect_kernel_times()
    print(f"Filter.apply {t_f*1e3:.3f} ms; SlabFilter.apply_local (world 1, time_kernels={tk}) {t_s*1e3:.3f} ms")
