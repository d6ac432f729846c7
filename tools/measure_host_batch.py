"""PCIe-inclusive rate of a batched host array (numpy in / out), pipelined vs one shot."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
shape, nb = (2400, 3600), 12
gv = T.scalar_grid_vars("IRREGULAR_WITH_LAND", shape)
dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
f = np.stack([T.random_field(shape, 100 + b) for b in range(nb)])
for mb in (sys.argv[1:] or ("0", "32", "140")):
    os.environ["GCMF_HOST_CHUNK_MB"] = mb
    clear_plan_cache()
    flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    flt.apply(f)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); r = flt.apply(f); ts.append(time.perf_counter() - t0)
    cells = nb * shape[0] * shape[1] * flt.n_steps
    print(f"GCMF_HOST_CHUNK_MB={mb:>4s}: {min(ts)*1e3:7.2f} ms for {nb} fields = {min(ts)*1e3/nb:.2f} ms/field -> {cells/min(ts)/1e9:.1f} G cell-steps/s (PCIe-inclusive)", flush=True)
# the same with a page-locked INPUT (numpy view of a torch pinned tensor): is the remaining cost the pageable upload?
import torch
os.environ["GCMF_HOST_CHUNK_MB"] = "32"
clear_plan_cache()
fp = torch.empty(f.shape, dtype=torch.float64).pin_memory()
fp.numpy()[...] = f
flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
flt.apply(fp.numpy())
ts = []
for _ in range(4):
    t0 = time.perf_counter(); r = flt.apply(fp.numpy()); ts.append(time.perf_counter() - t0)
print(f"page-locked input, chunk 32 MB: {min(ts)*1e3/nb:.2f} ms/field -> {cells/min(ts)/1e9:.1f} G cell-steps/s")
