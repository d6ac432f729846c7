"""Which earlier activity of a process makes the batched host path (upload || recurrence || download) lose its overlap?  (bench.py: 2.8 ms per
field after the other configs have run, 1.7 ms in a fresh process.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
wl = T.baseline_workload(3, (2400, 3600))
fk = wl["fk"]
mk = lambda: Filter(grid_type=GridType[wl["grid"]], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]])
flt = mk()
f = wl["fields"][0]
fb = np.ascontiguousarray(np.broadcast_to(f, (8,) + f.shape))
def batch(tag):
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); flt.apply(fb); ts.append(1e3 * (time.perf_counter() - t0) / 8)
    print(f"{tag:60s}: " + " ".join(f"{t:.2f}" for t in ts[-4:]), flush=True)
batch("fresh process")
x = torch.empty(20 << 30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
batch("after a 20-GB device allocation came and went")
clear_plan_cache(); batch("... and a new plan")
xs = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(40)]; torch.cuda.synchronize(); del xs; torch.cuda.empty_cache()
clear_plan_cache(); batch("after 40 x 1 GB came and went, new plan")
wl5 = T.baseline_workload(5, (2400, 3600), nlev=50)
f5 = Filter(grid_type=GridType.VECTOR_C_GRID, grid_vars=wl5["grid_vars"], filter_scale=wl5["fk"]["filter_scale"], dx_min=wl5["fk"]["dx_min"], filter_shape=FilterShape[wl5["fk"]["filter_shape"]])
u, v = (torch.from_numpy(a).cuda() for a in wl5["fields"])
r = f5.apply_to_vector(u, v); torch.cuda.synchronize(); del r, u, v; torch.cuda.empty_cache()
batch("after config 5 ran on the device (same host plan)")
clear_plan_cache(); batch("... and a new plan")
big = np.ones((50, 2400, 3600), dtype=np.float32); r = f5.apply_to_vector(big, big); del r, big
clear_plan_cache(); batch("after config 5 ran from HOST arrays (3.4 GB through the pipeline), new plan")
