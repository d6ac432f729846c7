import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
wl = T.baseline_workload(3, (2400, 3600))
fk = wl["fk"]
flt = Filter(grid_type=GridType[wl["grid"]], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]])
f = wl["fields"][0]
fb = np.ascontiguousarray(np.broadcast_to(f, (8,) + f.shape))
d = torch.from_numpy(f).cuda()
def batch(tag, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); flt.apply(fb); ts.append(1e3 * (time.perf_counter() - t0) / 8)
    ts2 = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); flt.apply(d); torch.cuda.synchronize(); ts2.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag:50s}: host batch ms/field " + " ".join(f"{t:.2f}" for t in ts) + "   resident field ms " + " ".join(f"{t:.3f}" for t in ts2), flush=True)
batch("fresh process", 6)
x = torch.empty(20 << 30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize()
del x; torch.cuda.empty_cache()
t0 = time.perf_counter()
for k in range(12):
    batch(f"20 GB freed, +{time.perf_counter() - t0:5.2f} s")
    time.sleep(0.5)
x = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
batch("2 GB allocated again (alive)")
