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
def batch(tag):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); flt.apply(fb); ts.append(1e3 * (time.perf_counter() - t0) / 8)
    print(f"{tag:70s}: " + " ".join(f"{t:.2f}" for t in ts[-3:]), flush=True)
batch("fresh process")
y = torch.empty(1 << 20, dtype=torch.uint8, device="cuda"); y.fill_(1); torch.cuda.synchronize()
batch("after a torch kernel on 1 MB")
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
batch("1 GB allocated (alive, untouched)")
del x; torch.cuda.empty_cache()
batch("1 GB freed")
x = torch.empty(4 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
batch("4 GB allocated (alive, untouched)")
x.fill_(1); torch.cuda.synchronize()
batch("4 GB touched by a kernel")
del x; torch.cuda.empty_cache()
batch("4 GB freed")
x = torch.empty(20 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
batch("20 GB allocated (alive, untouched)")
del x; torch.cuda.empty_cache()
batch("20 GB freed")
x = torch.empty(20 << 30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize()
batch("20 GB allocated and touched")
del x; torch.cuda.empty_cache()
batch("20 GB freed")
