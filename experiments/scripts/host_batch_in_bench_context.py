"""bench.py's run_host_path, step by step, to find what makes its batch-of-8 figure (2.8 ms per field) differ from a fresh process (1.7)."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
wl = T.baseline_workload(3, (2400, 3600))
fk = wl["fk"]
flt = Filter(grid_type=GridType[wl["grid"]], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]])
f = wl["fields"][0]
fb = np.ascontiguousarray(np.broadcast_to(f, (8,) + f.shape))
def batch(tag):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); flt.apply(fb); ts.append(1e3 * (time.perf_counter() - t0) / 8)
    print(f"{tag:50s}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
batch("fresh process")
d = torch.from_numpy(f).to("cuda:0"); flt.apply(d); torch.cuda.synchronize()
batch("after a device-resident call")
pin = torch.from_numpy(f).pin_memory(); d.copy_(pin, non_blocking=True); pin.copy_(d, non_blocking=True); torch.cuda.synchronize()
batch("after pin_memory copies")
for _ in range(6):
    out = flt.apply(f)
batch("after single-field host calls (row-block pipeline built)")
clear_plan_cache()
batch("after clear_plan_cache")
os.environ["GCMF_HOST_BLOCKS"] = "0"
for _ in range(3):
    flt.apply(f)
os.environ.pop("GCMF_HOST_BLOCKS")
batch("after one-plan single-field host calls")
