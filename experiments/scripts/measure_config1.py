import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from gcm_filters_amd import Filter, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
f = T.random_field((512, 512), 100)
flt = Filter(filter_scale=4.0, dx_min=1.0, n_steps=16, grid_type=GridType.REGULAR)
d = torch.from_numpy(f).cuda()
for _ in range(5): out = flt.apply(d)
torch.cuda.synchronize()
plan = ALL_KERNELS[GridType.REGULAR]()._plan(_lib.F64, (512, 512))
for reps in (200, 200):
    t0 = time.perf_counter()
    for _ in range(reps): out = flt.apply(d)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue us", (t1 - t0) / reps * 1e6, "total us", (t2 - t0) / reps * 1e6, plan.last_kernel(), plan.last_kernel_geometry())
for strip in (20, 16, 12, 8, 4, 32, 0):
    plan.set_tuning(multi_s=8, strip_rows=strip)
    for _ in range(5): out = flt.apply(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): out = flt.apply(d)
    torch.cuda.synchronize()
    print("strip", strip, "total us", (time.perf_counter() - t0) / 300 * 1e6, plan.last_kernel_geometry())
