import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import _lib, testing as T, GridType
from gcm_filters_amd.kernels import ALL_KERNELS
def timed(fn, reps=30):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
nx = 3600
for grid, rows in (("IRREGULAR_WITH_LAND", 364), ("REGULAR", 364), ("IRREGULAR_WITH_LAND", 96)):
    w = T.baseline_workload(3, (rows, nx))
    gv, f = (w["grid_vars"], w["fields"][0]) if grid != "REGULAR" else ({}, w["fields"][0])
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, (rows, nx))
    d = torch.from_numpy(f).cuda()
    u, v, out = torch.zeros_like(d), torch.zeros_like(d), torch.zeros_like(d)
    s = torch.cuda.current_stream().cuda_stream
    for L in (1, 2, 4, 5, 8, 9, 16, 32, 64):
        pk = np.full(L, 0.01)
        t = timed(lambda: plan.resident_levels(None, None, u.data_ptr(), v.data_ptr(), d.data_ptr(), None, pk, 0.5, 0.1, _lib.STEP_FIRST, 0, rows, stream=s), reps=20)
        print(f"{grid} {rows} x {nx}, L = {L}: {t*1e6:.1f} us  {plan.last_kernel()} {plan.last_kernel_geometry()}", flush=True)
