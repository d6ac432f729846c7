"""Halo depth K of the on-chip kernel on small whole grids: us per Filter.apply with GCMF_RESIDENT_K forced (run once per K: the
variable is read once per process).   GCMF_RESIDENT_K=8 python experiments/scripts/measure_resident_k.py"""
import os, sys, time
os.environ["GCMF_RESIDENT"] = "1"
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS

def timed(fn, reps=50):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for grid, shape, kw in (("REGULAR", (512, 512), dict(filter_scale=4.0, n_steps=16)), ("REGULAR", (512, 512), dict(filter_scale=32.0)),
                        ("REGULAR_WITH_LAND", (512, 512), dict(filter_scale=50.0)),
                        ("IRREGULAR_WITH_LAND", (256, 256), dict(filter_scale=16.0, filter_shape=FilterShape.TAPER)),
                        ("IRREGULAR_WITH_LAND", (512, 512), dict(filter_scale=16.0, filter_shape=FilterShape.TAPER)),
                        ("IRREGULAR_WITH_LAND", (512, 512), dict(filter_scale=8.0)),
                        ("IRREGULAR_WITH_LAND", (600, 640), dict(filter_scale=16.0, filter_shape=FilterShape.TAPER)),
                        ("REGULAR_WITH_LAND", (720, 1440), dict(filter_scale=50.0))):
    f, gv = T.scalar_case(grid, shape)
    dx = T.grid_dx_min(grid, gv) if grid.startswith("IRREG") else 1.0
    kw = dict(kw); kw["filter_scale"] *= dx
    flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, **kw)
    d = torch.from_numpy(f).cuda()
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    plan.last_kernel()
    t = timed(lambda: flt.apply(d))
    print(f"K={os.environ.get('GCMF_RESIDENT_K', 'auto'):4s} {grid} {shape} n_steps {flt.n_steps}: {t*1e6:7.1f} us  {plan.last_kernel()} {plan.last_kernel_geometry()}", flush=True)
