"""Where does the on-chip kernel stop paying on whole grids?  us per Filter.apply, resident (GCMF_RESIDENT=1) against the strip-marching
launches (GCMF_RESIDENT=0), by kind and size -- the data behind the auto policy of gcmf_apply (resident_supported, gcmf_resident.hip)."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS

def timed(fn, reps=40):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for grid, kw in (("IRREGULAR_WITH_LAND", dict(filter_scale=16.0, filter_shape=FilterShape.TAPER)), ("REGULAR_WITH_LAND", dict(filter_scale=50.0)),
                 ("REGULAR", dict(filter_scale=50.0)), ("REGULAR", dict(filter_scale=20.0))):
    for shape in ((384, 384), (512, 512), (640, 640), (800, 800), (720, 1440), (1000, 1200)):
        f, gv = T.scalar_case(grid, shape)
        dx = T.grid_dx_min(grid, gv) if grid.startswith("IRREG") else 1.0
        k2 = dict(kw); k2["filter_scale"] *= dx
        flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, **k2)
        d = torch.from_numpy(f).cuda()
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
        res = {}
        for mode in ("1", "0"):
            os.environ["GCMF_RESIDENT"] = mode
            plan.last_kernel()
            t = timed(lambda: flt.apply(d))
            res[mode] = (t, plan.last_kernel())
        print(f"{grid:22s} {str(shape):13s} {shape[0]*shape[1]/1e3:7.0f} k cells n {flt.n_steps:3d}: on chip {res['1'][0]*1e6:7.1f} us [{res['1'][1]}]   strips {res['0'][0]*1e6:7.1f} us   ratio {res['1'][0]/res['0'][0]:.2f}", flush=True)
