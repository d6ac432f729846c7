"""Mid-size grids (0.5 - 4 M cells: too big for the on-chip kernel, small enough that 1024 lone waves march short strips): G cell-steps/s of the
strip-marching launches by blocking depth -- is 8 levels per launch still the right cut when a strip is 14 rows + 16 ghost rows?
    python tools/measure_midsize.py [grid types...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["GCMF_RESIDENT"] = "0"
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS

def timed(fn, reps=30):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn(); torch.cuda.synchronize()
    for _ in range(3):      # (untimed bursts: the runtime's one-off ~30-50 ms stall after the first few hundred launches of a process
        for _ in range(reps):   #  otherwise lands in the first timed loop -- 1.3 ms per application instead of 0.16, round 6)
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

grids = sys.argv[1:] or ["IRREGULAR_WITH_LAND", "REGULAR_WITH_LAND"]
for grid in grids:
    for shape in ((720, 1440), (1080, 1440), (1440, 2880), (2400, 3600)):
        f, gv = T.scalar_case(grid, shape)
        if os.environ.get("MIDSIZE_DT", "f8") == "f4":   # f32 fields and grid variables
            f = f.astype("f4"); gv = {k: np.asarray(v).astype("f4") for k, v in gv.items()}
        dx = T.grid_dx_min(grid, gv) if grid.startswith("IRREG") else 1.0
        kw = dict(filter_scale=16.0 * dx, filter_shape=FilterShape.TAPER) if grid.startswith("IRREG") else dict(filter_scale=50.0 * dx)
        flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, **kw)
        d = torch.from_numpy(f).cuda()
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64 if f.dtype == np.float64 else _lib.F32, shape)
        out = []
        for depth in (8, 7, 6, 5):
            plan.set_tuning(multi_s=depth)
            t = timed(lambda: flt.apply(d))
            out.append(f"S{depth}: {t*1e6:7.1f} us {shape[0]*shape[1]*flt.n_steps/t/1e9:6.1f} G [{plan.last_kernel_geometry()}]")
        plan.set_tuning(multi_s=8)
        print(f"{grid:22s} {str(shape):13s} n {flt.n_steps:3d}: " + "   ".join(out), flush=True)
