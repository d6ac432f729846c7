import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["GCMF_RESIDENT"] = "0"
import numpy as np, torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
grid = "TRIPOLAR_POP_WITH_LAND"
for shape in ((720, 1440), (1080, 1440)):
    f, gv = T.scalar_case(grid, shape)
    dx = 1.0 if len(sys.argv) > 1 else T.grid_dx_min(grid, gv)
    flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, filter_scale=50.0 * dx) if len(sys.argv) > 1 else Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, filter_scale=16.0 * dx, filter_shape=FilterShape.TAPER)
    d = torch.from_numpy(f).cuda()
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    for z in (0, 1, 2, 3):
        plan.set_option("ringc_zip", z)
        plan.ring_fallbacks(); plan.last_kernel()
        flt.apply(d); torch.cuda.synchronize()
        k = plan.last_kernel(); g = plan.last_kernel_geometry(); nfb = plan.ring_fallbacks()
        plan.set_timing(2); flt.apply(d); ms, nl, lo, hi = plan.last_kernel_timing(); tot, launches = plan.last_timing(); plan.set_timing(False)
        t0 = time.perf_counter()
        for _ in range(20): flt.apply(d)
        torch.cuda.synchronize()
        o = flt.apply(d); print("finite", bool(torch.isfinite(o).all()), float(o.abs().max()), end=" ")
        print(shape, "zip", z, k, g, "fallbacks", nfb, f"dominant {ms/max(nl,1)*1e3:.1f} us x {nl} (min {lo*1e3:.1f} max {hi*1e3:.1f}), total {tot*1e3:.1f} us in {launches} launches; wall {(time.perf_counter()-t0)/20*1e6:.1f} us", flush=True)
