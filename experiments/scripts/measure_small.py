"""Small-grid latency: where does a Filter.apply call on a small device-resident field spend its time?
Prints the host cost of the whole Python call and of the bare C call (gcmf_apply through ctypes)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
for shape in ((128, 128), (512, 512), (1024, 1024)):
    for grid in ("REGULAR", "REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND"):
        gv = T.scalar_grid_vars(grid, shape)
        dx = T.grid_dx_min(grid, gv) if grid == "IRREGULAR_WITH_LAND" else 1.0
        flt = Filter(filter_scale=8 * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType[grid], grid_vars=gv)
        d = torch.from_numpy(T.random_field(shape, 1)).cuda()
        o = torch.empty_like(d)
        for _ in range(3):
            flt.apply(d)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            flt.apply(d)
        t_enq = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
        lap = ALL_KERNELS[GridType[grid]](**gv)
        plan = lap._plan(_lib.F64, shape, 0)
        p = np.asarray(flt.filter_spec.p, dtype=np.float64)
        c = 2 / flt.filter_spec.s_max if lap.is_dimensional else 2 / (flt.filter_spec.s_max * flt.filter_spec.dx_min_sq)
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            plan.apply(p, c, [d.data_ptr()], [o.data_ptr()], 1, device_ptrs=True, stream=st)
        t_c = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_c_all = (time.perf_counter() - t0) / n
        print(f"{grid:22s} {shape}: n_steps={flt.n_steps:3d}  Filter.apply host {t_enq*1e6:7.1f} us (steady {t_all*1e6:7.1f});"
              f"  bare gcmf_apply host {t_c*1e6:6.1f} us (steady {t_c_all*1e6:6.1f})", flush=True)
