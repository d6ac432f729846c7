"""Plan-creation cost (grid upload + plan-time folding + validation), first and later plans in a process."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
from gcm_filters_amd import GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS, clear_plan_cache
t0 = time.perf_counter(); _lib.load(); print(f"library load: {1e3*(time.perf_counter()-t0):.1f} ms")
for shape in ((512, 512), (2400, 3600)):
    for grid in ("REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND", "VECTOR_C_GRID"):
        vec = grid in T.VECTOR_GRIDS
        gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
        for rep in range(2):
            clear_plan_cache()
            t0 = time.perf_counter()
            lap = ALL_KERNELS[GridType[grid]](**gv)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            print(f"{grid:24s} {shape} plan #{rep}: {1e3*(t1-t0):7.1f} ms ({sum(v.nbytes for v in gv.values())/1e6:.0f} MB of grid planes)", flush=True)
        f = [T.random_field(shape, 3)] * (2 if vec else 1)
        d = [torch.from_numpy(x).cuda() for x in f]
        t0 = time.perf_counter(); lap._run(d); torch.cuda.synchronize(); t1 = time.perf_counter()
        lap._run(d); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"    first laplacian call {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.2f} ms")
