"""VECTOR_B_GRID 2400x3600: forward (default, bit-exact with numpy) against backward (GCMF_CLENSHAW=2) evaluation."""
import sys, time, warnings
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
shape = (2400, 3600)
for dt, nlev in (("f8", 20), ("f4", 40), ("f8", 1)):
    gv = {k: v.astype(dt) for k, v in T.vector_grid_vars("VECTOR_B_GRID", shape).items()}
    dx = T.grid_dx_min("VECTOR_B_GRID", gv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=20 * dx, dx_min=dx, n_steps=24, grid_type=GridType.VECTOR_B_GRID, grid_vars=gv)
    u = torch.from_numpy(np.stack([T.random_field(shape, 42 + l).astype(dt) for l in range(nlev)])).cuda()
    v = torch.from_numpy(np.stack([T.random_field(shape, 43 + l).astype(dt) for l in range(nlev)])).cuda()
    plan = ALL_KERNELS[GridType.VECTOR_B_GRID](**gv)._plan(_lib.dtype_code(dt), shape)
    for cl in (1, 2, 1, 2):
        plan.set_tuning(multi_s=8, clenshaw=cl)
        flt.apply_to_vector(u, v); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            flt.apply_to_vector(u, v)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / 3
        print(f"{dt} x{nlev}: clenshaw={cl} {nlev*shape[0]*shape[1]*24/dtm/1e9:7.1f} G cell-steps/s  {plan.last_kernel()}", flush=True)
    plan.set_tuning(multi_s=8, clenshaw=2)
    del u, v
