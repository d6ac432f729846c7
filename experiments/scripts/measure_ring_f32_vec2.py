import sys, time, warnings
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O
shape = (2400, 3600)
for grid in ("REGULAR_WITH_LAND", "REGULAR", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED"):
    gv = {k: v.astype("f4") for k, v in T.scalar_grid_vars(grid, shape).items()}
    dx = 1.0
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F32, shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    for nb in (1, 8):
        d = torch.from_numpy(np.stack([T.random_field(shape, 100 + b) for b in range(nb)]).astype("f4")).cuda()
        outs = {}
        for rnd in range(2):
            for opt in (0, 1):
                plan.set_option("ring_f32_vec2", opt)
                flt.apply(d); torch.cuda.synchronize()
                td = []
                for _ in range(4):
                    t0 = time.perf_counter(); r = flt.apply(d); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
                outs[opt] = r
                print(f"{grid:26s} f4 nb={nb} ring_f32_vec2={opt}: {min(td)*1e3:7.2f} ms {nb*shape[0]*shape[1]*flt.n_steps/min(td)/1e9:7.1f} G  {plan.last_kernel()[:60]}", flush=True)
        print("   same bits:", bool(torch.equal(outs[0], outs[1])))
        del d
    plan.set_option("ring_f32_vec2", 0)
