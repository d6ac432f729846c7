"""k_ringcz (strips zipped in pairs) against the plain strips: same bits?  how fast?   python experiments/scripts/zip_ab.py
Rows: (option "ringc_zip", option "ringc_smax") -- ringc_zip 0 = plain strips, 1 = the default policy, 2 = zipped with early exits, 3 = zipped in
whole ring periods; ringc_smax 0 = the default cut, 9 / 8 / 7 = at most that many levels per launch.  One process, alternating."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["GCMF_RESIDENT"] = "0"
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS

def timed(fn, reps=40):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best

grid = "IRREGULAR_WITH_LAND"
shapes = [(300, 3600), (720, 1440), (1080, 1440), (1440, 2880), (1800, 3600), (2400, 3600)]
for shape in shapes:
    for nan in (False, True):
        f, gv = T.scalar_case(grid, shape)
        if nan:
            rng = np.random.default_rng(5)
            f = f.copy()
            idx = rng.integers(0, f.size, 7)
            f.reshape(-1)[idx] = np.nan
            f.reshape(-1)[rng.integers(0, f.size, 2)] = np.inf
        dx = T.grid_dx_min(grid, gv)
        for scale in (16.0,):
            flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, filter_scale=scale * dx, filter_shape=FilterShape.TAPER)
            d = torch.from_numpy(f).cuda()
            plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
            res = {}
            for zi, zip_ in enumerate(((0, 9), (2, 9), (3, 9), (2, 8), (3, 8), (2, 7), (0, 0), (1, 0))):
                plan.set_option("ringc_zip", zip_[0])
                plan.set_option("ringc_smax", zip_[1])
                plan.last_kernel()
                o = flt.apply(d)
                torch.cuda.synchronize()
                k = plan.last_kernel(); g = plan.last_kernel_geometry()
                t = timed(lambda: flt.apply(d)) if not nan else 0.0
                res[(zi,) + zip_] = (o.cpu().numpy(), k, g, t)
            a = res[(0, 0, 9)][0]
            same = all(np.array_equal(a, res[z][0], equal_nan=True) for z in res)
            print(f"{shape} n {flt.n_steps} nan {nan}: same bits {same}")
            if not nan:
                for z in res:
                    print(f"     {z}: {res[z][1]} [{res[z][2]}] {res[z][3]*1e6:.1f} us {shape[0]*shape[1]*flt.n_steps/max(res[z][3],1e-9)/1e9:.0f} G", flush=True)
            if not same:
                for z in res:
                    b = res[z][0]
                    bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
                    if len(bad):
                        print("   ", z, res[z][1], "differs in", len(bad), "cells; first", bad[:4].tolist(), "rows", np.unique(bad[:, 0])[:24].tolist(), "cols", np.unique(bad[:, 1])[:12].tolist(), flush=True)
