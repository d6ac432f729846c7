"""float32 SCALAR fields: error of each evaluation order against the same filter in f64 arithmetic (small grid, oracle = the reference's
own f32 path: f32 T_k, f64 running sum) and the speed of both on a 2400 x 3600 grid.  The vector twin: tools/measure_cgrid_f32_error.py."""
import sys, time, warnings
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

GRIDS = sys.argv[1:] or ["REGULAR", "REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND"]
rel = lambda a, b: float(np.nanmax(np.abs(a - b)) / np.nanmax(np.abs(b)))
for grid in GRIDS:
    shape = (96, 160)
    f64, gv64 = T.scalar_case(grid, shape)
    f = f64.astype("f4")
    gv = {k: v.astype("f4") for k, v in gv64.items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    for n, scale in ((16, 8), (44, 40), (98, 90)):
        flts = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for ev in ("auto", "reference"):
                flts[ev] = Filter(filter_scale=scale * dx, dx_min=dx, n_steps=n, grid_type=GridType[grid], grid_vars=gv, evaluation=ev)
        fs = flts["auto"].filter_spec
        spec = O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq)
        with np.errstate(all="ignore"):
            ref = O.filter_func(spec, grid, f, gv)
            truth = O.filter_func(spec, grid, f.astype("f8"), {k: v.astype("f8") for k, v in gv.items()})
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F32, shape)
        out = {}
        for ev in ("auto", "reference"):
            plan.last_kernel()
            out[ev] = (flts[ev].apply(f), plan.last_kernel().split("(")[0][:40])
        print(f"{grid:28s} n={n:3d}: reference's f32 path vs f64 {rel(ref, truth):.2e} | auto {rel(out['auto'][0], truth):.2e} ({out['auto'][1]}) | "
              f"evaluation='reference' {rel(out['reference'][0], truth):.2e}, bit-equal to the reference: {np.array_equal(out['reference'][0], ref, equal_nan=True)} ({out['reference'][1]})",
              flush=True)
for grid in GRIDS:
    shape = (2400, 3600)
    gv = {k: v.astype("f4") for k, v in T.scalar_grid_vars(grid, shape).items()}
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    for nb in (1, 8):
        d = torch.from_numpy(np.stack([T.random_field(shape, 100 + b) for b in range(nb)]).astype("f4")).cuda()
        for ev in ("auto", "reference"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv, evaluation=ev)
            flt.apply(d); torch.cuda.synchronize()
            td = []
            for _ in range(4):
                t0 = time.perf_counter(); flt.apply(d); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
            print(f"{grid:28s} f4 nb={nb} {ev:9s}: {min(td)*1e3:7.2f} ms  {nb*shape[0]*shape[1]*flt.n_steps/min(td)/1e9:7.1f} G cell-steps/s (n_steps {flt.n_steps})", flush=True)
        del d
