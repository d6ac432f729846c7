import sys, warnings
sys.path.insert(0, "/root/repo")
import numpy as np
from gcm_filters_amd import Filter, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O
# usage: measure_cgrid_f32_error.py [VECTOR_C_GRID | VECTOR_B_GRID]   (f32 fields: error of each evaluation order against f64 arithmetic)
GRID = sys.argv[1] if len(sys.argv) > 1 else "VECTOR_C_GRID"
shape = (96, 160)
nlev = 8
gv = {k: v.astype("f4") for k, v in T.vector_grid_vars(GRID, shape).items()}
u = np.stack([T.random_field(shape, 42 + 2 * l).astype("f4") for l in range(nlev)])
v = np.stack([T.random_field(shape, 43 + 2 * l).astype("f4") for l in range(nlev)])
dx = T.grid_dx_min(GRID, gv)
plan = ALL_KERNELS[GridType[GRID]](**gv)._plan(_lib.F32, shape)
for n, scale in ((44, 40), (63, 57), (98, 90), (125, 114)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=scale * dx, dx_min=dx, n_steps=n, grid_type=GridType[GRID], grid_vars=gv)
    fs = flt.filter_spec
    spec = O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq)
    with np.errstate(all="ignore"):
        ru, rv = O.filter_func_vec(spec, GRID, u, v, gv)                      # the reference's own f32 path (f32 T, f64 fbar)
        tu, tv = O.filter_func_vec(spec, GRID, u.astype("f8"), v.astype("f8"), {k: x.astype("f8") for k, x in gv.items()})
    rel = lambda a, b: max(np.abs(a[0] - b[0]).max() / np.abs(b[0]).max(), np.abs(a[1] - b[1]).max() / np.abs(b[1]).max())
    res = {}
    for name, cl in (("backward", 2), ("forward", 0)):
        plan.set_tuning(multi_s=8, clenshaw=cl)
        g = flt.apply_to_vector(u, v)
        res[name] = (rel(g, (ru, rv)), rel(g, (tu, tv)), plan.last_kernel()[:28])
    plan.set_tuning(multi_s=8, clenshaw=2)
    print(f"n={n:4d} scale {scale}: reference f32 path vs f64 truth {rel((ru, rv), (tu, tv)):.2e} | " +
          " | ".join(f"{k}: vs reference {a:.2e}, vs f64 truth {b:.2e} ({kn})" for k, (a, b, kn) in res.items()), flush=True)
