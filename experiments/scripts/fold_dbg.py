import os, sys, warnings
sys.path.insert(0, os.getcwd())
os.environ["GCMF_RESIDENT"] = "0"
import numpy as np
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
grid = "TRIPOLAR_POP_WITH_LAND"
shape = (int(sys.argv[1]), int(sys.argv[2])); n = int(sys.argv[3])
f, gv = T.scalar_case(grid, shape)
dx = T.grid_dx_min(grid, gv)
warnings.simplefilter("ignore")
flt = Filter(filter_scale=4.0 * dx, dx_min=dx, n_steps=n, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
outs = []
for zf in (0, 1):
    plan.set_option("zip_fold", zf)
    outs.append(flt.apply(f)); print(plan.last_kernel(), plan.last_kernel_geometry())
a, b = outs
bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
print("differs in", len(bad), "cells; rows", np.unique(bad[:, 0]).tolist()[:40], "cols", np.unique(bad[:, 1]).tolist()[:60])
if len(bad):
    d = np.abs(a - b); print("max abs diff", np.nanmax(d), "at", np.unravel_index(np.nanargmax(d), d.shape), "max |a|", np.nanmax(np.abs(a)))
    top = shape[0] - 1
    print("top row diffs (col: a, b):", [(int(c), float(a[top, c]), float(b[top, c])) for c in np.unique(bad[bad[:, 0] == top][:, 1])[:6]])
