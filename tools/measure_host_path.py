"""PCIe-inclusive rate: Filter.apply on host numpy arrays (H2D + filter + D2H), next to the HBM-resident rate, and
where the host path's time goes (fresh output pages vs the copies)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T, _lib
from gcm_filters_amd.kernels import ALL_KERNELS
shape = (2400, 3600)
gv = T.scalar_grid_vars("IRREGULAR_WITH_LAND", shape)
dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
f = T.random_field(shape, 100)
t0 = time.perf_counter(); flt.apply(f); t_first = time.perf_counter() - t0
ts = []
for _ in range(5):
    t0 = time.perf_counter(); flt.apply(f); ts.append(time.perf_counter() - t0)
d = torch.from_numpy(f).cuda()
flt.apply(d); torch.cuda.synchronize()
td = []
for _ in range(5):
    t0 = time.perf_counter(); flt.apply(d); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
cells = shape[0] * shape[1] * flt.n_steps
print(f"first call (plan build + upload of 8 grid planes): {t_first*1e3:.1f} ms")
print(f"host numpy in/out: {min(ts)*1e3:.2f} ms -> {cells/min(ts)/1e9:.1f} G cell-steps/s (PCIe-inclusive)")
print(f"HBM-resident tensor: {min(td)*1e3:.2f} ms -> {cells/min(td)/1e9:.1f} G cell-steps/s")
# breakdown of the host path
lap = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)
plan = lap._plan(_lib.F64, shape)
p = np.asarray(flt.filter_spec.p, dtype=np.float64)
c = 2 / flt.filter_spec.s_max
out = np.empty(shape)
out[:] = 0  # touched pages
tw = []
for _ in range(5):
    t0 = time.perf_counter(); plan.apply(p, c, [f.ctypes.data], [out.ctypes.data], 1, device_ptrs=False); tw.append(time.perf_counter() - t0)
tf = []
for _ in range(5):
    t0 = time.perf_counter(); o2 = np.empty(shape); plan.apply(p, c, [f.ctypes.data], [o2.ctypes.data], 1, device_ptrs=False); tf.append(time.perf_counter() - t0)
    del o2
print(f"bare gcmf_apply, host pointers, output pages already touched: {min(tw)*1e3:.2f} ms; fresh np.empty output: {min(tf)*1e3:.2f} ms")
