"""Why does bench.py's host_path report 2.8 ms per field for a batch of 8 where tools/measure_host_batch.py sees 1.6 for 12?"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
shape = (2400, 3600)
gv = T.scalar_grid_vars("IRREGULAR_WITH_LAND", shape)
dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
f1 = T.random_field(shape, 100)
for name, make in (("stack 12", lambda: np.stack([T.random_field(shape, 100 + b) for b in range(12)])),
                   ("stack 8", lambda: np.stack([T.random_field(shape, 100 + b) for b in range(8)])),
                   ("broadcast 8", lambda: np.ascontiguousarray(np.broadcast_to(f1, (8,) + f1.shape))),
                   ("stack 4", lambda: np.stack([T.random_field(shape, 100 + b) for b in range(4)])),
                   ("stack 2", lambda: np.stack([T.random_field(shape, 100 + b) for b in range(2)]))):
    fb = make()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); r = flt.apply(fb); ts.append(1e3 * (time.perf_counter() - t0) / fb.shape[0])
    print(f"{name:12s}: ms per field, call by call: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); flt.apply(fb); ts.append(1e3 * (time.perf_counter() - t0) / fb.shape[0])
    print(f"{'':12s}  result dropped at once:        " + " ".join(f"{t:.2f}" for t in ts), flush=True)
