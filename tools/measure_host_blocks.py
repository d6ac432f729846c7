"""One 2400x3600 f64 host field through Filter.apply: the one-plan path against the row-block pipeline (K = 2, 3, 4)."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, host_blocks, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
for cfg in (3, 2):
    w = T.baseline_workload(cfg)
    fk = w["fk"]
    flt = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]],
                 grid_type=GridType[w["grid"]], grid_vars=w["grid_vars"])
    f = w["fields"][0]
    cells = f.size * flt.n_steps
    ref = None
    for k in (0, 2, 3, 4, 5):
        os.environ["GCMF_HOST_BLOCKS"] = str(k)
        clear_plan_cache()
        t0 = time.perf_counter()
        for _ in range(host_blocks.BUILD_AFTER_CALLS + 1):
            out = flt.apply(f)
        t_build = time.perf_counter() - t0
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); out = flt.apply(f); ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = out
        print(f"config {cfg} {w['grid']:22s} blocks {k}: {min(ts)*1e3:.2f} ms per field (median {np.median(ts)*1e3:.2f}) "
              f"-> {cells/min(ts)/1e9:.0f} G cell-steps/s PCIe-inclusive; first {host_blocks.BUILD_AFTER_CALLS + 1} calls {t_build*1e3:.0f} ms; "
              f"identical {np.array_equal(out, ref, equal_nan=True)}", flush=True)
