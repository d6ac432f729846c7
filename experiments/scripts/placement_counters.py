"""Where do the two launch-time modes of the strip-marching kernels come from (a plan runs its blocked launches in one
of two modes ~10 % apart depending on where its planes landed in HBM)?  Re-creates the config-3 plan NPLANS times in one process (a
dummy allocation in between, so that the planes land elsewhere), runs a few applications on each and prints the HIP-event time per
launch of the dominant kernel.  Run it plainly for the times, and under `rocprofv3 --kernel-trace --pmc <counters>` (one counter group per
run, experiments/scripts/placement_counters.sh) for per-dispatch counters; experiments/scripts/placement_counters.sh joins the two by dispatch order.

    python experiments/scripts/placement_counters.py [NPLANS=8] [APPS=2]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS, clear_plan_cache

nplans = int(sys.argv[1]) if len(sys.argv) > 1 else 8
apps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = T.baseline_workload(3)
fk = wl["fk"]
cls = ALL_KERNELS[GridType[wl["grid"]]]
d = torch.from_numpy(wl["fields"][0]).cuda()
keep = []
for i in range(nplans):
    clear_plan_cache()
    keep.append(torch.empty((16 + 24 * (i % 3)) << 20, dtype=torch.uint8, device="cuda"))   # nudge the allocator
    flt = Filter(grid_type=GridType[wl["grid"]], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"],
                 filter_shape=FilterShape[fk["filter_shape"]])
    lap = cls(*[wl["grid_vars"][k] for k in cls.required_grid_args()])
    plan = lap._plan(_lib.F64, wl["fields"][0].shape, 0)
    flt.apply(d); torch.cuda.synchronize()          # untimed: lazy state buffers
    plan.set_timing(2)
    tot, n = 0.0, 0
    for _ in range(apps):
        flt.apply(d)
        ms, nl, lo, hi = plan.last_kernel_timing()
        tot, n = tot + ms, n + nl
    plan.set_timing(False)
    print(f"PLAN {i}: {1e3 * tot / max(n, 1):.1f} us per launch of {plan.last_kernel()} ({n} launches)", flush=True)
    if len(keep) > 3:
        keep.pop(0)
