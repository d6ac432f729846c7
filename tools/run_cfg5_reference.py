#!/usr/bin/env python3
"""BASELINE config 5 under Filter(evaluation="reference") -- the reference's own scheme, k_cgrid_ringf -- a few applications: the program
tools/profile_cfg5_reference.sh puts under rocprofv3 (bench.py has no switch for the evaluation order of its main workload)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from gcm_filters_amd import Filter, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS

wl = T.baseline_workload(5, (2400, 3600))
fk = wl["fk"]
flt = Filter(grid_type=GridType.VECTOR_C_GRID, grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], evaluation="reference")
cls = ALL_KERNELS[GridType.VECTOR_C_GRID]
plan = cls(*[wl["grid_vars"][k] for k in cls.required_grid_args()])._plan(_lib.F32, (2400, 3600), 0)
d = [torch.from_numpy(f).cuda() for f in wl["fields"]]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = flt.apply_to_vector(d[0], d[1])
torch.cuda.synchronize()
print(plan.last_kernel(), plan.last_kernel_geometry())
