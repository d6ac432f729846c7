"""Worker of tests/test_gpu_resident.py::test_two_processes_*: filters a small IRREGULAR_WITH_LAND grid (the kind gcmf_apply runs on the
chip by itself) over and over for a few seconds next to another process doing the same on the same GPU, and checks every result bit for
bit against the strip-marching launches (GCMF_RESIDENT=0).  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS


def main():
    seconds, seed = float(sys.argv[1]), int(sys.argv[2])
    shape = (512, 512)
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
    f = f + 0.01 * seed
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=16.0 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    x = torch.from_numpy(f).to("cuda:0")
    os.environ["GCMF_RESIDENT"] = "0"
    ref = flt.apply(x).cpu().numpy()
    del os.environ["GCMF_RESIDENT"]
    plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, shape)
    kernels, n, nan_results, errors, wrong = set(), 0, 0, [], 0
    print("READY", flush=True)
    sys.stdin.readline()   # both workers start their loops together
    t_end = time.time() + seconds
    while time.time() < t_end:
        try:
            plan.last_kernel()
            out = flt.apply(x).cpu().numpy()
            kernels.add(plan.last_kernel().split("<")[0])
        except Exception as e:   # noqa: BLE001  (counted and reported: the parent decides)
            errors.append(str(e)[:200])
            continue
        n += 1
        if np.isnan(out).any() and not np.isnan(ref).any():
            nan_results += 1
        elif not np.array_equal(out, ref, equal_nan=True):
            wrong += 1
    print(json.dumps({"n": n, "kernels": sorted(kernels), "nan_results": nan_results, "wrong": wrong, "errors": errors}), flush=True)


if __name__ == "__main__":
    main()
