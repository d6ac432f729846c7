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
    import warnings
    seconds, seed = float(sys.argv[1]), int(sys.argv[2])
    delay_before = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # idle seconds before the loop (after "go")
    idle_after = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0     # idle seconds after it (the process stays alive: an idle notebook)
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
    kernels, n, nan_results, errors, wrong, paths = set(), 0, 0, [], 0, set()
    print("READY", flush=True)
    sys.stdin.readline()   # both workers start their loops together
    time.sleep(delay_before)
    t_end = time.time() + seconds
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        while time.time() < t_end:
            try:
                plan.last_kernel()
                out = flt.apply(x).cpu().numpy()
                kernels.add(plan.last_kernel().split("<")[0])
                paths.add(flt.last_path)
            except Exception as e:   # noqa: BLE001  (counted and reported: the parent decides)
                errors.append(str(e)[:200])
                continue
            n += 1
            if np.isnan(out).any() and not np.isnan(ref).any():
                nan_results += 1
            elif not np.array_equal(out, ref, equal_nan=True):
                wrong += 1
    lock_warnings = [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning) and "on-chip" in str(w.message)]
    time.sleep(idle_after)
    print(json.dumps({"n": n, "kernels": sorted(kernels), "nan_results": nan_results, "wrong": wrong, "errors": errors,
                      "paths": sorted(p or "none" for p in paths), "lock_warnings": lock_warnings, "status": _lib.resident_status(0),
                      "path_counts": plan.path_counts()}), flush=True)


if __name__ == "__main__":
    main()
