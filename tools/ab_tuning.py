"""In-process A/B of two plan tunings on one BASELINE config: the same process, buffers and clocks, settings alternated.
(Separate processes of the same binary differ by up to 10 % on this pool -- box and power state -- so A/B across processes
needs many samples; this does not.)

    python tools/ab_tuning.py <config 2|3|4> "xcd_remap=0" "xcd_remap=1" [rounds]
"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS

cfg = int(sys.argv[1])
settings = [dict((k, int(v)) for k, v in (kv.split("=") for kv in s.split(",") if kv)) for s in sys.argv[2:4]]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
w = T.baseline_workload(cfg)
fk = w["fk"]
flt = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]],
             grid_type=GridType[w["grid"]], grid_vars=w["grid_vars"])
f0 = w["fields"][0]
plan = ALL_KERNELS[GridType[w["grid"]]](**w["grid_vars"])._plan(_lib.dtype_code(f0.dtype.str[1:]), f0.shape[-2:])
nrep = 2 if cfg == 5 else 10
d = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in w["fields"]]
def run():
    return flt.apply(d[0]) if len(d) == 1 else flt.apply_to_vector(d[0], d[1])
res = [[], []]
for r in range(rounds):
    for i, s in enumerate(settings):
        plan.set_tuning(**{"multi_s": 8, **s})
        run(); torch.cuda.synchronize()
        plan.set_timing(2)
        for _ in range(max(1, nrep // 2)):
            run()
        torch.cuda.synchronize()
        ms, n, lo, hi = plan.last_kernel_timing()
        plan.set_timing(False)
        t0 = time.perf_counter()
        for _ in range(nrep):
            run()
        torch.cuda.synchronize()
        res[i].append(((time.perf_counter() - t0) / nrep * 1e3, ms / max(n, 1) * 1e3))
for i, s in enumerate(settings):
    a = np.array(res[i])
    print(s, f"apply {a[:,0].mean():.4f} ms (min {a[:,0].min():.4f})  dominant launch {a[:,1].mean():.1f} us (min {a[:,1].min():.1f})", plan.last_kernel())
