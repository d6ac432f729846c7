"""The on-chip (resident) kernel against the strip-marching launches, on one GPU:

  * the slab ONE rank of an 8-GPU run of BASELINE config 3 owns (300 rows + 2 x 32 ghost rows, 3600 columns), as a ring of one rank
    (ghost rows exchanged with itself: same kernels, streams and exchange calls as on 8 GPUs, no wire) -- ms per application with the
    p2p and the RCCL exchange, resident on / off; and the 600-row slab of a 4-GPU run (does not fit: falls back by itself);
  * one resident launch of L levels alone (no exchange): us per launch and per level;
  * small whole grids through Filter.apply (BASELINE config 1 and others): resident on / off.

    python tools/measure_resident.py
"""
import os, sys, time
os.environ["GCMF_RESIDENT"] = "1"
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.distributed import SlabFilter
from gcm_filters_amd.kernels import ALL_KERNELS


def timed(fn, reps=30):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


nx = 3600
for cfg in (3, 2):
    for rows, halo in ((300, 32), (300, 16), (600, 32)):
        w = T.baseline_workload(cfg, (rows, nx))
        fk = dict(w["fk"])
        if cfg == 2:
            fk["filter_scale"] = 50.0
        f = w["fields"][0]
        for exchange in ("p2p", "native"):
            sf = SlabFilter(w["grid"], w["grid_vars"], fk, rows, nx, halo=halo, device=0, rank=0, world=1, self_ring=True, exchange=exchange)
            local = sf.scatter_from_global([f[None]])
            res = {}
            for resident in (True, False, True, False):
                sf.resident = resident
                sf.engine.plan.last_kernel()
                t = timed(lambda: sf.apply_local(local))
                res.setdefault(resident, []).append((t, sf.engine.plan.last_kernel()))
            fmt = lambda r: " / ".join(f"{t*1e3:.3f}" for t, _ in res[r]) + f" ms [{res[r][0][1]}]"
            print(f"config {cfg} slab {rows} rows + 2 x {halo} ghost rows, n_steps {sf.n_steps}, exchange {exchange}: resident {fmt(True)}   strip-marching {fmt(False)}",
                  flush=True)
            del sf

# one resident launch alone
w = T.baseline_workload(3, (364, nx))
gv, f = w["grid_vars"], w["fields"][0]
plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, (364, nx))
d = torch.from_numpy(f).cuda()
u, v, out = torch.zeros_like(d), torch.zeros_like(d), torch.zeros_like(d)
s = torch.cuda.current_stream().cuda_stream
for L in (4, 8, 16, 32, 64):
    pk = np.full(L, 0.01)
    t = timed(lambda: plan.resident_levels(None, None, u.data_ptr(), v.data_ptr(), d.data_ptr(), None, pk, 0.5, 0.1, _lib.STEP_FIRST, 0, 364, stream=s), reps=50)
    print(f"one resident launch, 364 x {nx} IRREGULAR, L = {L}: {t*1e6:.1f} us ({t*1e6/L:.2f} us per level)  {plan.last_kernel_geometry()}", flush=True)

# small whole grids through Filter.apply
for grid, shape, kw in (("REGULAR", (512, 512), dict(filter_scale=4.0, n_steps=16)), ("REGULAR", (512, 512), dict(filter_scale=32.0)),
                        ("IRREGULAR_WITH_LAND", (512, 512), dict(filter_scale=16.0, filter_shape=FilterShape.TAPER)),
                        ("REGULAR_WITH_LAND", (720, 1440), dict(filter_scale=50.0)), ("IRREGULAR_WITH_LAND", (720, 1440), dict(filter_scale=16.0, filter_shape=FilterShape.TAPER)),
                        ("IRREGULAR_WITH_LAND", (1080, 1440), dict(filter_scale=16.0, filter_shape=FilterShape.TAPER))):
    f, gv = T.scalar_case(grid, shape)
    dx = T.grid_dx_min(grid, gv) if grid != "REGULAR" and grid != "REGULAR_WITH_LAND" else 1.0
    kw = dict(kw); kw["filter_scale"] *= dx
    flt = Filter(dx_min=dx, grid_type=GridType[grid], grid_vars=gv, **kw)
    d = torch.from_numpy(f).cuda()
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    line = []
    for mode in ("1", "0", "1", "0"):
        os.environ["GCMF_RESIDENT"] = mode
        os.environ["GCMF_RESIDENT_MAX_CELLS"] = "100000000"
        plan.last_kernel()
        t = timed(lambda: flt.apply(d), reps=50)
        line.append(f"{'resident' if mode == '1' else 'strips'} {t*1e6:.1f} us [{plan.last_kernel()}]")
    os.environ["GCMF_RESIDENT"] = "1"
    print(f"{grid} {shape} n_steps {flt.n_steps}: " + "   ".join(line), flush=True)
