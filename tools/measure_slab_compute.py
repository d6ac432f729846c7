"""What ONE rank of an N-GPU row-slab run computes, timed on one GPU: the slab of rank N//2 of BASELINE configs 3 / 4 (strong
scaling: 2400/N rows) and of an N-times taller grid (weak scaling: 2400 rows per rank) with the halo exchange stubbed out
(ghost rows keep stale values: the results are wrong, the launches and their row ranges are the real ones).  This is the
compute side of the scaling curve the pool cannot measure (one GPU per box): speed-up bound = T(1) / T(N).

    python tools/measure_slab_compute.py [config 3|4]
"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.distributed import SlabFilter

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = T.baseline_workload(cfg)
fk = dict(w["fk"]); fk["filter_shape"] = FilterShape[fk["filter_shape"]]
grid, gv, f = w["grid"], w["grid_vars"], w["fields"][0]
ny, nx = f.shape

def timed(sf, local, reps=10):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:      # a GPU that idled while the host folded the plan takes tens of ms to clock up again
        sf.apply_local(local); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sf.apply_local(local)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

base = None
for world in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    rank = world // 2 if not grid.startswith("TRIPOLAR") else world - 1   # tripolar: the rank that also advances the fold band
    sf = SlabFilter(grid, gv, fk, ny, nx, rank=rank, world=world, device=0, exchange="torch")
    n_ex = [0]
    def count_only(tensors, n_ex=n_ex):
        n_ex[0] += 1
        return None
    sf.native_driver = False                  # the Python choreography: its exchange hooks can be stubbed out
    sf._exchange_start = count_only           # no peers here: count the exchanges, move nothing
    sf._exchange_finish = lambda ticket: None
    local = [torch.from_numpy(np.ascontiguousarray(f[None, sf.row_begin:sf.row_end])).cuda()]
    t = timed(sf, local)
    ex = n_ex[0] // 11
    alt = []
    for depth in (6, 5, 4):
        sf.multi_depth = depth
        alt.append(f"depth {depth}: {timed(sf, local)*1e3:.3f} ms")
    sf.multi_depth = 8
    base = base or t
    print(f"config {cfg} strong, {world} ranks: rank {rank} owns {sf.rows_owned} rows (+{sf.halo} ghost rows per side), "
          f"{t*1e3:.3f} ms per application, {ex} exchanges -> compute-side speed-up bound {base/t:.2f}x  [{', '.join(alt)}]", flush=True)
