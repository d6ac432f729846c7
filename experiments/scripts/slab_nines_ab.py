"""Nine against eight levels per launch on the slab of one rank (ring of one rank, exchanges stubbed), alternating in one process."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gcm_filters_amd import testing as T
from gcm_filters_amd.distributed import SlabFilter
def timed(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best
for rows in (400, 500, 600, 800, 1200):
    wr = T.baseline_workload(3, (rows, 3600))
    for nb in (1, 4):
        fr = np.stack([wr["fields"][0] + 0.01 * k for k in range(nb)])
        out = []
        for rep in range(2):
            for nines in ("1", "0"):
                os.environ["GCMF_SLAB_NINES"] = nines
                sf = SlabFilter(wr["grid"], wr["grid_vars"], dict(wr["fk"]), rows, 3600, device=0, rank=0, world=1, self_ring=True, exchange="p2p")
                sf.native_driver = False
                sf._exchange_start = lambda tensors: None
                sf._exchange_finish = lambda ticket: None
                local = sf.scatter_from_global([fr])
                t = timed(lambda: sf.apply_local(local))
                out.append(f"{'nines' if max(sf._cut_for(nb)) == 9 else 'eights'} {t*1e6:.1f}")
                del sf, local
        print(rows, "rows, batch", nb, ":", " | ".join(out), flush=True)
