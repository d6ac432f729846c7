"""Cost of one halo exchange, measured as a ring of ONE rank on one GPU (the slab keeps ghost rows and both neighbours are itself: same
packing, posting order, kernels and streams as on N GPUs, no wire): SlabFilter on a 300-row periodic grid (the slab of one rank of
an 8-GPU run of BASELINE config 3), exchange = native (RCCL send / recv to self on a side stream) / p2p (mailbox + flag kernels on the
compute stream) / none (ghost rows left stale: the compute alone).

    python tools/measure_exchange.py [rows=300] [halo=16]
"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import testing as T
from gcm_filters_amd.distributed import SlabFilter

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300
halo = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nx = 3600
w = T.baseline_workload(3, (rows, nx))
fk = dict(w["fk"])
f = w["fields"][0]
for exchange in ("none", "native", "p2p", "none", "native", "p2p"):
    sf = SlabFilter(w["grid"], w["grid_vars"], fk, rows, nx, halo=halo, device=0, rank=0, world=1, self_ring=True,
                    exchange="p2p" if exchange == "none" else exchange)
    if len(sys.argv) > 3:
        sf.overlap = bool(int(sys.argv[3]))       # force the edge / interior split on or off
    if exchange == "none":
        sf.native_driver = False                    # (the Python choreography: its exchange hooks can be stubbed out)
        sf._exchange_start = lambda tensors: None
        sf._exchange_finish = lambda ticket: None
    local = sf.scatter_from_global([f[None]])
    for _ in range(3):
        sf.apply_local(local)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        sf.apply_local(local)
    t_host = (time.perf_counter() - t0) / reps      # the host side alone: Python choreography + C calls, nothing waited for
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nex = -(-sf.n_steps // halo) + 1
    print(f"exchange={exchange:7s} {rows} rows, halo {halo}, n_steps {sf.n_steps}: {dt*1e3:.3f} ms per application ({nex} exchanges); host enqueue {t_host*1e3:.3f} ms", flush=True)
