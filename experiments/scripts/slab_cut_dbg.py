import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gcm_filters_amd import testing as T
from gcm_filters_amd.distributed import SlabFilter
for rows in (300, 600, 1200):
    wr = T.baseline_workload(3, (rows, 3600))
    sf = SlabFilter(wr["grid"], wr["grid_vars"], dict(wr["fk"]), rows, 3600, device=0, rank=0, world=1, self_ring=True, exchange="p2p")
    print(rows, "rows_owned", sf.rows_owned, "halo", sf.halo, "cut", sf.backward_cut, "plan cut", sf.engine.plan.clenshaw_cut(sf.n_steps), flush=True)
    del sf
