#!/usr/bin/env python3
"""Strong scaling of a BATCH of fields over 8 GPUs, emulated on one GPU: one 2400 x 3600 field cut 8 ways leaves
300-row slabs whose strips are 11 rows tall and march 11 + 2 S rows (2.45 x redundant) -- bound ~3.6 x.  A batch of `nb` fields (time
levels of one variable: the reference's own leading dims) gives every wave nb x taller strips again.

For nb in --batches:   T1 = the whole grid, nb fields, one plan (Filter.apply on a device tensor)
                       T8 = the slab ONE rank of an 8-rank run owns, nb fields, with its halo exchanges
                            config 3 (periodic): a ring of one rank (both neighbours are itself: same packing, kernels, streams, no wire),
                                                 exchange = p2p / native (RCCL to self);
                            config 4 (tripolar): the top rank (it also advances the seam rows), exchange stubbed out + the exchange cost
                                                 measured on config 3's ring added
                       bound on the 8-GPU speed-up = T1 / T8

    python tools/measure_batched_scaling.py [--config 3|4] [--batches 1,2,4,8,16] [--world 8]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.distributed import SlabFilter


def timed(fn, reps):
    t_w, n_w = time.perf_counter(), 0
    while time.perf_counter() - t_w < 0.05 or n_w < 5:   # (at least five: the first calls of a new kernel family load its code objects)
        fn()
        torch.cuda.synchronize()
        n_w += 1
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--batches", default="1,2,4,8,16")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--halo", type=int, default=0, help="ghost rows per side of the slab (0 = SlabFilter's default: as deep as the filter, <= 64)")
    ap.add_argument("--overlap", type=int, default=-1, help="1: split every launch that is followed by an exchange into edge strips + interior and "
                    "run the exchange beside the interior; 0: never; -1: SlabFilter's own choice")
    a = ap.parse_args()
    ny, nx = 2400, 3600
    wl = T.baseline_workload(a.config, (ny, nx))
    fk = dict(wl["fk"])
    f = wl["fields"][0]
    flt = Filter(grid_type=GridType[wl["grid"]], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"],
                 filter_shape=FilterShape[fk["filter_shape"]])
    n = int(flt.n_steps)
    rows = ny // a.world
    # the exchange cost on the periodic config-3 slab (a ring of one rank): with and without the exchange, same launches
    w3 = T.baseline_workload(3, (rows, nx)) if a.config != 3 else None
    print(f"config {a.config}: {wl['grid']} {ny}x{nx}, n_steps {n}; slab of one of {a.world} ranks: {rows} rows", flush=True)
    for nb in [int(x) for x in a.batches.split(",")]:
        fb = np.stack([f + 0.01 * k for k in range(nb)])
        d = torch.from_numpy(fb).cuda()
        t1 = timed(lambda: flt.apply(d), a.reps)
        del d
        res = {}
        if a.config == 3:
            wr = T.baseline_workload(3, (rows, nx))
            fr = np.stack([wr["fields"][0] + 0.01 * k for k in range(nb)])
            for ex in ("none", "p2p", "native"):
                sf = SlabFilter(wr["grid"], wr["grid_vars"], dict(wr["fk"]), rows, nx, device=0, rank=0, world=1, self_ring=True,
                                exchange="p2p" if ex == "none" else ex, halo=a.halo or None)
                if a.overlap >= 0:
                    sf.overlap = bool(a.overlap)
                if ex == "none":
                    sf.native_driver = False
                    sf._exchange_start = lambda tensors: None
                    sf._exchange_finish = lambda ticket: None
                local = sf.scatter_from_global([fr])
                res[ex] = (timed(lambda: sf.apply_local(local), a.reps), sf.halo)
                del sf, local
        else:
            rank = a.world - 1
            sf = SlabFilter(wl["grid"], wl["grid_vars"], fk, ny, nx, rank=rank, world=a.world, device=0, exchange="torch", halo=a.halo or None)
            if a.overlap >= 0:
                sf.overlap = bool(a.overlap)
            sf.native_driver = False
            sf._exchange_start = lambda tensors: None
            sf._exchange_finish = lambda ticket: None
            local = [torch.from_numpy(np.ascontiguousarray(fb[:, sf.row_begin:sf.row_end])).cuda()]
            res["none"] = (timed(lambda: sf.apply_local(local), a.reps), sf.halo)
            del sf, local
            fr = np.stack([w3["fields"][0] + 0.01 * k for k in range(nb)])
            tt = {}
            for ex in ("none", "p2p", "native"):
                s3 = SlabFilter(w3["grid"], w3["grid_vars"], dict(w3["fk"]), rows, nx, device=0, rank=0, world=1, self_ring=True,
                                exchange="p2p" if ex == "none" else ex, halo=a.halo or None)
                if a.overlap >= 0:
                    s3.overlap = bool(a.overlap)
                if ex == "none":
                    s3.native_driver = False
                    s3._exchange_start = lambda tensors: None
                    s3._exchange_finish = lambda ticket: None
                l3 = s3.scatter_from_global([fr])
                tt[ex] = timed(lambda: s3.apply_local(l3), a.reps)
                del s3, l3
            for ex in ("p2p", "native"):
                res[ex] = (res["none"][0] + max(0.0, tt[ex] - tt["none"]), res["none"][1])
        cells = ny * nx * n * nb
        line = f"  batch {nb:3d}: whole grid {t1 * 1e3:8.3f} ms ({cells / t1 / 1e9:6.1f} G)"
        for ex, (t8, halo) in res.items():
            line += f" | slab, exchange {ex:6s}: {t8 * 1e3:7.3f} ms (halo {halo}) -> bound {t1 / t8:4.2f} x"
        print(line, flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
