"""bench.py's `host_path` leg: BASELINE config 3 on HOST buffers (PCIe-inclusive, never `value`)."""
import os
import time

import numpy as np

def run_host_path(dev, args):
    """SURVEY 8d "H2D / D2H reported separately": BASELINE config 3 with HOST (numpy) buffers, the reference's default call shape
    (`filter_func(field, *grid_args)` on host arrays, reference filter.py:181-214) -- PCIe-inclusive, never `value`.  Milliseconds per
    2400 x 3600 f64 field (69.12 MB): the copies alone (pageable and page-locked), the recurrence alone (field resident in HBM), the
    three in sequence through gcmf_apply with host pointers (one plan), Filter.apply's row-block pipeline that overlaps them
    (gcm_filters_amd/host_blocks.py), and a batch of 8 fields streamed through two HBM staging slots."""
    import torch

    from gcm_filters_amd import Filter, FilterShape, GridType, testing as T

    wl = T.baseline_workload(3, (args.ny, args.nx))
    fk = wl["fk"]
    flt = Filter(grid_type=GridType[wl["grid"]], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"],
                 filter_shape=FilterShape[fk["filter_shape"]])
    f = wl["fields"][0]
    mb = f.nbytes / 1e6

    def best(fn, reps=5, sync=True):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            if sync:
                torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return 1e3 * min(ts)
    d = torch.from_numpy(f).to(dev)
    flt.apply(d)
    torch.cuda.synchronize()
    rec = {"field_MB": mb}   # config 3 on HOST buffers, ms per 2400x3600 f64 field; PCIe-inclusive, never `value`
    rec["recurrence_ms_field_resident"] = best(lambda: flt.apply(d))
    hbuf = torch.empty_like(d, device="cpu")
    rec["h2d_ms_pageable"] = best(lambda: d.copy_(torch.from_numpy(f)))
    rec["d2h_ms_pageable"] = best(lambda: hbuf.copy_(d))
    pin = torch.from_numpy(f).pin_memory()
    rec["h2d_ms_page_locked"] = best(lambda: d.copy_(pin, non_blocking=True))
    rec["d2h_ms_page_locked"] = best(lambda: pin.copy_(d, non_blocking=True))
    rec["h2d_GBps_page_locked"] = mb / rec["h2d_ms_page_locked"]
    from gcm_filters_amd.kernels import clear_plan_cache
    before = os.environ.get("GCMF_HOST_BLOCKS")
    try:
        os.environ.pop("GCMF_HOST_BLOCKS", None)
        for _ in range(4):                         # (the row-block pipeline is built at the third single-field host call on a plan)
            out = flt.apply(f)
        rec["row_block_pipeline_ms"] = best(lambda: flt.apply(f), sync=False)
        want = flt.apply(d).cpu().numpy()
        rec["row_block_pipeline_same_bits"] = bool(np.array_equal(out, want, equal_nan=True))
        out = want = None
        clear_plan_cache()                         # (a plan remembers its pipeline: the in-sequence figure needs a fresh one)
        time.sleep(0.5)                            # (the block plans' memory is being scrubbed: see free_gpu)
        os.environ["GCMF_HOST_BLOCKS"] = "0"       # one plan: upload, recurrence, download in sequence (gcmf_apply with host pointers)
        flt.apply(f)
        rec["one_plan_in_sequence_ms"] = best(lambda: flt.apply(f), sync=False)
    finally:
        if before is None:
            os.environ.pop("GCMF_HOST_BLOCKS", None)
        else:
            os.environ["GCMF_HOST_BLOCKS"] = before
    fb = np.ascontiguousarray(np.broadcast_to(f, (8,) + f.shape))
    for _ in range(3):   # (the first calls with a new result size pay for its page-locked result buffers: 5-8 ms per field, then the pool reuses them)
        flt.apply(fb)
    rec["batch_of_8_ms_per_field"] = best(lambda: flt.apply(fb), reps=4, sync=False) / 8
    n = int(flt.n_steps)
    rec["n_steps"] = n
    rec["cells_steps_per_s_host_buffers"] = {"one_plan_in_sequence": f.size * n / (rec["one_plan_in_sequence_ms"] * 1e-3),
                                             "row_block_pipeline": f.size * n / (rec["row_block_pipeline_ms"] * 1e-3),
                                             "batch_of_8": f.size * n / (rec["batch_of_8_ms_per_field"] * 1e-3)}
    return rec
