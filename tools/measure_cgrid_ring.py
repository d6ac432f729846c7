#!/usr/bin/env python3
"""BASELINE config 5 (VECTOR_C_GRID, 50 x 2400 x 3600 f32, Gaussian n 44) through the variants of the backward C-grid kernel, in ONE
process, alternating: k_cgrid_stream2c (round 2-4: "stream") against k_cgrid_ring (round 5) with plain loads ("r4", "r5") and with
LDS-direct loads ("m4", "m5") at 4 / 5 levels per launch (suffix hNNN: tallest strip); prints ms per application, G cells.steps/s, the dominant kernel's time per launch, and checks every variant's result
against the first one bit for bit.

    python tools/measure_cgrid_ring.py [--nlev 50] [--reps 3] [--variants stream,r4,m4,m5,m5h120] [--strip H]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlev", type=int, default=50)
    ap.add_argument("--ny", type=int, default=2400)
    ap.add_argument("--nx", type=int, default=3600)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--strip", type=int, default=0)
    ap.add_argument("--variants", default="stream,m4,m5")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    wl = T.baseline_workload(5, (a.ny, a.nx), nlev=a.nlev)
    fk = wl["fk"]
    flt = Filter(grid_type=GridType.VECTOR_C_GRID, grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"],
                 filter_shape=FilterShape[fk["filter_shape"]])
    n_steps = int(flt.n_steps)
    cls = ALL_KERNELS[GridType.VECTOR_C_GRID]
    lap = cls(*[wl["grid_vars"][k] for k in cls.required_grid_args()])
    plan = lap._plan(_lib.F32, (a.ny, a.nx), dev.index)
    d_in = [torch.from_numpy(f).to(dev) for f in wl["fields"]]
    cells = a.ny * a.nx * a.nlev

    def setv(name):
        if name == "stream":
            plan.set_option("cgrid_ring", 0)
        else:
            plan.set_option("cgrid_ring", 1)   # (round 5's A/B runs also had "r" variants with plain loads; only the LDS-direct form "m" is built now)
            plan.set_option("cgrid_ring_smax", int(name[1]))
            # optional suffixes: x0 = groups dealt round-robin to the XCDs instead of contiguous ranges, hNNN = tallest strip
            import re
            plan.set_option("cgrid_ring_hmax", int(re.search(r"h(\d+)", name).group(1)) if re.search(r"h(\d+)", name) else 0)
            # cN: how many levels carry their previous row's scaled copies in registers (c1: round 5's form; default: all; m6: c1 unless c6)
            plan.set_option("cgrid_ring_ncarry", int(re.search(r"c(\d+)", name).group(1)) if re.search(r"c(\d+)", name) else 0)
        plan.set_tuning(multi_s=8, strip_rows=a.strip, xcd_remap=0 if "x0" in name else 1)

    ref = None
    results = {}
    for rnd in range(a.rounds):
        for name in a.variants.split(","):
            setv(name)
            out = flt.apply_to_vector(d_in[0], d_in[1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                out = flt.apply_to_vector(d_in[0], d_in[1])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.reps
            plan.last_kernel()
            plan.set_timing(2)
            flt.apply_to_vector(d_in[0], d_in[1])
            ms, nl, lo, hi = plan.last_kernel_timing()
            tot_ms, launches = plan.last_timing()
            plan.set_timing(False)
            kern, geom = plan.last_kernel(), plan.last_kernel_geometry()
            same = None
            if rnd == 0:
                o = [x.cpu().numpy() for x in out]
                if ref is None:
                    ref = o
                else:
                    same = all(np.array_equal(r, g, equal_nan=True) for r, g in zip(ref, o))
                    if not same:
                        print("   max |diff| =", max(float(np.nanmax(np.abs(r - g))) for r, g in zip(ref, o)), "of", float(np.nanmax(np.abs(ref[0]))))
                del o
            results.setdefault(name, []).append(dt)
            print(f"{name:7s} round {rnd}: {dt * 1e3:8.2f} ms / application = {cells * n_steps / dt / 1e9:7.1f} G   dominant {kern}: "
                  f"{ms / max(nl, 1):.3f} ms x {nl} (min {lo:.3f} max {hi:.3f}), launches {launches}, {geom}"
                  + ("" if same is None else f"   same bits as the first variant: {same}"), flush=True)
    print("redo workgroups (gcmf_ring_fallbacks):", plan.ring_fallbacks() if hasattr(plan, "ring_fallbacks") else "n/a")
    for name, ts in results.items():
        print(f"{name:7s} best {min(ts) * 1e3:8.2f} ms = {cells * n_steps / min(ts) / 1e9:7.1f} G cells.steps/s")


if __name__ == "__main__":
    main()
