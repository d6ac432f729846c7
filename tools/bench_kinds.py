#!/usr/bin/env python3
"""`python bench.py --kinds`: a roofline record for EVERY Laplacian kind of SURVEY 8(a) that is not a BASELINE config:
the POP B-grid (20 levels f64, 40 levels f32; reference kernels.py:702-840), MOM5U / MOM5T (:321-432), the two area-weighted regular kinds
and TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED (:127-219, :435-492) at the BASELINE size 2400 x 3600 -- per kind: G cells.steps/s, the
dominant kernel with its HIP-event launch time, launches per application, `frac` = algorithmic bytes of one launch / launch time / 8 TB/s
(the same definition as the headline's `roofline.frac`), and parity against the oracle on the same full-size grid with the polynomial
truncated to a few levels on both sides (the oracle needs ~1 s per level there).  Not timed by the driver: run by hand, the JSON line kept
under profiles/.

    python bench.py --kinds [--kinds-only NAME[,NAME...]] [--no-cpu]
"""
import json
import os
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0

# (name, grid type, dtype, levels, filter shape, filter scale in dx_min units, parity levels)
KINDS = [
    ("bgrid_f64_x20", "VECTOR_B_GRID", "f8", 20, "GAUSSIAN", 40.0, 4),
    ("bgrid_f32_x40", "VECTOR_B_GRID", "f4", 40, "GAUSSIAN", 40.0, 4),
    ("mom5u", "MOM5U", "f8", 1, "TAPER", 16.0, 8),
    ("mom5t", "MOM5T", "f8", 1, "TAPER", 16.0, 8),
    ("tripolar_regular_area", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", "f8", 1, "GAUSSIAN", 50.0, 8),
    ("regular_area", "REGULAR_AREA_WEIGHTED", "f8", 1, "GAUSSIAN", 50.0, 8),
    ("regular_land_area", "REGULAR_WITH_LAND_AREA_WEIGHTED", "f8", 1, "GAUSSIAN", 50.0, 8),
    ("regular", "REGULAR", "f8", 1, "GAUSSIAN", 50.0, 8),
]


def alg_bytes_per_cell_launch(grid, w, nlev, backward, out_w):
    """Every operand plane of ONE blocked launch read once, every result plane written once, per cell and level.
    backward (Clenshaw): two state planes read and written + the constant input read; forward: two state planes read and written + the
    running sum (f64 for f32 state: out_w) read and written.  Coefficient planes are shared by the levels of a batch."""
    ncomp = 2 if grid.startswith("VECTOR") else 1
    coef = {"VECTOR_B_GRID": 8 * w, "VECTOR_C_GRID": 14 * w, "MOM5U": 3 * w + 1, "MOM5T": 3 * w + 1, "IRREGULAR_WITH_LAND": 3 * w + 1,
            "TRIPOLAR_POP_WITH_LAND": 3 * w + 1, "REGULAR": 0, "REGULAR_AREA_WEIGHTED": w, "REGULAR_WITH_LAND": 1,
            "REGULAR_WITH_LAND_AREA_WEIGHTED": 1 + w, "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED": 1 + w}[grid]
    state = ncomp * (5 * w if backward else (4 * w + 2 * out_w))
    return state + coef / nlev


def run_kind(name, grid, dt, nlev, shape_name, scale, n_par, dev, shape, no_cpu):
    import torch

    from gcm_filters_amd import Filter, FilterShape, GridType, _lib, filter as F, testing as T
    from gcm_filters_amd.kernels import ALL_KERNELS, clear_plan_cache

    vec = grid.startswith("VECTOR")
    if vec:
        wl = T.baseline_workload(6, shape, nlev=nlev, f32=(dt == "f4"))
        fields, gv, fk = wl["fields"], wl["grid_vars"], dict(wl["fk"])
    else:
        f, gv = T.scalar_case(grid, shape)
        fields = [f]
        dx = T.grid_dx_min(grid, gv) if ALL_KERNELS[GridType[grid]].is_dimensional else 1.0
        fk = dict(filter_scale=scale * dx, dx_min=dx, filter_shape=shape_name)
    w = 8 if dt == "f8" else 4
    flt = Filter(grid_type=GridType[grid], grid_vars=gv, filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]])
    n = int(flt.n_steps)
    cls = ALL_KERNELS[GridType[grid]]
    plan = cls(*[gv[k] for k in cls.required_grid_args()])._plan(_lib.F64 if w == 8 else _lib.F32, shape, dev.index)
    d_in = [torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in fields]
    run = (lambda: flt.apply_to_vector(d_in[0], d_in[1])) if vec else (lambda: (flt.apply(d_in[0]),))
    t_w = time.perf_counter()
    run()
    torch.cuda.synchronize()
    while time.perf_counter() - t_w < 0.1:
        run()
        torch.cuda.synchronize()
    reps = max(2, int(0.15 / max(1e-4, (time.perf_counter() - t_w) / 8)))
    reps = min(reps, 40)
    blocks = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / reps)
    t_app = sorted(blocks)[2]
    plan.last_kernel()
    plan.set_timing(2)
    run()
    ms, nl, lo, hi = plan.last_kernel_timing()
    tot_ms, launches = plan.last_timing()
    plan.set_timing(False)
    kern, geom = plan.last_kernel(), plan.last_kernel_geometry()
    backward = any(k in kern for k in ("k_ringc<", "k_ringcs<", "k_ringcp<", "k_ringcz<", "stream2c<", "k_cgrid_ring<", "k_resident<"))
    cells = shape[0] * shape[1] * nlev
    bpc = alg_bytes_per_cell_launch(grid, w, nlev, backward, 8)
    rec = {"kind": name, "grid": grid, "dtype": "f64" if w == 8 else "f32", "levels": nlev, "n_steps": n,
           "filter": f"{fk['filter_shape']} filter_scale={fk['filter_scale']:.6g} dx_min={fk['dx_min']:.6g}",
           "value": cells * n / t_app, "unit": "cell-steps/s", "ms_per_application": 1e3 * t_app,
           "ms_per_application_min_max": [1e3 * min(blocks), 1e3 * max(blocks)], "launches_per_application": launches,
           "evaluation": "backward" if backward else "forward (the reference's scheme: the default for this dtype / kind)",
           "roofline": None}
    if nl:
        avg = ms / nl
        targs = kern[kern.index("<") + 1: kern.rindex(">")].split(", ")
        rec["roofline"] = {"bound": "hbm", "kernel": kern, "geometry": geom, "avg_launch_ms": avg, "min_launch_ms": lo, "max_launch_ms": hi,
                           "launches_of_it_per_application": nl, "alg_bytes_per_cell_launch": bpc, "alg_bytes_per_launch": bpc * cells,
                           "achieved": bpc * cells / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bpc * cells / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "template_args": targs}
    if not no_cpu:
        from oracle import gcmf_oracle as O
        spec = flt.filter_spec
        short = F.FilterSpec(n_par, spec.s_max, np.asarray(spec.p)[: n_par + 1], spec.dx_min_sq)
        oshort = O.FilterSpec(n_par, spec.s_max, np.asarray(spec.p)[: n_par + 1], spec.dx_min_sq)
        lev0 = [d if d.ndim == 2 else d[:1] for d in d_in]
        gargs = [gv[k] for k in cls.required_grid_args()]
        host0 = [np.asarray(f if f.ndim == 2 else f[0], dtype=np.float64) for f in fields]
        gv64 = {k: np.asarray(v, dtype=np.float64) for k, v in gv.items()}
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            if vec:
                got = F._create_filter_func_vec(short, cls)(lev0[0], lev0[1], *gargs)
                want = O.filter_func_vec(oshort, grid, host0[0], host0[1], gv64)
            else:
                got = (F._create_filter_func(short, cls)(lev0[0], *gargs),)
                want = (O.filter_func(oshort, grid, host0[0], gv64),)
        t_cpu = time.perf_counter() - t0
        worst, same = 0.0, True
        for g, wv in zip(got, want):
            g = g.cpu().numpy().reshape(shape).astype(np.float64)
            same = same and bool(np.array_equal(np.isnan(g), np.isnan(wv)))
            ok = np.isfinite(wv)
            worst = max(worst, float(np.abs(g[ok] - wv[ok]).max() / np.abs(wv[ok]).max()))
        rec["parity"] = {"rel_err": worst, "nan_pattern_equal": same, "tolerance": 1e-6 if w == 8 else 1e-4,
                         "checked_against": f"oracle (f64), same {shape[0]}x{shape[1]} grid and field (level 0), polynomial truncated to n_steps={n_par} on both sides",
                         "oracle_seconds": t_cpu}
    del d_in, plan
    clear_plan_cache()
    torch.cuda.empty_cache()
    return rec


def main(args):
    import torch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    shape = (args.ny, args.nx)
    only = set(args.kinds_only.split(",")) if args.kinds_only else None
    recs, failed = [], []
    for k in KINDS:
        if only and k[0] not in only:
            continue
        rec = run_kind(*k, dev, shape, args.no_cpu)
        recs.append(rec)
        p = rec.get("parity")
        if p and not (p["rel_err"] <= p["tolerance"] and p["nan_pattern_equal"]):
            failed.append(f"{rec['kind']}: rel_err {p['rel_err']:.3e}, nan_pattern_equal {p['nan_pattern_equal']}")
        time.sleep(0.5)
    summary = {r["kind"]: [round(r["value"] / 1e9, 1), None if not r["roofline"] else round(r["roofline"]["frac"], 3),
                           None if "parity" not in r else float(f"{r['parity']['rel_err']:.1e}")] for r in recs}
    print(json.dumps({"metric": "grid-cells*Laplacian-steps/sec per Laplacian kind (not BASELINE configs)", "unit": "cell-steps/s", "n_gpus": 1,
                      "grid": list(shape), "data": "synthetic", "kinds": recs,
                      "summary": {"cols": ["G cell-steps/s", "frac", "parity rel_err"], **summary}}))
    if failed:
        print("bench.py --kinds: PARITY FAILURE -- " + "; ".join(failed), file=sys.stderr)
        return 1
    return 0
