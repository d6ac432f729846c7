#!/usr/bin/env python3
"""bench.py -- throughput of the fused iterated-Laplacian filter on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--scaling strong|weak] [--no-cpu] [--no-extra]

A "step" is ONE whole filter application (all n_steps Chebyshev/Laplacian steps) of the workload's field(s),
inputs already resident in HBM.  metric = grid-cells * Laplacian-steps / second (BASELINE.json).
Default workload = BASELINE config 3, the one the north-star target is quoted on:
IRREGULAR_WITH_LAND 2400x3600 fp64, Taper filter, filter_scale = 16 dx_min  =>  n_steps 63.

N > 1: one rank per GPU.  Under torch.distributed.run the ranks are already there; started as a plain
`python bench.py --gpus N` the script launches its own N ranks (a child `python -m torch.distributed.run`, started
before this process touches a GPU).  Scalar configs are cut into row slabs along y with halo rows exchanged over
xGMI (gcm_filters_amd/distributed.py); the default is STRONG scaling (the one BASELINE grid cut N ways), a weak
figure (N x ny rows) is measured in the same run and printed as `weak`.  Config 5 (50 levels) shards its levels
over the ranks: no communication at all.

Prints ONE JSON line on rank 0 (task contract) with `roofline`, `cpu_baseline`, `parity` and `extra_configs`.
The run FAILS (exit 1) if the timed workload's output differs from the oracle / the reference-generated probes.
"""
import argparse
import json
import math
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0   # measured float4-copy ceiling, same guide
GOLDEN_FULLSIZE = os.path.join(REPO, "tests", "golden", "reference_fullsize.npz")
# (config, scale override) -> key in reference_fullsize.npz (outputs of the imported reference, tests/golden/make_golden.py)
FIXTURE_KEY = {(2, 0.0): "cfg2_n56", (2, 50.0): "cfg2_n56", (2, 10.0): "cfg2_n11", (3, 0.0): "cfg3_n63", (4, 0.0): "cfg4_n56",
               (5, 0.0): "cfg5_lev0_n44"}
NCOEF = {"REGULAR": 0, "REGULAR_WITH_LAND": None, "IRREGULAR_WITH_LAND": 3, "TRIPOLAR_POP_WITH_LAND": 3,
         "VECTOR_C_GRID": 14, "VECTOR_B_GRID": 8}


def b_alg(grid, w, f, L):
    """SURVEY 8d / BASELINE.md 4: algorithmic bytes per cell and Laplacian step of the ONE-PASS-PER-STEP streaming model
    (read T_{k-1}, T_{k-2}, fbar; write T_k, fbar; + folded coefficient planes shared by L levels)."""
    ncomp = 2 if grid.startswith("VECTOR") else 1
    coef = 1.0 if NCOEF[grid] is None else NCOEF[grid] * w
    return ncomp * (3 * w + 2 * f) + coef / L


def min_bytes_per_cell_launch(grid, w, f, L, backward=False):
    """Compulsory HBM bytes per cell of ONE temporally blocked launch, whatever its depth S: every operand plane
    read once (T_{k-1}, T_{k-2}, fbar, coefficients), every result written once (T_{k+S-1}, T_{k+S-2}, fbar).
    backward (k_ringc, Clenshaw): two state planes read and written, the constant input and its land byte read, no fbar."""
    ncomp = 2 if grid.startswith("VECTOR") else 1
    coef = 1.0 if NCOEF[grid] is None else NCOEF[grid] * w
    if backward and ncomp == 2:   # per component: two state planes read and written, the input read; coefficients shared by L levels
        return ncomp * 5 * w + coef / L
    if backward:
        return 5 * w + 1 + coef
    return ncomp * 2 * (2 * w + f) + coef / L


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (numpy restatement of the reference) on the host cores
# ------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, budget_steps):
    """(i) SURVEY 8d: single process / thread -- what the reference does for one 2-D field -- on a bounded sample: the
    SAME grid and field (one level of a batched workload), polynomial truncated to `budget_steps` steps if longer.
    Returns (record, oracle outputs, n_steps actually run)."""
    from oracle import gcmf_oracle as O

    fk = wl["fk"]
    full = O.make_spec(fk["filter_scale"], fk["dx_min"], fk["filter_shape"])
    n = min(budget_steps, full.n_steps)
    spec = O.FilterSpec(n, full.s_max, full.p[: n + 1], full.dx_min_sq)
    fields = [f if f.ndim == 2 else f[0] for f in wl["fields"]]
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        if len(fields) == 2:
            res = O.filter_func_vec(spec, wl["grid"], fields[0], fields[1], wl["grid_vars"])
        else:
            res = (O.filter_func(spec, wl["grid"], fields[0], wl["grid_vars"]),)
    dt = time.perf_counter() - t0
    ny, nx = fields[0].shape
    rec = {"value": ny * nx * n / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
           "sample": f"same {ny}x{nx} grid and field, 1 level, "
                     + ("whole polynomial" if n == full.n_steps else f"polynomial truncated to n_steps={n}")
                     + f" (n_steps={n}, {dt:.1f} s)",
           "host": {"cpu_count": os.cpu_count(), "cpu_model": _cpu_model(), "numpy": np.__version__}}
    return rec, res, n


def _pool_level(job):
    """One level of config 5 through the oracle (worker of cpu_baseline_pool); inputs rebuilt from seeds in the worker."""
    cfg, ny, nx, level, n = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from gcm_filters_amd import testing as T
    from oracle import gcmf_oracle as O

    wl = T.baseline_workload(cfg, (ny, nx), levels=[level])
    fk = wl["fk"]
    full = O.make_spec(fk["filter_scale"], fk["dx_min"], fk["filter_shape"])
    spec = O.FilterSpec(n, full.s_max, full.p[: n + 1], full.dx_min_sq)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        O.filter_func_vec(spec, wl["grid"], wl["fields"][0][0], wl["fields"][1][0], wl["grid_vars"])
    return time.perf_counter() - t0


def cpu_baseline_pool(cfg, ny, nx, nlev, budget_steps):
    """(ii) SURVEY 8d / BASELINE.md 3: an os.cpu_count()-way process pool of the oracle over the levels of a batched
    workload -- the analogue of the reference's dask="parallelized" over non-core dims (gcm_filters/filter.py:485).
    Bounded: one level per worker, polynomial truncated to `budget_steps`; workers capped by free memory (~3 GB each)."""
    import multiprocessing as mp

    workers = min(os.cpu_count() or 1, nlev)
    try:
        avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] * 1024
        workers = max(1, min(workers, int(avail // (3 << 30))))
    except Exception:
        pass
    jobs = [(cfg, ny, nx, l, budget_steps) for l in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        per = pool.map(_pool_level, jobs)
    wall = time.perf_counter() - t0
    return {"value": workers * ny * nx * budget_steps / max(per), "unit": "cell-steps/s", "cores": workers, "kind": "port",
            "sample": f"{workers} levels of the {nlev}, one per worker process, polynomial truncated to n_steps={budget_steps}; "
                      f"slowest worker {max(per):.1f} s, pool wall incl. start-up and input generation {wall:.1f} s",
            "host": {"cpu_count": os.cpu_count(), "cpu_model": _cpu_model(), "numpy": np.__version__}}


# ------------------------------------------------------------------------------------------------------------------
# parity helpers
# ------------------------------------------------------------------------------------------------------------------
def rel_err_and_nan(got, want):
    """max |got - want| / max |want| over the finite cells of `want`, and whether the NaN patterns agree."""
    worst, same = 0.0, True
    for g, w in zip(got, want):
        g, w = np.asarray(g, dtype=np.float64), np.asarray(w, dtype=np.float64)
        same = same and bool(np.array_equal(np.isnan(g), np.isnan(w)))
        ok = np.isfinite(w)
        if ok.any():
            with np.errstate(invalid="ignore"):
                worst = max(worst, float(np.nanmax(np.abs(g[ok] - w[ok])) / np.abs(w[ok]).max()))
    return worst, same


def golden_probe_check(cfg, scale, shape, outs, row_begin=0, row_end=None):
    """Compare (device or host) outputs `outs` (ncomp arrays (..., rows, nx); rows = [row_begin, row_end) of the grid)
    with the probes the imported reference produced for this BASELINE config (tests/golden/reference_fullsize.npz).
    Returns None when no fixture covers the workload, else dict(rel_err, n_probes, key); vector configs: level 0."""
    from gcm_filters_amd import testing as T

    key = FIXTURE_KEY.get((cfg, float(scale)))
    if key is None or tuple(shape) != T.BASELINE_SHAPE or not os.path.exists(GOLDEN_FULLSIZE):
        return None
    with np.load(GOLDEN_FULLSIZE) as z:
        want = np.atleast_2d(z[key + "/probe"])
    jj, ii = T.probe_points(T.BASELINE_SHAPE)
    row_end = shape[0] if row_end is None else row_end
    mine = (jj >= row_begin) & (jj < row_end)
    got = np.full(want.shape, np.nan)
    for c, o in enumerate(outs):
        lev0 = o if o.ndim == 2 else o.reshape(-1, o.shape[-2], o.shape[-1])[0]
        v = lev0[jj[mine] - row_begin, ii[mine]]
        got[c, mine] = v.double().cpu().numpy() if hasattr(v, "cpu") else np.asarray(v, dtype=np.float64)
    return {"key": key, "got": got, "want": want, "mine": mine}


def finish_probe_check(chk):
    got, want, mine = chk["got"], chk["want"], chk["mine"]
    err = float(np.abs(got[:, mine] - want[:, mine]).max() / np.abs(want).max()) if mine.any() else 0.0
    return {"fixture": "tests/golden/reference_fullsize.npz:" + chk["key"], "n_probes": int(mine.sum()) * got.shape[0],
            "rel_err": err, "source": "imported reference (tests/golden/make_golden.py --fullsize)"}


# ------------------------------------------------------------------------------------------------------------------
def load_traffic(cfg, kernel_ran, geometry=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes -- only if the record names
    the kernel that actually ran in this process (gcmf_last_kernel) at the launch geometry it ran with."""
    tf = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if not os.path.exists(tf):
        return None, "no profiles/hbm_traffic.json"
    try:
        tab = json.load(open(tf))
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable profiles/hbm_traffic.json: {e}"
    rec = tab.get(f"config{cfg}")
    if not rec:
        return None, f"no record for config {cfg}"
    prof = rec.get("kernel", "").replace("void ", "").strip()
    key = f"config{cfg}"
    if kernel_ran and prof != kernel_ran:   # e.g. the forward-recurrence kernel of the same config, profiled in the same round
        for k, alt in tab.items():
            if k.startswith(f"config{cfg}_") and isinstance(alt, dict) and alt.get("kernel", "").replace("void ", "").strip() == kernel_ran \
                    and alt.get("round") == rec.get("round"):
                rec, prof, key = alt, kernel_ran, k
                break
    if not kernel_ran or prof != kernel_ran:
        return None, f"profiled kernel '{prof}' is not the kernel that ran ('{kernel_ran}'): traffic withheld"
    # the bytes a strip-marched kernel moves depend on its launch geometry (strip height, strip count, XCD order, grid):
    # the record must have been profiled at the geometry this run used (gcmf_last_kernel_geometry)
    want = rec.get("geometry")
    if not want:
        return None, f"profiles/hbm_traffic.json:{key} has no launch geometry recorded: traffic withheld"
    diff = {k: (want.get(k), (geometry or {}).get(k)) for k in ("H", "nstrips", "nwx", "xcd", "grid", "rows")
            if want.get(k) != (geometry or {}).get(k)}
    if diff:
        return None, f"profiles/hbm_traffic.json:{key} was profiled at another launch geometry {diff} (profiled, ran): traffic withheld"
    return rec, f"profiles/hbm_traffic.json:{key} <- {rec.get('source')} ({prof}, geometry {want})"


def run_single(cfg, args, dev, steps, warmup, scale=0.0, levels=None, tuned=False, evaluation="auto"):
    """Time `steps` filter applications of BASELINE config `cfg` on this process's GPU.  Returns a dict with the raw
    measurements, the workload and the device outputs of the last application."""
    import torch

    from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
    from gcm_filters_amd.kernels import ALL_KERNELS

    nlev = args.nlev if (args.nlev and cfg == args.config) else 0
    wl = T.baseline_workload(cfg, (args.ny, args.nx), nlev=nlev, f32=args.f32, f64=args.f64, scale=scale, levels=levels)
    grid, fk = wl["grid"], wl["fk"]
    itemsize = wl["fields"][0].dtype.itemsize
    nbatch = 1 if wl["fields"][0].ndim == 2 else wl["fields"][0].shape[0]
    flt = Filter(grid_type=GridType[grid], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"], dx_min=fk["dx_min"],
                 filter_shape=FilterShape[fk["filter_shape"]], evaluation=evaluation)
    n_steps = int(flt.n_steps)
    cls = ALL_KERNELS[GridType[grid]]
    lap = cls(*[wl["grid_vars"][k] for k in cls.required_grid_args()])

    def make_plan():
        plan = lap._plan(_lib.F64 if itemsize == 8 else _lib.F32, (args.ny, args.nx), dev.index)
        if tuned:
            plan.set_tuning(args.rows_per_wave, args.xcd_remap, args.multi or 8, args.strip, args.prefetch)
        plan.set_timing(False)
        return plan
    plan = make_plan()
    d_in = [torch.from_numpy(f).to(dev) for f in wl["fields"]]
    run = (lambda: flt.apply_to_vector(d_in[0], d_in[1])) if len(d_in) == 2 else (lambda: (flt.apply(d_in[0]),))
    outs = None
    t_w = time.perf_counter()
    for _ in range(warmup):
        outs = run()
    torch.cuda.synchronize()
    while time.perf_counter() - t_w < 0.05:   # (a GPU that idled while the host folded the plan needs tens of ms to clock up again)
        outs = run()
        torch.cuda.synchronize()
    # ... and one untimed burst as long as a timed block, enqueued the same way (no synchronisation in between): the FIRST such burst of
    # a process stalls once for 30-60 ms inside the runtime (seen as one timed block 2-5 x slow on tripolar plans, whose launches fork /
    # join two queues; it never comes back) -- a one-off of the process, not a rate
    for _ in range(max(1, min(40, -(-steps // max(1, min(5, steps)))))):
        outs = run()
    torch.cuda.synchronize()
    plan.last_kernel()  # reset
    torch.cuda.synchronize()
    # The timed region: EXACTLY K applications, enqueued back to back with no host synchronisation between the applications of a
    # block (event timing is OFF: reading an event back after every application would idle the GPU while the host prepares the
    # next one).  The K applications are timed as up to 5 blocks, each bracketed by a synchronisation, because a plan runs its
    # blocked launches in one of two modes ~10 % apart depending on where its planes landed in HBM: half-way
    # through the plan is destroyed and folded again (untimed), so the blocks sample more than one placement.  `value` is the
    # median block; value_min / value_max / value_mean (all K applications over the summed block time) carry the spread.
    nblocks = max(1, min(5, steps))
    per_block = [steps // nblocks + (1 if b < steps % nblocks else 0) for b in range(nblocks)]
    block_s, replans = [], 0
    for b, nb_ in enumerate(per_block):
        if b == (nblocks + 1) // 2 and nblocks >= 2 and not args.no_replan:
            outs = None
            from gcm_filters_amd.kernels import clear_plan_cache
            clear_plan_cache()
            keep_away = torch.empty(48 << 20, dtype=torch.uint8, device=dev)   # nudge the allocator: the new planes land elsewhere
            plan = make_plan()
            t_w = time.perf_counter()
            outs = run()          # untimed: first application on the new plan (lazy state buffers) ...
            torch.cuda.synchronize()
            while time.perf_counter() - t_w < 0.05:   # ... and 50 ms of them: the GPU idled while the host folded the plan and takes tens of
                outs = run()                          # milliseconds to clock up again (one block in five came out 2-3 x slow without this)
                torch.cuda.synchronize()
            for _ in range(min(40, nb_)):             # ... and the one-off stall of a new plan's first unsynchronised burst (see above): 36 ms
                outs = run()                          # in the block after the re-plan of one full run (55 ms against 19 ms for 20 applications)
            torch.cuda.synchronize()
            del keep_away
            replans += 1
        t0 = time.perf_counter()
        for _ in range(nb_):
            outs = run()
        torch.cuda.synchronize()
        block_s.append(time.perf_counter() - t0)
    elapsed = sum(block_s)
    plan.last_kernel()  # reset
    outs = run()   # (untimed) names the dominant kernel of the plan now in use
    torch.cuda.synchronize()
    # kernel-level timing in a second, untimed pass: an event pair around the whole recurrence and one around every
    # launch of the dominant kernel, recorded on the stream the kernels run on
    plan.set_timing(2)
    dom_ms, dom_n, dom_min, dom_max = 0.0, 0, 1e30, 0.0
    kernel_ms, launches = 0.0, 0
    dom_reps = max(1, min(steps, 3))
    for _ in range(dom_reps):
        run()
        ms, nl = plan.last_timing()
        kernel_ms, launches = kernel_ms + ms, launches + nl
        ms, nl, lo, hi = plan.last_kernel_timing()
        dom_ms, dom_n, dom_min, dom_max = dom_ms + ms, dom_n + nl, min(dom_min, lo), max(dom_max, hi)
    plan.set_timing(False)
    return dict(dom_ms=dom_ms, dom_n=dom_n, dom_min=dom_min, dom_max=dom_max, dom_reps=dom_reps, wl=wl, grid=grid, fk=fk, itemsize=itemsize, nbatch=nbatch, n_steps=n_steps, elapsed=elapsed,
                kernel_ms=kernel_ms, launches=launches, outs=list(outs), kernel=plan.last_kernel(), geometry=plan.last_kernel_geometry(),
                flt=flt, d_in=d_in, cells=args.ny * args.nx * nbatch, block_s=block_s, per_block=per_block, replans=replans)


def spread_of(r):
    """Per-block rates of the timed region (run_single): median (= `value`), min, max, mean."""
    unit = r["cells"] * r["n_steps"]
    rates = sorted(unit * n / t for n, t in zip(r["per_block"], r["block_s"]))
    per_app = sorted(t / n for n, t in zip(r["per_block"], r["block_s"]))
    med = lambda v: v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return {"value": med(rates), "value_min": rates[0], "value_max": rates[-1], "value_mean": unit * sum(r["per_block"]) / sum(r["block_s"]),
            "ms_per_step": 1e3 * med(per_app), "blocks": len(rates), "applications_per_block": r["per_block"],
            "block_ms": [1e3 * t for t in r["block_s"]], "plans_refolded_between_blocks": r["replans"]}


_COPY_GBS = []


def device_copy_gbs():
    """Rate of a plain 600 MB device-to-device copy on THIS box (read + written bytes / time, HIP events; measured once per
    process, outside every timed region): what the memory system gives the simplest stream there is.  On the gpurun pool
    this is 5.1-5.3 TB/s, not the 6.3 TB/s of the guide -- the figure `hbm_frac` should be read against."""
    if not _COPY_GBS:
        import torch
        x = torch.empty(75_000_000, dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        y.copy_(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        _COPY_GBS.append(2 * x.numel() * 8 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del x, y
        torch.cuda.empty_cache()
    return _COPY_GBS[0]


def roofline_of(cfg, r, steps, default_tuning):
    w, nb, grid = r["itemsize"], r["nbatch"], r["grid"]
    if not (r["dom_n"] and r["dom_ms"] > 0):
        return None
    balg = b_alg(grid, w, 8, nb)
    avg_ms = r["dom_ms"] / r["dom_n"]                       # HIP events around each launch of the dominant kernel
    # recurrence steps one launch of that kernel advances: its last (vector kernels: fourth) template argument
    targs = r["kernel"][r["kernel"].index("<") + 1: r["kernel"].rindex(">")].split(", ")
    backward = any(k in r["kernel"] for k in ("k_ringc<", "k_ringcs<", "k_cgrid_stream2c<", "k_cgrid_ring<"))
    steps_per_launch = float(targs[1] if any(k in r["kernel"] for k in ("k_ringcs<", "k_cgrid_ring<", "k_cgrid_ringf<")) else   # (<T, S, ...>: the early-exit form of short strips, the static-ring C-grid kernel)
                             targs[2] if backward else
                             targs[3] if any(k in r["kernel"] for k in ("stream2", "k_scalar_multi", "k_ring")) else
                             (targs[-1] if "k_flux_multi2" in r["kernel"] else 1))
    one_pass_per_step = balg * r["cells"] * steps_per_launch / (avg_ms * 1e-3) / 1e9
    minb = min_bytes_per_cell_launch(grid, w, 8, nb, backward=backward) * r["cells"]
    rec, src = load_traffic(cfg, r["kernel"], r.get("geometry")) if default_tuning else (None, "non-default tuning: traffic withheld")
    # `achieved` / `frac` are PHYSICAL: the algorithmic bytes of ONE launch = every operand plane of SURVEY 8d's byte count read
    # once and every result plane written once (a launch is one pass over HBM however many Chebyshev steps it advances),
    # over the launch duration measured with HIP events.  0 < frac < 1 always; counter traffic / this figure = wasted re-reads.
    achieved = minb / (avg_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": rec.get("bytes_per_launch") if rec else None, "traffic_source": src,
           "kernel": r["kernel"], "geometry": r.get("geometry"), "avg_launch_ms": avg_ms, "min_launch_ms": r["dom_min"], "max_launch_ms": r["dom_max"],
           "launches_of_it_per_application": r["dom_n"] / r["dom_reps"], "steps_per_launch": steps_per_launch,
           "recurrence_ms_per_application": r["kernel_ms"] / r["dom_reps"],
           "alg_bytes_per_launch": minb, "alg_bytes_per_cell_launch": minb / r["cells"],
           # SURVEY 8d's streaming model prices every Chebyshev step with one pass over HBM; a blocked launch advances
           # steps_per_launch steps per pass, so this pair exceeds the peak by design: a speed-up over that model, NOT a hardware fraction
           "alg_one_pass_per_step_GBps": one_pass_per_step, "alg_one_pass_per_step_frac": one_pass_per_step / HBM_PEAK_GBS,
           "alg_bytes_per_cell_step": balg,
           "hbm_frac": None, "hbm_frac_of_copy_ceiling": None}
    if rec:
        gbs = rec["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
        out["hbm_frac"] = gbs / HBM_PEAK_GBS                      # counter bytes / this run's launch time / 8 TB/s
        out["hbm_frac_of_copy_ceiling"] = gbs / HBM_COPY_GBS
        try:  # the same against what a plain device copy reaches on this box, measured now
            cp = device_copy_gbs()
            out["device_copy_GBps_this_box"] = cp
            out["hbm_rate_over_device_copy"] = gbs / cp
        except Exception:
            pass
        out["traffic_over_alg_bytes"] = rec["bytes_per_launch"] / minb   # > 1: halo re-reads of the strip-marching scheme
        if rec.get("fetch_scale"):   # FETCH_SIZE calibrated on this kernel's own access pattern instead of the x2 rule for 16-byte-per-lane loads
            out["traffic_calibration"] = {"read_bytes_per_FETCH_SIZE_byte": rec["fetch_scale"], "note": rec.get("fetch_scale_note"),
                                          "traffic_by_the_x2_rule": rec.get("bytes_per_launch_x2_rule")}
        # SQ-counter view of the same kernel (profiles/): how the wave cycles split; `bound` stays the contract's enum
        for k in ("bound", "valu_active_frac", "salu_active_frac", "wait_memory_frac", "wait_issue_frac", "valu_arith_share",
                  "counters_source"):
            if k in rec:
                out["bound_detail" if k == "bound" else k] = rec[k]
    return out


def run_config1(dev):
    """BASELINE config 1 (REGULAR 512x512 fp64, Gaussian filter_scale 4, n_steps 16): the reference's CPU-runnable case.  On the
    GPU it is a latency measurement (two launches); output checked against the probe the imported reference produced
    (tests/golden/reference_generated.npz: REGULAR/config1)."""
    import torch

    from gcm_filters_amd import Filter, GridType, testing as T

    f = T.random_field((512, 512), 100)
    flt = Filter(filter_scale=4.0, dx_min=1.0, n_steps=16, grid_type=GridType.REGULAR)
    d = torch.from_numpy(f).to(dev)
    for _ in range(5):
        out = flt.apply(d)
    torch.cuda.synchronize()
    reps, blocks = 200, []
    for _ in range(5):   # five blocks, the median one is reported (a 30-us measurement is easily hit by a one-off stall)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = flt.apply(d)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    el = sorted(blocks)[2]
    rec = {"config": "BASELINE config 1: REGULAR 512x512, Gaussian filter_scale=4, n_steps=16 (HBM-resident, back-to-back calls)",
           "n_steps": 16, "steps": reps, "value": 512 * 512 * 16 * reps / el, "unit": "cell-steps/s",
           "us_per_application": 1e6 * el / reps, "us_per_application_min_max": [1e6 * min(blocks) / reps, 1e6 * max(blocks) / reps],
           "dtype": "f64"}
    gp = os.path.join(REPO, "tests", "golden", "reference_generated.npz")
    if os.path.exists(gp):
        with np.load(gp) as z:
            probe, sums = z["REGULAR/config1/probe"], z["REGULAR/config1/sums"]
        got = out.cpu().numpy()
        rec["parity"] = {"fixture": "tests/golden/reference_generated.npz:REGULAR/config1", "n_probes": int(probe.size),
                         "rel_err": float(np.abs(got[::8, ::8] - probe).max() / np.abs(probe).max()),
                         "sum_rel_err": float(abs(got.sum() - sums[0]) / abs(sums[0])), "tolerance": 1e-6,
                         "source": "imported reference (tests/golden/make_golden.py)"}
    return rec


def run_small_onchip(dev, no_cpu):
    """Not a BASELINE config: a small flux-form grid (IRREGULAR_WITH_LAND 512x512, the headline's filter: Taper, filter_scale 16 dx_min,
    n_steps 63) -- the north star's "whole n_steps polynomial fused into a single launch": gcmf_apply runs whole grids of up to 400 k
    cells on the chip (csrc/gcmf_resident.hip).  Reports the on-chip launch, the strip-marching launches of the same
    filter (GCMF_RESIDENT=0: eight launches, same bits) and, unless --no-cpu, the oracle parity of the result."""
    import torch

    from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
    from gcm_filters_amd.kernels import ALL_KERNELS

    shape = (512, 512)
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, shape, dev.index)
    d = torch.from_numpy(f).to(dev)

    def timed():
        for _ in range(5):
            out = flt.apply(d)
        torch.cuda.synchronize()
        plan.last_kernel()
        blocks = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(100):
                out = flt.apply(d)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 100)
        return sorted(blocks)[2], out, plan.last_kernel(), plan.last_kernel_geometry()
    before = os.environ.get("GCMF_RESIDENT")
    try:
        t_chip, out, k_chip, geom = timed()
        os.environ["GCMF_RESIDENT"] = "0"
        t_strip, out2, k_strip, _ = timed()
    finally:
        if before is None:
            os.environ.pop("GCMF_RESIDENT", None)
        else:
            os.environ["GCMF_RESIDENT"] = before
    n = int(flt.n_steps)
    rec = {"config": f"extra: IRREGULAR_WITH_LAND 512x512 f64, Taper filter_scale=16 dx_min, n_steps={n} (small grid: whole polynomial on the chip)",
           "n_steps": n, "value": shape[0] * shape[1] * n / t_chip, "unit": "cell-steps/s", "us_per_application": 1e6 * t_chip,
           "kernel": k_chip, "geometry": geom, "launches_per_application": 1 if "k_resident" in k_chip else None,
           "strip_marching": {"us_per_application": 1e6 * t_strip, "kernel": k_strip, "same_bits": bool(torch.equal(out.nan_to_num(), out2.nan_to_num()))},
           "dtype": "f64"}
    if not no_cpu:
        from oracle import gcmf_oracle as O
        fs = flt.filter_spec
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), "IRREGULAR_WITH_LAND", f, gv)
        got = out.cpu().numpy()
        rec["parity"] = {"rel_err": float(np.abs(got - want).max() / np.abs(want).max()), "tolerance": 1e-6,
                         "checked_against": "oracle, same grid / field / whole polynomial"}
    return rec


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


def run_midsize(dev):
    """The reference's own tutorial size (1080 x 1440, /docs/examples/example_tripole_grid.ipynb: a 1/4-degree ocean) with the headline's
    grid type and filter (IRREGULAR_WITH_LAND f64, Taper, filter_scale 16 dx_min, n_steps 63): too big for the on-chip kernel, small enough
    that the strips of the marching launches are short.  Not a BASELINE config; reported in `summary`."""
    import torch

    from gcm_filters_amd import Filter, FilterShape, GridType, testing as T

    shape = (1080, 1440)
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    d = torch.from_numpy(f).to(dev)
    for _ in range(5):
        flt.apply(d)
    torch.cuda.synchronize()
    blocks = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            flt.apply(d)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / 50)
    t = sorted(blocks)[2]
    n = int(flt.n_steps)
    return {"config": f"extra: IRREGULAR_WITH_LAND 1080x1440 f64, Taper filter_scale=16 dx_min, n_steps={n} (the reference's tutorial size)",
            "n_steps": n, "value": shape[0] * shape[1] * n / t, "unit": "cell-steps/s", "us_per_application": 1e6 * t, "dtype": "f64"}


def build_summary(out):
    """The whole round in <= 1.5 KB, printed LAST in the line (the driver's record keeps the tail): per workload
    [G cell-steps/s, roofline frac (algorithmic bytes of one launch / launch time / 8 TB/s), counter traffic / algorithmic bytes, parity
    rel_err vs the imported reference's probes] -- None where a figure does not exist -- plus the host path's three numbers."""
    g = lambda v: None if v is None else round(v / 1e9, 1)
    r3 = lambda v: None if v is None else round(v, 3)
    e = lambda v: None if v is None else float(f"{v:.1e}")

    def row(rec, parity=None):
        rf = rec.get("roofline") or {}
        par = parity if parity is not None else (rec.get("parity") or {})
        err = par.get("rel_err", (par.get("reference_probes") or {}).get("rel_err"))
        return [g(rec.get("value")), r3(rf.get("frac")), r3(rf.get("traffic_over_alg_bytes")), e(err)]
    s = {"cols": ["G cell-steps/s", "frac", "traffic/alg", "parity rel_err"]}
    main_cfg = out["config"]["workload"].split(":")[0].split()[-1]
    s["cfg" + main_cfg] = row(out, out.get("parity"))
    for rec in out.get("extra_configs") or []:
        name = rec["config"]
        if name.startswith("BASELINE config"):
            key = "cfg" + name.split(":")[0].split()[-1]
        elif "512x512" in name:
            key = "onchip_512"
        elif "1080x1440" in name:
            key = "mid_1080x1440"
        else:
            key = name[:24]
        s[key] = row(rec)
        opt = rec.get("forward_reference_opt_in")
        if opt:
            s[key + "_ref_scheme"] = row(opt)
    cb = out.get("cpu_baseline") or {}
    if cb:
        s["cpu_1core_M"] = round(cb["value"] / 1e6, 1)
    hp = out.get("host_path")
    if hp:
        s["host_ms"] = {"h2d": r3(hp.get("h2d_ms_page_locked")), "d2h": r3(hp.get("d2h_ms_page_locked")),
                        "pipeline": r3(hp.get("row_block_pipeline_ms")), "batch8_per_field": r3(hp.get("batch_of_8_ms_per_field"))}
    return s


def free_gpu():
    import gc

    import torch

    from gcm_filters_amd.kernels import clear_plan_cache
    free0 = torch.cuda.mem_get_info()[0]
    clear_plan_cache()
    gc.collect()
    torch.cuda.empty_cache()
    # The driver scrubs freed device memory in the background with the copy engines: for about a second after 20 GB have been
    # freed an upload and a download take turns instead of sharing the link (host path: 2.8 instead of 1.7 ms per field) and
    # resident kernels run ~5 % slower (experiments/scripts/host_batch_bisect3.py, profiles/r05/vram_scrub_after_free.txt).
    # What is timed next must not overlap with the clean-up of what was timed before: wait it out (0.1 s per GB freed).
    freed_gb = max(0, torch.cuda.mem_get_info()[0] - free0) / 2**30
    if freed_gb > 0.5:
        time.sleep(min(4.0, 0.1 * freed_gb))


# ------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child torch.distributed.run.  Nothing in this
    process has touched a GPU yet (device_count() does not initialise HIP)."""
    import torch

    share = os.environ.get("GCMF_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} HIP device(s) are visible "
                         f"(set GCMF_BENCH_SHARE_GPU=1 to run all ranks on one GPU over gloo for testing)")
    from gcm_filters_amd.testing import free_port
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # Watchdog: this parent never touches a GPU, so it is the one place that may kill a hung multi-rank run (an unmatched RCCL recv
    # waits for ever).  The children run in their own process group; on time-out the whole group is killed and the exit code says so.
    limit = float(os.environ.get("GCMF_BENCH_TIMEOUT_S", "1500"))
    t_start = time.time()
    for attempt in range(3):
        # the rendezvous port: below the ephemeral range (free_port), and if the launcher still finds it taken (EADDRINUSE: somebody else
        # bound it between our check and its listen) the run is started again on another port -- seen once in ~50 runs with OS-chosen ports
        port = free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        child = subprocess.Popen(cmd, env=env, start_new_session=True, stderr=subprocess.PIPE, text=True)
        seen = {"inuse": False}

        def pump(pipe=child.stderr, seen=seen):
            for line in pipe:
                if "EADDRINUSE" in line or "address already in use" in line.lower():
                    seen["inuse"] = True
                sys.stderr.write(line)
            pipe.close()
        import threading
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        try:
            rc = child.wait(timeout=max(limit - (time.time() - t_start), 0.001))
        except subprocess.TimeoutExpired:
            print(f"bench.py: the {args.gpus}-rank run did not finish within {limit:.0f} s (GCMF_BENCH_TIMEOUT_S): killing its process group",
                  file=sys.stderr)
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.wait()
            raise SystemExit(124)
        except KeyboardInterrupt:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            raise
        th.join(5)
        if rc != 0 and seen["inuse"] and attempt < 2:
            print(f"bench.py: rendezvous port {port} was taken (EADDRINUSE): starting the {args.gpus}-rank run again on another port", file=sys.stderr)
            continue
        raise SystemExit(rc)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (2..5), default 3")
    ap.add_argument("--ny", type=int, default=2400)
    ap.add_argument("--nx", type=int, default=3600)
    ap.add_argument("--nlev", type=int, default=0, help="vertical levels of config 5 (default 50) / config 6 (default 1)")
    ap.add_argument("--f32", action="store_true", help="config 6 only: f32 state instead of f64")
    ap.add_argument("--f64", action="store_true", help="config 5 only: f64 state instead of the BASELINE's f32")
    ap.add_argument("--filter-scale", type=float, default=0.0, help="filter scale in dx_min units (0 = the config's own)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1, scalar configs: strong = one (ny, nx) grid cut into N row slabs (reported as value); "
                         "weak = every GPU owns ny rows of an (N*ny, nx) grid")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the second (weak-scaling) measurement")
    ap.add_argument("--batch-levels", type=int, default=16,
                    help="N>1, scalar configs: time levels of the third measurement -- the same grid cut N ways with a BATCH of fields "
                         "(strips get tall again on short slabs: the strong-scaling workload that can use 8 GPUs); 0 skips it")
    ap.add_argument("--halo", type=int, default=0, help="N>1: ghost rows per exchange (0 = auto)")
    ap.add_argument("--exchange", choices=["auto", "native", "torch", "p2p"], default="auto",
                    help="N>1 halo exchange: native = RCCL send / recv issued by libgcmf on a side stream; torch = torch.distributed P2P; "
                         "p2p = peer stores into IPC-mapped mailboxes + flags on the compute stream (csrc/gcmf_p2p.hip, one node)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and the oracle parity check)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs measurements (configs 2, 4, 5)")
    ap.add_argument("--no-replan", action="store_true", help="keep ONE plan for all timed blocks (profiling runs: one placement)")
    ap.add_argument("--kinds", action="store_true", help="instead of the BASELINE line: one roofline record per Laplacian kind that is not a "
                    "BASELINE config (B-grid, MOM5U/T, the area-weighted and tripolar regular kinds) at --ny x --nx (tools/bench_kinds.py)")
    ap.add_argument("--kinds-only", default="", help="--kinds: comma-separated subset of the kind names")
    ap.add_argument("--cpu-steps", type=int, default=64, help="Laplacian steps of the CPU sample")
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--xcd-remap", type=int, default=-1)
    ap.add_argument("--multi", type=int, default=0, help="recurrence steps fused per HBM pass (0 = library default, 1 = off)")
    ap.add_argument("--strip", type=int, default=0, help="rows per wave strip of the temporally blocked kernel (0 = auto)")
    ap.add_argument("--prefetch", type=int, default=0, help="operand rows in flight per wave (0 = default)")
    return ap.parse_args()


def workload_name(cfg, r, args, extra=""):
    nb = r["nbatch"]
    return (f"{'BASELINE config' if (cfg <= 5 and not args.f64) else 'extra config'} {cfg}"
            f"{' (f64 variant)' if (args.f64 and cfg == 5) else ''}: {r['grid']} {args.ny}x{args.nx}"
            + (f" x{nb} levels" if nb > 1 else "") + extra)


def main_single(args):
    import torch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tuned = bool(args.rows_per_wave or args.xcd_remap >= 0 or args.multi or args.strip or args.prefetch)
    # north-star gate (fp64).  f32 state: SURVEY 8d's gate is 1e-4; held to 1e-5 here -- config 5 (n_steps 44) measures 1.1e-6 against
    # the reference's probes (the reference's own f32 path is 3e-6 off f64 arithmetic, the default here 1.8e-6)
    tol = lambda itemsize: 1e-6 if itemsize == 8 else 1e-5
    r = run_single(args.config, args, dev, args.steps, args.warmup, scale=args.filter_scale, tuned=tuned)
    sp = spread_of(r)
    fk = r["fk"]
    out = {
        "metric": "grid-cells*Laplacian-steps/sec", "value": sp["value"], "unit": "cell-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": sp["ms_per_step"],
        # the K timed applications ran as `blocks` synchronised blocks; value = the median block, the plan was folded anew between two
        # of them (a plan's placement in HBM decides between two launch-time modes ~10 % apart)
        "value_min": sp["value_min"], "value_max": sp["value_max"], "value_mean": sp["value_mean"],
        "timing": {k: sp[k] for k in ("blocks", "applications_per_block", "block_ms", "plans_refolded_between_blocks")},
        # N = 1 is the first point of the strong-scaling curve `--gpus N` reports (same global grid at every N)
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64" if r["itemsize"] == 8 else "f32", "data": "synthetic",
        "config": {"workload": workload_name(args.config, r, args),
                   "filter": f"{fk['filter_shape']} filter_scale={fk['filter_scale']:.6g} dx_min={fk['dx_min']:.6g}",
                   "n_steps": r["n_steps"], "global_grid": [args.ny, args.nx], "parallelism": "single GPU"},
    }
    out["roofline"] = roofline_of(args.config, r, args.steps, not tuned)
    nbatch_main = r["nbatch"]
    failed = []
    # ---- parity of the TIMED workload's output -----------------------------------------------------------------
    parity = {"tolerance": tol(r["itemsize"])}
    chk = golden_probe_check(args.config, args.filter_scale, (args.ny, args.nx), r["outs"])
    if chk is not None:
        parity["reference_probes"] = finish_probe_check(chk)
        if not parity["reference_probes"]["rel_err"] <= parity["tolerance"]:
            failed.append(f"reference probes: rel_err {parity['reference_probes']['rel_err']:.3e}")
    if not args.no_cpu:
        cpu, want, n_cpu = cpu_baseline(r["wl"], args.cpu_steps)
        out["cpu_baseline"] = cpu
        if n_cpu == r["n_steps"]:
            got = [o if o.ndim == 2 else o[0] for o in r["outs"]]
            what = "oracle, same grid / field / whole polynomial as the timed workload" + (" (level 0)" if r["nbatch"] > 1 else "")
        else:  # the CPU sample ran a truncated polynomial: one more (untimed) GPU application with the same one
            from gcm_filters_amd import filter as F
            spec = r["flt"].filter_spec
            short = F.FilterSpec(n_cpu, spec.s_max, np.asarray(spec.p)[: n_cpu + 1], spec.dx_min_sq)
            lev0 = [d if d.ndim == 2 else d[:1] for d in r["d_in"]]
            cls = r["flt"].Laplacian
            gargs = [r["wl"]["grid_vars"][k] for k in cls.required_grid_args()]
            if len(lev0) == 2:
                got = F._create_filter_func_vec(short, cls)(lev0[0], lev0[1], *gargs)
            else:
                got = (F._create_filter_func(short, cls)(lev0[0], *gargs),)
            got = [g.reshape(g.shape[-2:]) for g in got]
            what = f"oracle, same grid / field, polynomial truncated to n_steps={n_cpu} on both sides (level 0)"
        err, same = rel_err_and_nan([g.cpu().numpy() for g in got], want)
        parity.update({"rel_err": err, "nan_pattern_equal": same, "checked_against": what})
        if not (err <= parity["tolerance"] and same):
            failed.append(f"oracle: rel_err {err:.3e}, nan_pattern_equal {same}")
        del want
    else:
        out["cpu_baseline"] = None
    out["parity"] = parity
    # ---- the other BASELINE configs, briefly ---------------------------------------------------------------------
    if not args.no_extra:
        extras = [run_config1(dev)]
        if not extras[0].get("parity", {"rel_err": 0})["rel_err"] <= 1e-6:
            failed.append(f"config 1 reference probe: rel_err {extras[0]['parity']['rel_err']:.3e}")
        small = run_small_onchip(dev, args.no_cpu)
        if not small.get("parity", {"rel_err": 0})["rel_err"] <= 1e-6 or not small["strip_marching"]["same_bits"]:
            failed.append(f"small on-chip grid: parity {small.get('parity')} same bits as the strip-marching launches: {small['strip_marching']['same_bits']}")
        for cfg in (2, 4, 5):
            if cfg == args.config:
                continue
            r = None
            free_gpu()
            # an application of configs 2 / 4 takes ~1 ms: three of them after the idle seconds of the CPU leg are timed on a
            # GPU that has not clocked up again (measured 7 % low); config 5 takes 70 ms per application
            xs, xw = (5, 2) if cfg == 5 else (50, 5)
            r = run_single(cfg, args, dev, steps=xs, warmup=xw)
            xsp = spread_of(r)
            rec = {"config": workload_name(cfg, r, args), "n_steps": r["n_steps"], "steps": xs, "warmup": xw,
                   "value": xsp["value"], "value_min": xsp["value_min"], "value_max": xsp["value_max"], "unit": "cell-steps/s",
                   "ms_per_step": xsp["ms_per_step"], "dtype": "f64" if r["itemsize"] == 8 else "f32",
                   "roofline": roofline_of(cfg, r, xs, True)}
            chk = golden_probe_check(cfg, 0.0, (args.ny, args.nx), r["outs"])
            if chk is not None:
                rec["parity"] = dict(finish_probe_check(chk), tolerance=tol(r["itemsize"]))
                if not rec["parity"]["rel_err"] <= rec["parity"]["tolerance"]:
                    failed.append(f"config {cfg} reference probes: rel_err {rec['parity']['rel_err']:.3e}")
            if cfg == 5 and not args.no_cpu:
                rec["cpu_baseline_pool"] = cpu_baseline_pool(5, args.ny, args.nx, r["nbatch"], 4)
            if cfg == 5:
                # The default carries the whole polynomial in f32 (backward evaluation, k_cgrid_ring); the reference sums f32 fields in an f64
                # running sum (NumPy >= 2 promotion, reference filter.py:192-206).  Filter(evaluation="reference") runs exactly that scheme
                # (forward recurrence, f64 fbar): measured here so that the like-for-like figure is on record next to the default.
                r2 = None
                free_gpu()
                r2 = run_single(cfg, args, dev, steps=xs, warmup=xw, evaluation="reference")
                osp = spread_of(r2)
                opt = {"evaluation": "reference",   # forward recurrence, f32 T_k, f64 running sum: the reference's own scheme (NOT the default)
                       "kernel": r2["kernel"], "value": osp["value"], "value_min": osp["value_min"], "value_max": osp["value_max"], "unit": "cell-steps/s",
                       "ms_per_step": osp["ms_per_step"], "roofline": roofline_of(cfg, r2, xs, True)}
                chk2 = golden_probe_check(cfg, 0.0, (args.ny, args.nx), r2["outs"])
                if chk2 is not None:
                    opt["parity"] = dict(finish_probe_check(chk2), tolerance=tol(r2["itemsize"]))
                    if not opt["parity"]["rel_err"] <= opt["parity"]["tolerance"]:
                        failed.append(f"config {cfg} forward (reference) evaluation: rel_err {opt['parity']['rel_err']:.3e}")
                rec["forward_reference_opt_in"] = opt
                r2 = None
            if cfg == 2:
                # The default since round 4 evaluates the land-mask (and REGULAR) types backwards too (k_ringc: fused multiply-adds,
                # one plane less, <= 1e-14 from numpy).  Filter(evaluation="reference") is the escape that stays bit-exact with
                # numpy (the reference's forward recurrence, k_ring): measured here so that the price of bit-exactness is on record.
                r2 = None
                free_gpu()
                r2 = run_single(cfg, args, dev, steps=xs, warmup=xw, evaluation="reference")
                osp = spread_of(r2)
                opt = {"evaluation": "reference", "kernel": r2["kernel"],
                       "value": osp["value"], "value_min": osp["value_min"], "value_max": osp["value_max"], "unit": "cell-steps/s",
                       "ms_per_step": osp["ms_per_step"]}
                chk2 = golden_probe_check(cfg, 0.0, (args.ny, args.nx), r2["outs"])
                if chk2 is not None:
                    opt["parity"] = dict(finish_probe_check(chk2), tolerance=tol(r2["itemsize"]))
                    if not opt["parity"]["rel_err"] <= opt["parity"]["tolerance"]:
                        failed.append(f"config {cfg} forward (reference) evaluation: rel_err {opt['parity']['rel_err']:.3e}")
                rec["forward_reference_opt_in"] = opt
            extras.append(rec)
        extras.append(small)
        extras.append(run_midsize(dev))
        out["extra_configs"] = extras
        if args.config == 3 and (args.ny, args.nx) == (2400, 3600):
            r = None
            free_gpu()
            out["host_path"] = run_host_path(dev, args)
    if args.config == 5 and not args.no_cpu:
        out["cpu_baseline_pool"] = cpu_baseline_pool(5, args.ny, args.nx, nbatch_main, 4)
    out["summary"] = build_summary(out)   # LAST key: the tail of the line carries every config's value / frac / traffic / parity
    print(json.dumps(out))
    if failed:
        print("bench.py: PARITY FAILURE -- " + "; ".join(failed), file=sys.stderr)
        return 1
    return 0


def main_multi(args, world, rank, local_rank):
    import torch
    import torch.distributed as dist

    from gcm_filters_amd import testing as T

    # Under a launcher (torch.distributed.run) there is no parent of ours to watch the run: every rank carries its own dead-man timer.  A
    # rank that sits in an unmatched collective for GCMF_BENCH_TIMEOUT_S says so and leaves with exit code 124 (os._exit works from a
    # timer thread while the main thread is stuck in a HIP / RCCL call; the launcher then tears the other ranks down) -- a hang never
    # lasts until the driver's own limit.
    import threading
    limit = float(os.environ.get("GCMF_BENCH_TIMEOUT_S", "1500"))

    def _dead_man():
        print(f"bench.py: rank {rank} did not finish within {limit:.0f} s (GCMF_BENCH_TIMEOUT_S): leaving with exit code 124", file=sys.stderr, flush=True)
        os._exit(124)
    dead_man = threading.Timer(limit, _dead_man)
    dead_man.daemon = True
    dead_man.start()
    share_gpu = os.environ.get("GCMF_BENCH_SHARE_GPU") == "1"  # test hook: all ranks on cuda:0 over gloo
    if share_gpu:
        local_rank = 0
    if not share_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP devices visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if share_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    cpu_dev = "cpu" if share_gpu else dev

    spread = {}
    calls = {"warmup": 0, "timed": 0}   # applications of the collective workload this rank has run (checked equal across ranks)

    def agree_max(x):
        """MAX over ranks of a host-side number: the only way a rank may turn its own clock into a decision about a collective."""
        tt = torch.tensor([float(x)], device=cpu_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(fn, barrier=True):
        """W warm-up applications (and at least 50 ms of them: a GPU that idled while the host folded plans takes tens of milliseconds
        to clock up again -- a one-off 35 ms stall was seen in exactly this spot), then EXACTLY K applications timed as up to five
        blocks, each bracketed by barrier + synchronize on both sides, MAX over ranks per block.  Returns K x the median block's time
        per application (what `value` is computed from); min / max of the blocks go to `spread`.

        `fn` is a COLLECTIVE (halo exchanges): every rank must call it the same number of times.  Round 3 let every rank extend its
        warm-up by its own clock; a rank that started a little later ran one application more, its exchanges met a neighbour sitting
        in the barrier (p2p: a chain of 2 s time-outs and stale ghost rows; RCCL: an unmatched recv = a hang).  The number of extra
        warm-up applications is now derived from the all-reduced MAX of the elapsed time, i.e. identical on every rank, and bounded."""
        skew_ms = float(os.environ.get("GCMF_BENCH_SKEW_MS", "0") or 0)   # test hook: rank 1 arrives late (tests/test_gpu_bench_cli.py)
        if skew_ms and rank == 1:
            time.sleep(skew_ms * 1e-3)
        t_w = time.perf_counter()
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        calls["warmup"] += args.warmup
        spent = agree_max(time.perf_counter() - t_w)
        per_app_est = spent / max(args.warmup, 1)
        extra = 0 if spent >= 0.05 else int(min(200, math.ceil((0.05 - spent) / max(per_app_est, 1e-4))))
        for _ in range(extra):        # the same count on every rank (derived from an all-reduced figure)
            fn()
        torch.cuda.synchronize()
        calls["warmup"] += extra
        nblocks = max(1, min(5, args.steps))
        per_block = [args.steps // nblocks + (1 if b < args.steps % nblocks else 0) for b in range(nblocks)]
        per_app = []
        for nb_ in per_block:
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nb_):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            el = time.perf_counter() - t0
            calls["timed"] += nb_
            per_app.append(agree_max(el) / nb_)
        srt = sorted(per_app)
        med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        spread["last"] = {"blocks": nblocks, "applications_per_block": per_block, "ms_per_application_min": 1e3 * srt[0],
                          "ms_per_application_max": 1e3 * srt[-1], "ms_per_application_median": 1e3 * med}
        return med * args.steps

    failed = []
    cfg = args.config
    if cfg in (5, 6):
        # ---- levels over GPUs: the reference's own (dask) parallelism, zero communication -------------
        from gcm_filters_amd import Filter, FilterShape, GridType
        nlev = args.nlev or (50 if cfg == 5 else 8)
        lo, hi = (rank * nlev) // world, ((rank + 1) * nlev) // world
        wl = T.baseline_workload(cfg, (args.ny, args.nx), f32=args.f32, f64=args.f64, scale=args.filter_scale,
                                 levels=list(range(lo, hi)))
        grid, fk = wl["grid"], wl["fk"]
        flt = Filter(grid_type=GridType[grid], grid_vars=wl["grid_vars"], filter_scale=fk["filter_scale"],
                     dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]])
        n_steps = int(flt.n_steps)
        d_in = [torch.from_numpy(f).to(dev) for f in wl["fields"]]
        keep = {}

        def one():
            keep["o"] = flt.apply_to_vector(d_in[0], d_in[1]) if hi > lo else None
        elapsed = timed(one)
        main_spread = dict(spread["last"])
        itemsize = wl["fields"][0].dtype.itemsize
        cells = nlev * args.ny * args.nx
        scaling, par = "strong", f"levels x{world} ({nlev} levels in all, {hi - lo} on rank 0; no communication)"
        chk = golden_probe_check(cfg, args.filter_scale, (args.ny, args.nx), keep["o"]) if (rank == 0 and hi > lo) else None
        parity = dict(finish_probe_check(chk), tolerance=1e-6 if itemsize == 8 else 1e-4) if chk else None
        weak = None
        ny_global = args.ny
        kernel_ms = launches = 0
        kernel_apps, backward_slabs = 1, False
    else:
        from gcm_filters_amd.distributed import SlabFilter

        def build(ny_global, scaling):
            wl = T.baseline_workload(cfg, (ny_global, args.nx), scale=args.filter_scale)
            fk = dict(wl["fk"])
            sf = SlabFilter(wl["grid"], wl["grid_vars"], fk, ny_global, args.nx, halo=args.halo or None,
                            dtype=wl["fields"][0].dtype, device=local_rank, exchange=args.exchange)
            if args.multi:
                sf.multi_depth = args.multi
            sf.time_kernels = True
            return wl, sf, sf.scatter_from_global(wl["fields"])

        def measure(ny_global, scaling):
            wl, sf, local = build(ny_global, scaling)
            keep = {}

            def one():
                keep["o"] = sf.apply_local(local)
            for _ in range(args.warmup):
                one()
            sf.collect_kernel_times()
            sf.kernel_ms, sf.kernel_launches, sf.kernel_apps = 0.0, 0, 0
            elapsed = timed(one)
            sf.collect_kernel_times()
            return wl, sf, keep["o"], elapsed

        ny_global = args.ny * world if args.scaling == "weak" else args.ny
        wl, sf, outs, elapsed = measure(ny_global, args.scaling)
        main_spread = dict(spread["last"])
        grid, fk, n_steps = wl["grid"], wl["fk"], sf.n_steps
        itemsize = wl["fields"][0].dtype.itemsize
        cells = ny_global * args.nx
        scaling = args.scaling
        par = f"row-slabs x{world}, halo {sf.halo} rows exchanged every {sf.halo} steps ({sf.exchange_kind})"
        kernel_ms, launches, kernel_apps = sf.kernel_ms, sf.kernel_launches, max(sf.kernel_apps, 1)
        backward_slabs = bool(sf.backward_cut)
        # parity of the timed (strong) workload against the reference's probes: every rank checks the probes it owns
        parity = None
        chk = golden_probe_check(cfg, args.filter_scale, (ny_global, args.nx), outs, sf.row_begin, sf.row_end)
        if chk is not None:
            g = torch.from_numpy(np.nan_to_num(chk["got"])).to(cpu_dev)
            m = torch.from_numpy(np.broadcast_to(chk["mine"], chk["got"].shape).astype(np.float64)).to(cpu_dev)
            dist.all_reduce(g)
            dist.all_reduce(m)
            assert bool((m == 1).all()), "every probe must be owned by exactly one rank"
            chk["got"], chk["mine"] = g.cpu().numpy(), np.ones(chk["mine"].shape, dtype=bool)
            parity = dict(finish_probe_check(chk), tolerance=1e-6 if itemsize == 8 else 1e-4)
        # what the halo exchanges cost this run: the same slabs once more with the exchange stubbed out (ghost rows go stale, the
        # launches and their row ranges are the real ones) -- host + device cost per exchange = the difference / exchanges
        sf.exchanges = 0
        keep_o = {}
        def one_real():
            keep_o["o"] = sf.apply_local(local_main)
        local_main = sf.scatter_from_global(wl["fields"])
        one_real()
        ex_per_app = sf.exchanges
        real_start, real_finish, real_driver = sf._exchange_start, sf._exchange_finish, sf.native_driver
        sf._exchange_start, sf._exchange_finish, sf.native_driver = (lambda tensors: None), (lambda ticket: None), False
        el_stub = timed(one_real)
        sf._exchange_start, sf._exchange_finish, sf.native_driver = real_start, real_finish, real_driver
        exchange_rec = {"kind": sf.exchange_kind, "halo_rows": sf.halo, "exchanges_per_application": ex_per_app,
                        "ms_per_application_without_exchange": 1e3 * el_stub / args.steps,
                        "us_per_exchange_host_and_device": (1e6 * (elapsed - el_stub) / args.steps / ex_per_app) if ex_per_app else None,
                        "note": "max over ranks of the timed region with the exchange stubbed out, subtracted from the real run"}
        one_real = None
        keep_o.clear()
        torch.cuda.synchronize()
        if sf.p2p_timed_out():    # a wait inside the p2p exchange kernels failed: the numbers above mean nothing (results are NaN)
            failed.append(f"rank {rank}: a p2p halo exchange failed (time-out or a neighbour's abort)")
        # every rank must have run the same number of (collective) applications and exchanges
        mine = [calls["warmup"], calls["timed"], sf.exchanges, sf.p2p.seq() if sf.p2p is not None else -1]
        every = [None] * world
        dist.all_gather_object(every, mine)
        matched = all(e == every[0] for e in every)
        if not matched:
            failed.append(f"ranks ran different numbers of collective calls [warm-up, timed, exchanges, p2p seq]: {every}")
        exchange_rec.update({"collective_calls_rank0": {"warmup": mine[0], "timed": mine[1], "exchanges": mine[2], "p2p_seq": mine[3]},
                             "matched_across_ranks": matched, "backend": dist.get_backend(),
                             "rccl": sf.comm.describe() if getattr(sf, "comm", None) is not None else None})
        weak = None
        batched = None
        if args.scaling == "strong" and args.batch_levels > 1:
            # Third figure: the SAME grid cut N ways, a batch of time levels per application (the reference's own leading dims).  One field
            # leaves an 8-way slab 11-row strips that march 11 + 2 S rows (bound ~3.6 x); a batch makes the strips tall again.
            nb = args.batch_levels
            fb = [np.stack([f + 0.01 * k for k in range(nb)]) for f in wl["fields"]]
            local_b = sf.scatter_from_global(fb)
            keep_b = {}

            def one_b():
                keep_b["o"] = sf.apply_local(local_b)
            one_b()
            el_b = timed(one_b)
            batched = {"levels": nb, "value": nb * cells * n_steps * args.steps / el_b, "unit": "cell-steps/s",
                       "ms_per_step": 1e3 * el_b / args.steps, "timing": dict(spread["last"]), "scaling": "strong",
                       "note": f"third figure: the same {ny_global}x{args.nx} grid cut {world} ways, {nb} time levels per application "
                               "(a batch of fields through the slab path, same exchanges per application as the single field)"}
            # its level 0 is the timed single field: the slab path must give the same bits for it inside the batch
            same0 = bool(torch.equal(torch.nan_to_num(keep_b["o"][0][0]), torch.nan_to_num(outs[0].reshape(keep_b["o"][0][0].shape))))
            batched["level0_same_bits_as_the_single_field"] = same0
            if not same0:
                failed.append(f"rank {rank}: level 0 of the batched slab run differs from the single-field run")
            del local_b, fb
            keep_b.clear()
        if args.scaling == "strong" and not args.no_weak:
            del sf, outs, local_main
            free_gpu()
            wl2, sf2, _, el2 = measure(args.ny * world, "weak")
            weak = {"value": args.ny * world * args.nx * sf2.n_steps * args.steps / el2, "unit": "cell-steps/s",
                    "ms_per_step": 1e3 * el2 / args.steps, "global_grid": [args.ny * world, args.nx],
                    "note": "second figure: every GPU owns a full BASELINE-size slab of an (N*ny, nx) grid"}
    if parity is not None and not parity["rel_err"] <= parity["tolerance"]:
        failed.append(f"reference probes: rel_err {parity['rel_err']:.3e}")
    if rank == 0:
        out = {
            "metric": "grid-cells*Laplacian-steps/sec", "value": cells * n_steps * args.steps / elapsed,
            "unit": "cell-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64" if itemsize == 8 else "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config {cfg}: {grid} {ny_global}x{args.nx}"
                                   + (f" x{args.nlev or 50} levels" if cfg == 5 else ""),
                       "filter": f"{fk['filter_shape']} filter_scale={fk['filter_scale']:.6g} dx_min={fk['dx_min']:.6g}",
                       "n_steps": n_steps, "global_grid": [ny_global, args.nx], "parallelism": par},
            "timing": main_spread, "value_min": cells * n_steps / (main_spread["ms_per_application_max"] * 1e-3),
            "value_max": cells * n_steps / (main_spread["ms_per_application_min"] * 1e-3),
            "parity": parity, "weak": weak, "batched_strong": batched if cfg not in (5, 6) else None,
            "exchange": exchange_rec if cfg not in (5, 6) else None, "cpu_baseline": None,
            # physical, like the N = 1 line: algorithmic bytes of ONE launch (every operand plane read once, every result written once)
            # x the launches of an application over the time between rank 0's first and last launch of it
            "roofline": None if not launches else (lambda per_launch, ms_app: {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "kernel_ms_per_step_rank0": ms_app,
                "launches_per_step_rank0": launches / kernel_apps,
                "achieved": per_launch * (launches / kernel_apps) / (ms_app * 1e-3) / 1e9,
                "frac": per_launch * (launches / kernel_apps) / (ms_app * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "alg_bytes_per_launch": per_launch,
                "traffic": None, "note": "per GPU, rank 0: algorithmic bytes of one launch on its slab x launches / time between the first and "
                                         "the last launch of an application (includes exchange waits); see the N=1 line for the kernel-level roofline"})(
                min_bytes_per_cell_launch(grid, itemsize, 8, 1, backward=backward_slabs) * (cells / world), kernel_ms / kernel_apps),
        }
        print(json.dumps(out))
        if failed:
            print("bench.py: PARITY FAILURE -- " + "; ".join(failed), file=sys.stderr)
    dist.barrier()
    dist.destroy_process_group()
    dead_man.cancel()
    return 1 if failed else 0


def main():
    args = parse()
    if args.nlev <= 0 and args.config == 6:
        args.nlev = 0
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        self_launch(args)  # does not return
    world = max(world, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    if args.kinds:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import bench_kinds
        return bench_kinds.main(args)
    if world == 1:
        return main_single(args)
    return main_multi(args, world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")))


if __name__ == "__main__":
    sys.exit(main())
