"""Shared pieces of bench.py (the BASELINE line), tools/bench_multi.py (N > 1), tools/bench_host.py (host buffers) and tools/bench_kinds.py:
byte counts, the CPU baseline (the oracle on host cores), parity helpers, the timed region of one workload (run_single) and its roofline."""
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

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    for _ in range(3):   # (round 6: THREE such bursts -- the stall belongs to the n-th hundred launches of the process, not to the first burst:
        #                   with one burst it landed in the second or third timed block of every full run, 47-80 ms instead of 18)
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
    backward = any(k in r["kernel"] for k in ("k_ringc<", "k_ringcs<", "k_ringcp<", "k_ringcz<", "k_cgrid_stream2c<", "k_cgrid_ring<"))
    steps_per_launch = float(targs[1] if any(k in r["kernel"] for k in ("k_ringcs<", "k_ringcz<", "k_cgrid_ring<", "k_cgrid_ringf<")) else   # (<T, S, ...>: the early-exit form of short strips, the static-ring C-grid kernel)
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




def workload_name(cfg, r, args, extra=""):
    nb = r["nbatch"]
    return (f"{'BASELINE config' if (cfg <= 5 and not args.f64) else 'extra config'} {cfg}"
            f"{' (f64 variant)' if (args.f64 and cfg == 5) else ''}: {r['grid']} {args.ny}x{args.nx}"
            + (f" x{nb} levels" if nb > 1 else "") + extra)
