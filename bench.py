#!/usr/bin/env python3
"""bench.py -- throughput of the fused iterated-Laplacian filter on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--no-cpu]

A "step" is ONE whole filter application (all n_steps Chebyshev/Laplacian steps) of the workload's field,
inputs already resident in HBM.  metric = grid-cells * Laplacian-steps / second (BASELINE.json).
Default workload = BASELINE config 3, the one the north-star target is quoted on:
IRREGULAR_WITH_LAND 2400x3600 fp64, Taper filter, filter_scale = 16 dx_min  =>  n_steps 63.

N > 1 (launched by torch.distributed.run, one rank per GPU): the grid is cut into row slabs along y with
halo rows exchanged over RCCL (gcm_filters_amd/distributed.py); see --scaling.

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# algorithmic HBM bytes per cell per Laplacian step (SURVEY 8d / DESIGN.md): 5 state words + folded coefficients
# w = sizeof(state), f = sizeof(fbar) (f64 even for f32 state: NumPy >= 2 promotion), L = levels sharing the 2-D planes
B_ALG = {
    "REGULAR": lambda w, f, L: 3 * w + 2 * f,
    "REGULAR_WITH_LAND": lambda w, f, L: 3 * w + 2 * f + 1.0 / L,
    "IRREGULAR_WITH_LAND": lambda w, f, L: 3 * w + 2 * f + 3 * w / L,
    "TRIPOLAR_POP_WITH_LAND": lambda w, f, L: 3 * w + 2 * f + 3 * w / L,
    "VECTOR_C_GRID": lambda w, f, L: 2 * (3 * w + 2 * f) + 14 * w / L,
    "VECTOR_B_GRID": lambda w, f, L: 2 * (3 * w + 2 * f) + 8 * w / L,
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def build_workload(cfg: int, ny: int, nx: int, nlev: int, f32: bool = False, f64: bool = False):
    """Synthetic inputs of BASELINE.json configs: returns dict(grid, fields, grid_vars, filter kwargs)."""
    from gcm_filters_amd import FilterShape, testing as T

    shape = (ny, nx)
    if cfg == 1:  # BASELINE config 1 scaled up is not asked for; REGULAR at the benchmark size for reference
        grid = "REGULAR"
        gv = {}
        fields = [T.random_field(shape, 100)]
        fk = dict(filter_scale=50.0, dx_min=1.0, filter_shape=FilterShape.GAUSSIAN)
    elif cfg == 2:
        grid = "REGULAR_WITH_LAND"
        gv = {"wet_mask": T.land_mask(shape)}
        fields = [T.random_field(shape, 100)]
        fk = dict(filter_scale=50.0, dx_min=1.0, filter_shape=FilterShape.GAUSSIAN)
    elif cfg == 3:
        grid = "IRREGULAR_WITH_LAND"
        gv = T.scalar_grid_vars(grid, shape)
        fields = [T.random_field(shape, 100)]
        dx = T.grid_dx_min(grid, gv)
        fk = dict(filter_scale=16 * dx, dx_min=dx, filter_shape=FilterShape.TAPER)
    elif cfg == 4:
        grid = "TRIPOLAR_POP_WITH_LAND"
        gv = T.scalar_grid_vars(grid, shape)
        fields = [T.random_field(shape, 100)]
        dx = T.grid_dx_min(grid, gv)
        fk = dict(filter_scale=50 * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN)
    elif cfg == 5:
        grid = "VECTOR_C_GRID"
        cdt = np.float64 if f64 else np.float32   # BASELINE config 5 is f32; --f64 is an extra measurement
        gv = {k: v.astype(cdt) for k, v in T.vector_grid_vars(grid, shape).items()}
        gv["kappa_aniso"] = np.zeros(shape, dtype=cdt)
        fields = [np.stack([T.random_field(shape, 42 + c + 2 * l).astype(cdt) for l in range(nlev)])
                  for c in range(2)]
        dx = T.grid_dx_min(grid, gv)
        fk = dict(filter_scale=40 * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN)
    elif cfg == 6:  # not a BASELINE config: the POP B-grid vector Laplacian at the benchmark size, fp64
        grid = "VECTOR_B_GRID"
        gv = T.vector_grid_vars(grid, shape)
        if nlev <= 1:
            fields = [T.random_field(shape, 42), T.random_field(shape, 43)]
        else:
            fields = [np.stack([T.random_field(shape, 42 + c + 2 * l) for l in range(nlev)]) for c in range(2)]
        if f32:
            gv = {k: v.astype(np.float32) for k, v in gv.items()}
            fields = [f.astype(np.float32) for f in fields]
        dx = T.grid_dx_min(grid, gv)
        fk = dict(filter_scale=40 * dx, dx_min=dx, filter_shape=FilterShape.GAUSSIAN)
    else:
        raise SystemExit(f"unknown --config {cfg}")
    return dict(grid=grid, fields=fields, grid_vars=gv, fk=fk)


def cpu_baseline(wl, budget_steps: int):
    """The reference's numpy path (oracle port), single thread like the reference runs one 2-D field,
    on a bounded sample: the SAME grid and field with the polynomial truncated to `budget_steps` steps."""
    from oracle import gcmf_oracle as O

    fk = wl["fk"]
    full = O.make_spec(fk["filter_scale"], fk["dx_min"], fk["filter_shape"].name)
    n = min(budget_steps, full.n_steps)
    spec = O.FilterSpec(n, full.s_max, full.p[: n + 1], full.dx_min_sq)
    fields = [f if f.ndim == 2 else f[0] for f in wl["fields"]]  # one level of a batched workload
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        if len(fields) == 2:
            O.filter_func_vec(spec, wl["grid"], fields[0], fields[1], wl["grid_vars"])
        else:
            O.filter_func(spec, wl["grid"], fields[0], wl["grid_vars"])
    dt = time.perf_counter() - t0
    ny, nx = fields[0].shape
    return {"value": ny * nx * n / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
            "sample": f"same {ny}x{nx} grid and field, 1 level, "
                      + ("whole polynomial" if n == full.n_steps else f"polynomial truncated to n_steps={n}")
                      + f" (n_steps={n}, {dt:.1f} s, numpy {np.__version__}, host has {os.cpu_count()} cores)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (2..5), default 3")
    ap.add_argument("--ny", type=int, default=2400)
    ap.add_argument("--nx", type=int, default=3600)
    ap.add_argument("--nlev", type=int, default=0, help="vertical levels of config 5 (default 50) / config 6 (default 1)")
    ap.add_argument("--f32", action="store_true", help="config 6 only: f32 state instead of f64")
    ap.add_argument("--f64", action="store_true", help="config 5 only: f64 state instead of the BASELINE's f32")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = every GPU owns a full ny-row slab of a (N*ny, nx) grid; strong = one (ny, nx) grid")
    ap.add_argument("--halo", type=int, default=0, help="N>1: ghost rows per exchange (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-steps", type=int, default=64, help="Laplacian steps of the CPU sample")
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--xcd-remap", type=int, default=-1)
    ap.add_argument("--multi", type=int, default=0, help="recurrence steps fused per HBM pass (0 = library default, 1 = off)")
    ap.add_argument("--strip", type=int, default=0, help="rows per wave strip of the temporally blocked kernel (0 = auto)")
    ap.add_argument("--prefetch", type=int, default=0, help="operand rows in flight per wave (0 = default)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    # test hook: GCMF_BENCH_SHARE_GPU=1 runs all ranks on cuda:0 over gloo (a gpurun box has one GPU; RCCL refuses
    # two ranks on one device).  The driver's real multi-GPU runs use one GPU per rank over RCCL.
    share_gpu = os.environ.get("GCMF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gcm_filters_amd import Filter, GridType
    from gcm_filters_amd.kernels import ALL_KERNELS

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    ny_global = args.ny * world if (world > 1 and args.scaling == "weak") else args.ny
    if args.nlev <= 0:
        args.nlev = 50 if args.config == 5 else 1
    wl = build_workload(args.config, ny_global if world > 1 else args.ny, args.nx, args.nlev, args.f32, args.f64)
    grid, fk = wl["grid"], wl["fk"]
    itemsize = wl["fields"][0].dtype.itemsize
    nbatch = 1 if wl["fields"][0].ndim == 2 else wl["fields"][0].shape[0]

    if world == 1:
        flt = Filter(grid_type=GridType[grid], grid_vars=wl["grid_vars"], **fk)
        n_steps = int(flt.n_steps)
        lap = ALL_KERNELS[GridType[grid]](*[wl["grid_vars"][k] for k in ALL_KERNELS[GridType[grid]].required_grid_args()])
        from gcm_filters_amd import _lib
        plan = lap._plan(_lib.F64 if itemsize == 8 else _lib.F32, (args.ny, args.nx), local_rank)
        if args.rows_per_wave or args.xcd_remap >= 0 or args.multi or args.strip or args.prefetch:
            plan.set_tuning(args.rows_per_wave, args.xcd_remap, args.multi or 8, args.strip, args.prefetch)
        plan.set_timing(True)
        d_in = [torch.from_numpy(f).to(dev) for f in wl["fields"]]
        run = (lambda: flt.apply_to_vector(d_in[0], d_in[1])) if len(d_in) == 2 else (lambda: flt.apply(d_in[0]))
        for _ in range(args.warmup):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms, launches = 0.0, 0
        for _ in range(args.steps):
            run()
            ms, nl = plan.last_timing()  # hipEvents on the stream the kernels ran on
            kernel_ms += ms
            launches += nl
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        cells = args.ny * args.nx * nbatch
    else:
        from gcm_filters_amd.distributed import SlabFilter
        sf = SlabFilter(grid, wl["grid_vars"], fk, ny_global, args.nx, halo=args.halo or None,
                        dtype=np.float64 if itemsize == 8 else np.float32, device=local_rank)
        if args.multi:
            sf.multi_depth = args.multi
        sf.time_kernels = True
        n_steps = sf.n_steps
        local = sf.scatter_from_global(wl["fields"])
        for _ in range(args.warmup):
            sf.apply_local(local)
        sf.collect_kernel_times()
        sf.kernel_ms, sf.kernel_launches = 0.0, 0
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sf.apply_local(local)
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        tt = torch.tensor([elapsed], device="cpu" if share_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        sf.collect_kernel_times()
        kernel_ms, launches = sf.kernel_ms, sf.kernel_launches
        cells = ny_global * args.nx * nbatch

    value = cells * n_steps * args.steps / elapsed
    out = {
        "metric": "grid-cells*Laplacian-steps/sec",
        "value": value,
        "unit": "cell-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f64" if itemsize == 8 else "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{'BASELINE config' if (args.config <= 5 and not args.f64) else 'extra config'} {args.config}{' (f64 variant)' if args.f64 else ''}: {grid} {args.ny}x{args.nx}"
                        + (f" x{nbatch} levels" if nbatch > 1 else "") + (f" per GPU, {world} row slabs" if world > 1 and args.scaling == "weak" else ""),
            "filter": f"{fk['filter_shape'].name} filter_scale={fk['filter_scale']:.6g} dx_min={fk['dx_min']:.6g}",
            "n_steps": n_steps,
            "global_grid": [ny_global if world > 1 else args.ny, args.nx],
            "parallelism": f"row-slabs x{world}" if world > 1 else "single GPU",
        },
    }
    if rank == 0:
        w = itemsize
        b_alg = B_ALG[grid](w, 8, nbatch)
        cells_per_launch = (cells // world) if world > 1 else cells
        if launches and kernel_ms > 0:
            # a launch of the temporally blocked kernel advances several steps: price every launch with the
            # cell-steps it processed (sum over launches = cells * n_steps per filter application)
            avg_ms = kernel_ms / launches
            steps_per_launch = n_steps * args.steps / launches
            achieved = b_alg * cells_per_launch * steps_per_launch / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/),
            # valid for the default tuning at N=1 only
            traffic, kname = None, {"VECTOR_C_GRID": "k_cgrid_stream", "VECTOR_B_GRID": "k_bgrid_stream"}.get(grid, "k_scalar_multi")
            tf = os.path.join(REPO, "profiles", "hbm_traffic.json")
            default_tuning = not (args.multi or args.strip or args.prefetch or args.rows_per_wave or args.xcd_remap >= 0)
            if os.path.exists(tf) and world == 1 and default_tuning:
                try:
                    rec = json.load(open(tf)).get(f"config{args.config}", {})
                    traffic, kname = rec.get("bytes_per_launch"), rec.get("kernel_short", kname)
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": kname,
                               "avg_launch_ms": avg_ms, "steps_per_launch": steps_per_launch,
                               "alg_bytes_per_launch": b_alg * cells_per_launch * steps_per_launch,
                               "alg_bytes_per_cell_step": b_alg}
        else:
            out["roofline"] = None
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
