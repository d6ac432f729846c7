"""bench.py --gpus N: the launcher of the N ranks (self_launch) and the multi-rank measurement (main_multi): strong, weak and batched-strong
figures in one line, one rank per GPU, row slabs with halo exchange (gcm_filters_amd/distributed.py); config 5 shards its levels."""
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

from bench_common import *  # noqa: F401,F403  (the shared helpers: byte counts, parity, run_single, roofline_of ...)
from bench_common import _cpu_model  # noqa: F401

BENCH_PY = os.path.join(REPO, "bench.py")


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
               "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH_PY, *sys.argv[1:]]
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
