"""Row-slab path on the real GPU.  A gpurun box has ONE MI355X, so the 2-, 3- and 8-rank runs here share cuda:0
and exchange halos over gloo (packed buffers staged through host memory); the slab kernels, ghost-row
geometry, s-step shrinking row ranges and the tripole fold are exactly what the RCCL run uses."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    ("REGULAR", (48, 64), 1, 1, "f8"),
    ("REGULAR_WITH_LAND_AREA_WEIGHTED", (46, 64), 3, 2, "f8"),
    ("IRREGULAR_WITH_LAND", (50, 66), 4, 2, "f8"),
    ("TRIPOLAR_POP_WITH_LAND", (48, 64), 8, 1, "f8"),
    ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (48, 64), 2, 1, "f8"),
    ("MOM5T", (48, 64), 2, 1, "f8"),
    ("VECTOR_C_GRID", (48, 64), 3, 2, "f8"),
    ("VECTOR_B_GRID", (47, 64), 2, 1, "f8"),
    ("VECTOR_C_GRID", (48, 64), 4, 3, "f4"),
    ("VECTOR_C_GRID", (48, 64), 4, 4, "f4"),    # 4 levels: blocked C-grid kernel (S = 4) on the slabs
    ("VECTOR_C_GRID", (120, 64), 3, 8, "f4"),   # ... with the overlapped exchange
    ("VECTOR_C_GRID", (60, 64), 4, 4, "f8"),    # f64: S = 2
    ("VECTOR_B_GRID", (120, 64), 3, 5, "f8"),   # blocked B-grid kernel (f64: S <= 3), padded batch, overlapped exchange
    ("VECTOR_B_GRID", (49, 64), 4, 2, "f4"),
    ("TRIPOLAR_POP_WITH_LAND", (120, 64), 4, 2, "f8"),   # NaN on land + land kept out of the state + overlapped exchange
    ("MOM5U", (64, 64), 8, 3, "f4"),            # f32 scalar fields: the reference's forward scheme by default (round 5) ...
    ("MOM5U", (64, 64), 8, 3, "f4b"),           # ... "f4b": SlabFilter(evaluation="backward"), the backward f32 slab kernels
    # 58 rows over 3 ranks = 19 / 19 / 20: only the 20-row rank overlaps its exchange with the first launch and sends its
    # rows before zeroing their land; the receivers run the land-mask kernels in LAND_ZERO mode (found by tools/fuzz_slabs.py)
    ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (58, 16), 5, 5, "f8"),
    ("REGULAR_WITH_LAND", (58, 16), 5, 2, "f4"),
    ("REGULAR_WITH_LAND", (58, 16), 5, 2, "f4b"),
    # round 6: nine levels per launch on the slabs of an f64 flux grid without a tripole seam (a ghost zone of >= 9 rows, every rank >= 64
    # rows, decided collectively; a batch takes them at any height): 27 = 3 x 9
    ("IRREGULAR_WITH_LAND", (216, 64), 12, 2, "f8", 27),
    ("MOM5U", (216, 64), 9, 3, "f8", 36),
    # f32 POP, n_steps 19 = launches of 7 + 7 + 5 levels on slabs of 14 rows with 14 ghost rows: the second launch of the top rank starts
    # exactly 2 S rows below the seam (k_fold_band's input rows; found by tools/fuzz_slabs.py -- the row-range check was one row too strict)
    ("TRIPOLAR_POP_WITH_LAND", (28, 44), 16, 5, "f4b", 19),
    ("TRIPOLAR_POP_WITH_LAND", (40, 64), 16, 2, "f8", 24),
]


# eight ranks (BASELINE configs 4 and 5 are 8-GPU configs): slabs of 16-20 rows, the blocked kernels between exchanges
CASES_8 = [
    ("IRREGULAR_WITH_LAND", (144, 64), 8, 2, "f8"),
    ("TRIPOLAR_POP_WITH_LAND", (160, 64), 8, 2, "f8"),
    ("REGULAR_WITH_LAND", (131, 64), 5, 2, "f4"),
    ("REGULAR_WITH_LAND", (131, 64), 5, 2, "f4b"),
    ("VECTOR_C_GRID", (128, 64), 4, 4, "f4"),
    ("IRREGULAR_WITH_LAND", (560, 64), 10, 2, "f8", 27),    # nine levels per launch on eight slabs
]


def _free_port():
    from gcm_filters_amd.testing import free_port
    return free_port()     # (below the ephemeral range: see its docstring)


def _worker(rank, world, port, q, exchange="auto"):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
    from gcm_filters_amd.distributed import SlabFilter
    from oracle import gcmf_oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        for grid, shape, halo, nbatch, dt, *rest in (CASES_8 if world == 8 else CASES):
            if rest and world not in (2, 8):
                res[f"{grid}-{shape}-h{halo}-b{nbatch}-{dt}"] = (0.0, 0.0)   # (a two-rank geometry)
                continue
            name, back32 = f"{grid}-{shape}-h{halo}-b{nbatch}-{dt}", dt == "f4b"
            dt = "f4" if back32 else dt
            ev = "backward" if back32 else "auto"     # (f32 scalar fields: backward only when asked for, round 5)
            vec = grid in T.VECTOR_GRIDS
            gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
            fields = [np.stack([T.random_field(shape, 7 + 10 * c + b) for b in range(nbatch)]) for c in range(2 if vec else 1)]
            nanland = (not vec) and "wet_mask" in gv and nbatch >= 2   # ocean-like input in the last batch entry
            if nanland:
                fields[0][-1] = np.where(gv["wet_mask"] == 0, np.nan, fields[0][-1])
            if dt == "f4":
                gv = {k: v.astype(np.float32) for k, v in gv.items()}
                fields = [f.astype(np.float32) for f in fields]
            dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
            fk = dict(filter_scale=5.0 * dx, dx_min=dx, filter_shape="GAUSSIAN", **({"n_steps": rest[0]} if rest else {}))
            sf = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=halo, dtype=np.dtype(dt), device=0, exchange=exchange, evaluation=ev)
            if exchange == "p2p":   # peer stores into IPC-mapped mailboxes (csrc/gcmf_p2p.hip); rows that are not a multiple of 16 bytes fall back
                assert sf.exchange_kind == ("p2p" if (shape[1] * np.dtype(dt).itemsize) % 16 == 0 else "torch")
            sf.overlap = True   # small test slabs: force the overlapped (edge strips first) exchange where it fits
            sf.time_kernels = True
            got = sf.gather_to_global(sf.apply_local(sf.scatter_from_global(fields)))
            sf.collect_kernel_times()
            assert not sf.p2p_timed_out(), grid
            if dt == "f4" and not vec and not back32:
                assert not sf.backward_cut, name      # f32 scalar fields: the forward recurrence unless asked otherwise
            if sf.backward_cut:   # flux kinds: the slabs evaluate backwards like the one-GPU path (k_ringc; k_ringcs = its early-exit form for slabs)
                kern = sf.engine.plan.last_kernel()
                assert any(k in kern for k in ("k_ringc<", "k_ringcs<", "k_ringcp<", "k_ringcz<")), (grid, kern)   # (k_ringcp: packed batches, k_ringcz: zipped pairs, round 6)
            if rest and rest[0] in (27, 36) and halo >= 9:   # the nines were agreed on and taken (SlabFilter._cut_for)
                assert sf._cut9 and max(sf._cut_for(nbatch)) == 9, (name, sf._cut9, sf.backward_cut)
            if vec:   # the blocked vector kernels really ran on the slabs
                assert sf.kernel_launches < sf.n_steps, (sf.kernel_launches, sf.n_steps)
            if rank == 0:
                flt = Filter(filter_scale=fk["filter_scale"], dx_min=dx, grid_type=GridType[grid], grid_vars=gv, n_steps=fk.get("n_steps", 0),
                             evaluation=ev)
                one = flt.apply_to_vector(*fields) if vec else (flt.apply(fields[0]),)
                spec = O.make_spec(fk["filter_scale"], dx, "GAUSSIAN", n_steps=fk.get("n_steps", 0))
                with np.errstate(all="ignore"):
                    want = O.filter_func_vec(spec, grid, *fields, gv) if vec else (O.filter_func(spec, grid, fields[0], gv),)
                for g, w, o in zip(got, want, one):
                    assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.isnan(g), np.isnan(o)), grid
                nz = lambda a: np.nan_to_num(a, nan=0.0)
                e_one = max(float(np.abs(nz(g) - nz(w)).max() / np.abs(nz(w)).max()) for g, w in zip(got, one))
                e_ref = max(float(np.abs(nz(g) - nz(w)).max() / np.abs(nz(w)).max()) for g, w in zip(got, want))
                res[name] = (e_one, e_ref)
        if rank == 0:
            q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "auto"), (3, "auto"), (8, "auto"), (2, "p2p"), (3, "p2p"), (8, "p2p")])
def test_slabs_on_one_gpu_match_single_domain(world, exchange):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    res = q.get()
    assert len(res) == len(CASES_8 if world == 8 else CASES)
    for name, (e_one, e_ref) in res.items():
        f32 = name.endswith(("f4", "f4b"))
        assert e_one <= (1e-5 if f32 else 1e-13), (name, e_one)   # slab run == single-domain GPU run
        assert e_ref <= (1e-4 if f32 else 1e-11), (name, e_ref)   # and == the reference


def _worker_skipped_post(rank, world, port, q, native_driver):
    """Rank 1 "forgets" the post of its second exchange (gcmf_p2p_debug_skip_post): the ranks that wait for it must not carry on
    with stale ghost rows.  Every rank reports what its host saw and how long that took."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import time
    import torch
    import torch.distributed as dist
    from gcm_filters_amd import _lib, testing as T
    from gcm_filters_amd.distributed import SlabFilter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GCMF_P2P_TIMEOUT_MS"] = "1000"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid, shape = "IRREGULAR_WITH_LAND", (96, 64)
        gv = T.scalar_grid_vars(grid, shape)
        dx = T.grid_dx_min(grid, gv)
        fk = dict(filter_scale=12.0 * dx, dx_min=dx, filter_shape="GAUSSIAN")
        sf = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=8, dtype=np.float64, device=0, exchange="p2p")
        sf.native_driver = native_driver
        assert sf.exchange_kind == "p2p" and sf.backward_cut
        local = sf.scatter_from_global([T.random_field(shape, 3)[None]])
        good = sf.apply_local(local)[0].clone()          # a healthy application first (brings the mailboxes up)
        sf.synchronize()
        assert torch.isfinite(good).all()
        seqs = [None] * world
        dist.all_gather_object(seqs, sf.p2p.seq())
        assert len(set(seqs)) == 1 and seqs[0] >= 2, seqs    # every rank has started the same number of exchanges
        if rank == 1:
            sf.p2p.debug_skip_post(2 * sf.p2p.seq())     # the last exchange of the next application never leaves rank 1
        dist.barrier()
        t0 = time.perf_counter()
        status, last = [], None
        for _ in range(2):      # the application with the dropped post, and one more
            try:
                last = sf.apply_local(local)[0]
                sf.synchronize()
                status.append(None)
            except _lib.GcmfError as e:
                status.append(e.status)
        el = time.perf_counter() - t0
        q.put((rank, status, el, bool(torch.isnan(last).all()), sf.p2p.failed()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("native_driver", [True, False])
def test_p2p_rank_that_skips_a_post_fails_every_rank_loudly(native_driver):
    """a timed-out wait used to copy stale mailbox rows and carry on.  Now the waiting rank fails within the
    time-out (1 s here), its neighbours are aborted at once, every result is NaN and every host raises GCMF_ERR_P2P_TIMEOUT at its
    next synchronisation and at its next application."""
    import torch.multiprocessing as mp
    from gcm_filters_amd import _lib
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_skipped_post, args=(r, world, port, q, native_driver)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    for p in procs:
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    got = sorted(q.get() for _ in range(world))
    for rank, status, el, all_nan, why in got:
        # the ranks that waited for the dropped rows fail in that very application; rank 1 itself received everything it needed (its
        # result of that application is right) and fails in the next one, when its neighbours' abort reaches its first wait
        assert status[1] == _lib.ERR_P2P_TIMEOUT and (rank == 1 or status[0] == _lib.ERR_P2P_TIMEOUT), got
        assert el < 3.0, got                      # one time-out (1 s) + the abort going round, not a time-out per rank and exchange
        assert all_nan and why in (1, 2), got     # the last result is NaN on the device as well
    assert any(why == 1 for *_, why in got)       # somebody's wait ran out; the others may have been aborted by it
