"""Row-slab decomposition + s-step halo exchange on CPU: world_size 2, 3 and 8 over gloo, the per-slab step
executed by a test-only oracle engine (tests/slab_engines.py).  Checks that the halo choreography of
gcm_filters_amd.distributed reproduces the single-domain oracle filter."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcm_filters_amd import testing as T
from gcm_filters_amd.distributed import SlabFilter, slab_bounds
from oracle import gcmf_oracle as O


def _free_port():
    from gcm_filters_amd.testing import free_port
    return free_port()     # (below the ephemeral range: see its docstring)


CASES = [
    # grid, shape, halo, nbatch
    ("REGULAR", (24, 16), 1, 1),
    ("REGULAR", (24, 16), 4, 2),
    ("REGULAR_WITH_LAND_AREA_WEIGHTED", (22, 16), 3, 1),
    ("IRREGULAR_WITH_LAND", (25, 18), 2, 2),
    ("TRIPOLAR_POP_WITH_LAND", (24, 16), 3, 1),
    ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (24, 16), 8, 1),
    ("MOM5U", (24, 16), 2, 1),
    ("VECTOR_C_GRID", (24, 16), 3, 1),
    ("VECTOR_B_GRID", (23, 16), 2, 2),
    ("VECTOR_C_GRID", (24, 16), 4, 4),   # 4 levels: the blocked C-grid path (S = 4) between exchanges
    ("VECTOR_C_GRID", (72, 16), 3, 4),   # ... with the overlapped exchange (S = 3 uses up the ghost zone)
    ("VECTOR_B_GRID", (72, 16), 4, 3),   # blocked B-grid path, padded batch
    # slabs tall enough (rows_owned >= 4 halo) for the overlapped exchange: edge rows first, interior during the transfer
    ("IRREGULAR_WITH_LAND", (66, 16), 2, 1),
    ("REGULAR_WITH_LAND_AREA_WEIGHTED", (72, 16), 4, 2),
    ("TRIPOLAR_POP_WITH_LAND", (96, 16), 4, 1),
    ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (100, 16), 3, 1),
]


# BASELINE configs 4 and 5 are 8-GPU configs (config 4: 300 rows per rank): the same choreography on 8 ranks, slabs of 8-12 rows
CASES_8 = [
    ("IRREGULAR_WITH_LAND", (64, 16), 2, 1),
    ("TRIPOLAR_POP_WITH_LAND", (96, 16), 4, 1),     # rank 0 has no southern neighbour, rank 7 folds onto itself
    ("TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED", (80, 16), 3, 2),
    ("REGULAR_WITH_LAND", (67, 16), 8, 1),           # uneven slabs (8 or 9 rows), halo as deep as a slab
    ("VECTOR_C_GRID", (64, 16), 3, 4),
    # the batched strong-scaling workload of `bench.py --gpus 8` (round 5: `batched_strong`, --batch-levels): time levels through the slab path
    ("TRIPOLAR_POP_WITH_LAND", (96, 16), 4, 8),
    ("IRREGULAR_WITH_LAND", (64, 16), 2, 16),
]


# The backward (Clenshaw) slab driver (SlabFilter._apply_backward) with the overlapped exchange: (grid, shape, halo, depth of a
# launch, n_steps).  halo not a multiple of the depth leaves ghost rows over (v_out > 0) when the exchange is posted -- the
# case ADVICE r2 found sending rows the interior launch had not written yet.
CASES_BACKWARD = [
    ("IRREGULAR_WITH_LAND", (72, 16), 6, 4, 13),        # valid 6 -> 2 (overlap, v_out 2) -> 6 -> 2 ...
    ("REGULAR_WITH_LAND_AREA_WEIGHTED", (96, 16), 7, 5, 17),
    ("REGULAR", (72, 16), 4, 4, 12),                     # halo a multiple of the depth: v_out = 0 at every exchange
    ("MOM5T", (90, 16), 5, 3, 11),
    ("IRREGULAR_WITH_LAND", (24, 16), 4, 3, 9),         # slabs too short to overlap: one launch + blocking exchange
]


def _problem(grid, shape, nbatch):
    vec = grid in T.VECTOR_GRIDS
    gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
    ncomp = 2 if vec else 1
    fields = [np.stack([T.random_field(shape, 7 + 10 * c + b) for b in range(nbatch)]) for c in range(ncomp)]
    dimensional = O.DIMENSIONAL[grid]
    dx = T.grid_dx_min(grid, gv) if dimensional else 1.0
    fk = dict(filter_scale=5.0 * dx, dx_min=dx, filter_shape="GAUSSIAN")
    return gv, fields, fk


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from slab_engines import OracleSlabEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    errs = {}
    try:
        for grid, shape, halo, nbatch in (CASES_8 if world == 8 else CASES):
            gv, fields, fk = _problem(grid, shape, nbatch)
            sf = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=halo, engine_factory=OracleSlabEngine, device=-1)
            sf.overlap = True   # small test slabs: force the overlapped (edge strips first) exchange where it fits
            local = sf.scatter_from_global(fields)
            out = sf.apply_local(local)
            got = sf.gather_to_global(out)
            spec = O.make_spec(fk["filter_scale"], fk["dx_min"], "GAUSSIAN")
            assert spec.n_steps == sf.n_steps
            with np.errstate(all="ignore"):
                if len(fields) == 2:
                    want = O.filter_func_vec(spec, grid, fields[0], fields[1], gv)
                else:
                    want = (O.filter_func(spec, grid, fields[0], gv),)
            e = max(float(np.abs(g - w).max() / np.abs(w).max()) for g, w in zip(got, want))
            errs[f"{grid}-{shape}-h{halo}-b{nbatch}"] = (e, sf.exchanges, sf.n_steps, sf.halo)
        for grid, shape, halo, depth, n in (CASES_BACKWARD if world in (2, 3) else []):
            from slab_engines import OracleClenshawSlabEngine
            gv, fields, fk = _problem(grid, shape, 2)
            fk["n_steps"] = n
            eng = type("Eng", (OracleClenshawSlabEngine,), {"DEPTH": depth})
            sf = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=halo, engine_factory=eng, device=-1)
            assert sf.backward_cut and max(sf.backward_cut) == depth
            sf.overlap = True
            got = sf.gather_to_global(sf.apply_local(sf.scatter_from_global(fields)))
            spec = O.make_spec(fk["filter_scale"], fk["dx_min"], "GAUSSIAN", n_steps=n)
            with np.errstate(all="ignore"):
                want = O.filter_func(spec, grid, fields[0], gv)
            e = float(np.abs(got[0] - want).max() / np.abs(want).max())
            errs[f"backward-{grid}-{shape}-h{halo}-d{depth}-n{n}"] = (e, None, n, halo)
        if rank == 0:
            q.put(errs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_filter_matches_single_domain(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    errs = q.get()
    assert len(errs) == (len(CASES_8) if world == 8 else len(CASES) + len(CASES_BACKWARD))
    for name, (e, nex, n, halo) in errs.items():
        assert e < 1e-12, (name, e)
        if nex is not None:
            assert nex == -(-n // halo), (name, nex, n, halo)  # one exchange per `halo` steps


def test_slab_bounds_cover_grid():
    for ny in (7, 24, 2400):
        for world in (1, 2, 3, 8):
            if ny < world:
                continue
            edges = [slab_bounds(ny, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == ny
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
