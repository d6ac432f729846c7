"""The halo exchange issued by libgcmf itself (gcmf_halo_start / gcmf_halo_finish: RCCL send / recv on a side stream).
A gpurun box has one MI355X and RCCL refuses two ranks on one device, so the native path runs here as a ring of ONE
rank: the slab keeps ghost rows and its two neighbours are itself (GCMF_PLAN_SELF_RING) -- the same packing, posting
order, stream / event choreography and overlap with the interior launch as on N GPUs."""
import numpy as np
import pytest

from gcm_filters_amd import Filter, GridType, _lib, testing as T
from gcm_filters_amd.distributed import SlabFilter
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


def test_raw_exchange_fills_ghost_rows_direct_and_packed():
    import torch
    comm = _lib.Comm(_lib.Comm.unique_id(), 1, 0, 0)
    info = comm.describe()      # what RCCL itself says about the communicator (bench.py prints it with the N > 1 line)
    assert info["nranks"] == 1 and info["rank"] == 0 and info["rccl_version_code"] > 20000, info
    s = torch.cuda.current_stream().cuda_stream
    for nblocks, dt, code in [(1, torch.float64, _lib.F64), (3, torch.float32, _lib.F32), (12, torch.float64, _lib.F64)]:
        halo, owned, nx = 3, 20, 64
        rows = owned + 2 * halo
        a = torch.arange(nblocks * rows * nx, dtype=dt, device="cuda").reshape(nblocks, rows, nx)
        b = -a.clone()
        ea, eb = a.clone(), b.clone()
        for e in (ea, eb):
            e[:, :halo] = e[:, owned: owned + halo]              # south ghosts <- the northern owned rows (periodic)
            e[:, halo + owned:] = e[:, halo: 2 * halo]           # north ghosts <- the southern owned rows
        comm.halo_start([a.data_ptr(), b.data_ptr()], nblocks, rows, nx, halo, owned, halo, code, 0, 0, stream=s)
        comm.halo_finish(stream=s)
        torch.cuda.synchronize()
        assert torch.equal(a, ea) and torch.equal(b, eb), nblocks
    with pytest.raises(_lib.GcmfError):
        comm.halo_finish(stream=s)                              # nothing in flight
    comm.close()


CASES = [
    ("REGULAR", (96, 128), 4, 1, "f8"),
    ("REGULAR_WITH_LAND", (120, 128), 8, 2, "f8"),               # NaN on land in the last batch entry
    ("IRREGULAR_WITH_LAND", (130, 132), 8, 1, "f8"),
    ("MOM5U", (96, 64), 5, 3, "f4"),
    ("VECTOR_C_GRID", (128, 64), 4, 8, "f4"),                   # 16 blocks per state: the packed message path
    ("VECTOR_B_GRID", (120, 64), 3, 2, "f8"),
]


@pytest.mark.parametrize("exchange", ["native", "p2p"])
@pytest.mark.parametrize("grid,shape,halo,nbatch,dt", CASES)
def test_self_ring_native_exchange_equals_single_domain(grid, shape, halo, nbatch, dt, exchange):
    """exchange="p2p": the same ring of one rank through the mailbox / flag kernels of csrc/gcmf_p2p.hip (its neighbours' blocks are its own)."""
    vec = grid in T.VECTOR_GRIDS
    gv = T.vector_grid_vars(grid, shape) if vec else T.scalar_grid_vars(grid, shape)
    fields = [np.stack([T.random_field(shape, 7 + 10 * c + b) for b in range(nbatch)]) for c in range(2 if vec else 1)]
    if (not vec) and "wet_mask" in gv and nbatch >= 2:
        fields[0][-1] = np.where(gv["wet_mask"] == 0, np.nan, fields[0][-1])
    if dt == "f4":
        gv = {k: v.astype(np.float32) for k, v in gv.items()}
        fields = [f.astype(np.float32) for f in fields]
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    fk = dict(filter_scale=6.0 * dx, dx_min=dx, filter_shape="GAUSSIAN")
    sf = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=halo, dtype=np.dtype(dt), device=0, rank=0, world=1,
                    self_ring=True, exchange=exchange)
    sf.overlap = True
    assert sf.exchange_kind == exchange and sf.halo == halo and sf.rows_alloc == shape[0] + 2 * halo
    got = [t.cpu().numpy() for t in sf.apply_local(sf.scatter_from_global(fields))]
    assert sf.exchanges >= (sf.n_steps - 1) // halo and not sf.p2p_timed_out()
    if vec and halo >= 4:
        # vector kinds with a library-issued exchange: one call into libgcmf per application too (gcmf_slab_apply_backward_vec: the backward
        # kernels, cut as gcmf_apply cuts them) -- not the forward Python choreography
        kern = sf.engine.plan.last_kernel()   # (reading resets it)
        assert sf._vec_backward.get(nbatch) == 1 and ("stream2c<" in kern or "k_cgrid_ring<" in kern), kern
    if sf.backward_cut and not vec:
        # the scalar backward path ran inside libgcmf in one call (gcmf_slab_apply_backward); the Python choreography gives the same bits
        assert sf.native_driver
        n_native = sf.exchanges
        sf.native_driver, sf.exchanges = False, 0
        again = [t.cpu().numpy() for t in sf.apply_local(sf.scatter_from_global(fields))]
        assert sf.exchanges == n_native
        assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(got, again))
        # evaluation="reference": the forward recurrence on the slab (bit-identical with the single-domain forward filter)
        sfr = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=halo, dtype=np.dtype(dt), device=0, rank=0, world=1, self_ring=True,
                         exchange=exchange, evaluation="reference")
        assert not sfr.backward_cut
        fwd = [t.cpu().numpy() for t in sfr.apply_local(sfr.scatter_from_global(fields))]
        ref = Filter(filter_scale=fk["filter_scale"], dx_min=dx, grid_type=GridType[grid], grid_vars=gv, evaluation="reference").apply(fields[0])
        assert np.array_equal(fwd[0], ref, equal_nan=True)
    flt = Filter(filter_scale=fk["filter_scale"], dx_min=dx, grid_type=GridType[grid], grid_vars=gv)
    one = flt.apply_to_vector(*fields) if vec else (flt.apply(fields[0]),)
    for g, o in zip(got, one):
        assert np.array_equal(np.isnan(g), np.isnan(o))
        ok = ~np.isnan(o)
        assert np.abs(g[ok] - o[ok]).max() <= (1e-5 if dt == "f4" else 1e-13) * np.abs(o[ok]).max()
