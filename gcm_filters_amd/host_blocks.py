"""numpy in / numpy out for ONE large 2-D field: upload, recurrence and download overlapped by row blocks.

The reference's default call shape is a single host array (`filter_func(field, *grid_args)`, reference
gcm_filters/filter.py:181-214 behind `xr.apply_ufunc`, filter.py:478-486).  Through `gcmf_apply` with host pointers that
is upload (1.2 ms for 2400x3600 f64 over PCIe 5), filter (1.1 ms), download (1.2 ms) one after the other: the first blocked
launch needs every row of the field and the last one finishes every row of the result at the same time.

Here the grid is cut into K row blocks.  Every block is a slab plan of its own (`gcmf_plan_create` with `row_begin`,
`row_end`, `halo = n_steps`: the multi-GPU building block, SURVEY 8e "latency escape hatch") that carries n_steps ghost
rows per side, filled from the host array together with its own rows, so the whole polynomial runs on the block without
any exchange: a launch of S steps uses up S ghost rows (the rows it computes shrink by S per side), and what is left
after n_steps is exactly the block's own rows.  Results are bit-identical to the one-plan path (same kernels on row
ranges).  Block k+1 uploads while block k computes while block k-1 downloads (the download runs on a helper thread:
copies from / to pageable memory block the calling thread).

Measured (MI355X, PCIe 5, 2400x3600 f64 numpy in / out): config 2 3.19 -> 2.66 ms per field, config 3 3.60 -> 2.94 ms.

Costs: K extra plans (coefficient slabs of all blocks = (1 + 2 K n_steps / ny) x one set, plus eight state planes per block)
and ~50 ms to build them, so the pipeline is only built for a plan that keeps being called with single host fields
(`kernels._DeviceLaplacian._run`: from the third such call on).  `GCMF_HOST_BLOCKS=0` turns it off, `=K` fixes K.
"""
import os
import threading
import warnings
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

MIN_CELLS = 1 << 22          # below ~4 M cells the copies are too short to be worth overlapping
BUILD_AFTER_CALLS = 2        # single-field host calls on a plan before its pipeline is built
_DEPTHS = (8, 7, 6, 5, 4, 3, 2)


def configured_blocks() -> Optional[int]:
    """None: automatic; 0: off; K >= 2: that many blocks."""
    e = os.environ.get("GCMF_HOST_BLOCKS", "")
    if not e:
        return None
    k = int(e)
    return 0 if k < 2 else k


def choose_blocks(ny: int, n_steps: int) -> int:
    """Number of row blocks (0: not worth it).  More blocks overlap more of the copies, but every block recomputes
    2 * n_steps ghost rows, marches shorter strips, and concurrent uploads and downloads slow each other down.  Measured
    on 2400 x 3600 f64 (tools/measure_host_blocks.py; one plan 3.19 / 3.60 ms per field): n_steps 56: 2.78 / 2.66 / 2.76 /
    2.87 ms for K = 2 / 3 / 4 / 5, n_steps 63: 2.94 / 3.02 / 3.40 / 3.28 ms."""
    k = configured_blocks()
    if k == 0:
        return 0
    if k is None:
        k = 3 if ny // 3 >= 14 * n_steps else 2
    while k >= 2 and (ny // k < 2 * n_steps or ny // k < 128):
        k -= 1
    return k if k >= 2 else 0


def block_runs(first_row: int, rows: int, ny: int):
    """Rows [first_row, first_row + rows) of a global field that is periodic in y, as contiguous runs
    (offset in the block, global row, count)."""
    r, out = 0, []
    while r < rows:
        gj = (first_row + r) % ny
        n = min(rows - r, ny - gj)
        out.append((r, gj, n))
        r += n
    return out


class _Block:
    """One row block: slab plan + state planes + the launch schedule of gcmf_apply on shrinking row ranges."""

    def __init__(self, torch, grid_type: int, dtype: int, ny: int, nx: int, dev_planes: Sequence[int], device: int,
                 row_begin: int, row_end: int, ghost: int, skip_kappa_one: bool):
        self.torch = torch
        self.ny, self.nx, self.row_begin, self.row_end = ny, nx, row_begin, row_end
        self.plan = _lib.Plan(grid_type, dtype, ny, nx, dev_planes, device=device, row_begin=row_begin, row_end=row_end,
                              halo=ghost, planes_on_device=True, skip_kappa_one=skip_kappa_one)
        self.rows, self.fo, self.ro = self.plan.rows_alloc, self.plan.first_owned, self.plan.rows_owned
        self.gs, self.gn = self.fo, self.rows - self.fo - self.ro   # ghost rows (0 at a physical boundary of a tripolar grid)
        self.ghost = ghost
        self.tdt = torch.float64 if dtype == _lib.F64 else torch.float32
        dev = torch.device("cuda", device)
        mk = lambda dt: torch.empty((self.rows, nx), dtype=dt, device=dev)
        self.X = mk(self.tdt)
        self.pool = [mk(self.tdt) for _ in range(4)]
        self.F = [mk(torch.float64), mk(torch.float64)]
        self.O = mk(torch.float64)
        self.O32 = None
        self.has_land = self.plan.has_land()
        self.ok = all(self.plan.multi_supported(S) for S in _DEPTHS)

    def close(self):
        self.plan.close()

    # rows [row_begin - gs, row_end + gn) of the global field, wrapped in y, as contiguous runs
    def _runs(self):
        return block_runs(self.row_begin - self.gs, self.rows, self.ny)

    def upload(self, field: np.ndarray):
        t = self.torch
        with warnings.catch_warnings():   # a read-only caller array is fine: it is only read
            warnings.simplefilter("ignore", UserWarning)
            for r, gj, n in self._runs():
                self.X[r: r + n].copy_(t.from_numpy(field[gj: gj + n]), non_blocking=True)

    def run(self, p: np.ndarray, c: float, out_f32: bool, cut: Sequence[int] = ()):
        """Enqueue the whole polynomial on the current stream (the schedule of gcmf_apply, csrc/gcmf_api.hip)."""
        t = self.torch
        plan, fo, ro = self.plan, self.fo, self.ro
        n = len(p) - 1
        stream = t.cuda.current_stream().cuda_stream
        if out_f32 and self.O32 is None:
            self.O32 = t.empty((self.rows, self.nx), dtype=t.float32, device=self.X.device)
        O = self.O32 if out_f32 else self.O
        if cut:   # the backward evaluation gcmf_apply uses on the one-plan path for this polynomial: the same levels on the block
            u = v = None
            valid, lvl = self.ghost, 1
            for q, S in enumerate(cut):
                free = [b for b in self.pool if b is not u and b is not v]
                v_out = valid - S
                lo = fo - (v_out if self.gs else 0)
                hi = fo + ro + (v_out if self.gn else 0)
                last = (q == len(cut) - 1)
                mode = _lib.STEP_CLENSHAW | (_lib.STEP_FIRST if q == 0 else 0) | (_lib.STEP_LAST if last else 0)
                plan.cheb_multi(0 if u is None else u.data_ptr(), 0 if v is None else v.data_ptr(), free[0].data_ptr(),
                                free[1].data_ptr(), self.X.data_ptr(), O.data_ptr(), p[n - lvl - S + 1: n - lvl + 1][::-1], p[n], c,
                                mode, 1, lo, hi, out_f32=out_f32, stream=stream)
                u, v = free[0], free[1]
                valid = v_out
                lvl += S
            if self.has_land:
                plan.land_fix(p, c, [self.X.data_ptr()], [O.data_ptr()], 1, out_f32=out_f32, stream=stream)
            return
        Fc, Fn = self.F
        u, v = self.X, None
        valid, k, land_zeroed = self.ghost, 1, False
        while k <= n:
            left = n - k + 1
            S = next(cand for cand in _DEPTHS if cand <= left and left - cand != 1)
            free = [b for b in self.pool if b is not u and b is not v]
            v_out = valid - S
            lo = fo - (v_out if self.gs else 0)
            hi = fo + ro + (v_out if self.gn else 0)
            last = (k + S - 1 == n)
            mode = ((_lib.STEP_FIRST if k == 1 else 0) | (_lib.STEP_LAST if last else 0)
                    | (_lib.STEP_LAND_ZERO if land_zeroed else 0)
                    | (_lib.STEP_LAND_FIXED if (k == 1 and not last and self.has_land) else 0))
            plan.cheb_multi(u.data_ptr(), 0 if v is None else v.data_ptr(), free[0].data_ptr(), free[1].data_ptr(),
                            Fc.data_ptr(), O.data_ptr() if last else Fn.data_ptr(), p[k: k + S], p[0], c, mode, 1, lo, hi,
                            out_f32=out_f32, stream=stream)
            u, v = free[0], free[1]
            Fc, Fn = Fn, Fc
            if k == 1 and not last and self.has_land:
                # isolated cells leave the state (a first launch by k_ring has already taken them as zero; the general
                # kernels have not); k_land_fix writes their own polynomial into the result below
                plan.zero_land([u.data_ptr()], [v.data_ptr()], 1, stream=stream)
                land_zeroed = True
            valid = v_out
            k += S
        if land_zeroed:
            plan.land_fix(p, c, [self.X.data_ptr()], [O.data_ptr()], 1, out_f32=out_f32, stream=stream)

    def download(self, out: np.ndarray, out_f32: bool):
        O = self.O32 if out_f32 else self.O
        self.torch.from_numpy(out[self.row_begin: self.row_end]).copy_(O[self.fo: self.fo + self.ro], non_blocking=True)


class RowBlockPipeline:
    def __init__(self, grid_type: int, dtype: int, ny: int, nx: int, host_planes: Sequence[np.ndarray], device: int,
                 n_steps: int, nblocks: int, skip_kappa_one: bool = False, cut: Sequence[int] = ()):
        import torch
        self.torch = torch
        self.n_steps, self.nblocks, self.device = int(n_steps), int(nblocks), device
        self.cut = list(cut)   # launch depths of the backward evaluation the ONE-plan path uses ([]: forward recurrence)
        self.lock = threading.Lock()
        npdt = _lib.np_dtype(dtype)
        with torch.cuda.device(device):
            with warnings.catch_warnings():   # planes of a cached plan are write-protected (kernels.py); they are only read
                warnings.simplefilter("ignore", UserWarning)
                dev = [torch.from_numpy(np.ascontiguousarray(a, dtype=npdt)).cuda() for a in host_planes]
            torch.cuda.synchronize()
            base, rem = divmod(ny, nblocks)
            self.blocks: List[_Block] = []
            b = 0
            try:
                for k in range(nblocks):
                    e = b + base + (1 if k < rem else 0)
                    self.blocks.append(_Block(torch, grid_type, dtype, ny, nx, [t.data_ptr() for t in dev], device, b, e,
                                              self.n_steps, skip_kappa_one))
                    b = e
            except Exception:
                self.close()
                raise
            del dev
            self.ok = all(blk.ok for blk in self.blocks)
            if self.cut:   # the block plans evaluate the way the one-plan path does (whatever their own default would be)
                for blk in self.blocks:
                    blk.plan.set_tuning(multi_s=8, clenshaw=2)
                # ... cut as SLABS are cut (launches of 5..8; the one-plan path of a whole flux grid may use nines, round 5): the levels are
                # the same arithmetic however they are cut into launches, so the bits are the one-plan path's.  A polynomial the slabs
                # cannot evaluate backwards at all (9 levels) stays on the one-plan path.
                self.cut = list(self.blocks[0].plan.clenshaw_cut(self.n_steps)) if self.blocks else []
                if not self.cut or any(list(blk.plan.clenshaw_cut(self.n_steps)) != self.cut for blk in self.blocks):
                    self.ok = False
            self.s_up, self.s_comp, self.s_down = (torch.cuda.Stream(device) for _ in range(3))
        self.worker = ThreadPoolExecutor(1, thread_name_prefix="gcmf-d2h")

    def close(self):
        for blk in getattr(self, "blocks", []):
            blk.close()
        self.blocks = []
        w = getattr(self, "worker", None)
        if w is not None:
            w.shutdown(wait=False)

    def _download(self, blk: _Block, out: np.ndarray, out_f32: bool, ev):
        t = self.torch
        with t.cuda.device(self.device), t.cuda.stream(self.s_down):
            self.s_down.wait_event(ev)
            blk.download(out, out_f32)
            self.s_down.synchronize()

    def apply(self, p: np.ndarray, c: float, field: np.ndarray, out: np.ndarray, out_f32: bool):
        """field, out: C-contiguous (ny, nx) host arrays in the plan's state dtype / the result dtype."""
        t = self.torch
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert len(p) - 1 == self.n_steps
        with self.lock, t.cuda.device(self.device):
            futs = []
            for blk in self.blocks:
                with t.cuda.stream(self.s_up):
                    blk.upload(field)
                    ev_up = self.s_up.record_event()
                with t.cuda.stream(self.s_comp):
                    self.s_comp.wait_event(ev_up)
                    blk.run(p, c, out_f32, self.cut)
                    ev_c = self.s_comp.record_event()
                futs.append(self.worker.submit(self._download, blk, out, out_f32, ev_c))
            for f in futs:
                f.result()
