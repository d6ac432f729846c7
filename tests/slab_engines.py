"""TEST-ONLY stand-in for the HIP slab engine of gcm_filters_amd.distributed, so that the row-slab / halo
choreography can run under the gloo backend on CPU.  It executes a slab step by embedding the slab rows at
their global positions in a poisoned (ny, nx) array and applying the ORACLE's global Laplacian: rows whose
stencil only touches valid rows come out right, anything that reads a stale/ghost row picks up the poison."""
import numpy as np

from oracle import gcmf_oracle as O

POISON = 1e30
STEP_FIRST, STEP_LAST = 1, 2


class OracleSlabEngine:
    def __init__(self, grid_type, dtype_code, ny, nx, planes, row_begin, row_end, halo, device):
        self.name = grid_type.name
        self.ny, self.nx = ny, nx
        names = O.GRID_ARGS[self.name]
        self.gv = dict(zip(names, planes))
        self.lap = O.make_laplacian(self.name, self.gv)
        tripolar = self.name.startswith("TRIPOLAR")
        full = row_begin == 0 and row_end == ny
        gs = 0 if (full or (tripolar and row_begin == 0)) else halo
        gn = 0 if (full or (tripolar and row_end == ny)) else halo
        self.rows_owned = row_end - row_begin
        self.rows_alloc = gs + self.rows_owned + gn
        self.first_owned = gs
        self.gidx = (np.arange(self.rows_alloc) + row_begin - gs) % ny
        self.area = self.gv["area"] if self.name in O.AREA_WEIGHTED else None

    def _embed(self, x):  # (nbatch, rows_alloc, nx) -> (nbatch, ny, nx)
        g = np.full((x.shape[0], self.ny, self.nx), POISON, dtype=x.dtype)
        g[:, self.gidx, :] = x
        return g

    def prepare(self, ins, outs, nbatch, row_lo, row_hi):
        for i, o in zip(ins, outs):
            a = 1 if self.area is None else self.area[self.gidx[row_lo:row_hi]]
            o.numpy()[:, row_lo:row_hi, :] = i.numpy()[:, row_lo:row_hi, :] * a

    def step(self, t1, t2, fb_in, t0, fb_out, coef0, coef1, c, mode, nbatch, row_lo, row_hi):
        x = [self._embed(t.numpy()) for t in t1]
        with np.errstate(all="ignore"):
            L = self.lap(*x)
        L = L if isinstance(L, tuple) else (L,)
        rows = slice(row_lo, row_hi)
        for k in range(len(t1)):
            l_loc = L[k][:, self.gidx, :][:, rows, :]
            xc = t1[k].numpy()[:, rows, :]
            a = -xc - c * l_loc
            if mode & STEP_FIRST:
                tk = a
                fb = coef0 * xc + coef1 * a
            else:
                tk = 2 * a - t2[k].numpy()[:, rows, :]
                fb = fb_in[k].numpy()[:, rows, :] + coef0 * tk
            if mode & STEP_LAST and self.area is not None:
                fb = fb / self.area[self.gidx[rows]]
            if not (mode & STEP_LAST):
                t0[k].numpy()[:, rows, :] = tk
            fb_out[k].numpy()[:, rows, :] = fb

    # --- temporally blocked advance, emulated with S single oracle steps on a shrinking row range ---
    def multi_supported(self, S, nbatch=1):
        if self.name in O.VECTOR:  # mirrors cgrid_multi_supported / bgrid_multi_supported: S <= 4
            return S in (2, 3, 4, 5, 6) and self.rows_alloc >= S + 2
        return 2 <= S <= 8 and self.rows_alloc >= 3 * S + 2

    def multi(self, u, v, uo, vo, fb_in, fb_out, pk, p0, c, mode, nbatch, row_lo, row_hi):
        import torch
        S = len(pk)
        nc = len(u)
        first, last = bool(mode & STEP_FIRST), bool(mode & STEP_LAST)
        lvl = {0: [x.clone() for x in u], -1: None if v is None else [x.clone() for x in v]}
        if first and self.area is not None:
            lvl[0] = [x * torch.from_numpy(self.area[self.gidx]) for x in lvl[0]]
        fb = None if first else [x.clone() for x in fb_in]
        for t in range(1, S + 1):
            lo, hi = max(row_lo - (S - t), 0), min(row_hi + (S - t), self.rows_alloc)
            t0 = [torch.full_like(x, POISON) for x in u]
            fbo = [torch.full_like(x, POISON) for x in fb_out] if fb is None else [x.clone() for x in fb]
            m = (STEP_FIRST if (first and t == 1) else 0)
            self.step(lvl[t - 1], lvl[t - 2], fb, t0, fbo,
                      p0 if (first and t == 1) else pk[t - 1], pk[0], c, m, nbatch, lo, hi)
            lvl[t] = t0
            fb = fbo
        rows = slice(row_lo, row_hi)
        for q in range(nc):
            if last:
                res = fb[q][:, rows, :]
                if self.area is not None:
                    res = res / torch.from_numpy(self.area[self.gidx[rows]])
                fb_out[q][:, rows, :] = res
            else:
                uo[q][:, rows, :] = lvl[S][q][:, rows, :]
                vo[q][:, rows, :] = lvl[S - 1][q][:, rows, :]
                fb_out[q][:, rows, :] = fb[q][:, rows, :]


STEP_CLENSHAW = 0x10


class OracleClenshawSlabEngine(OracleSlabEngine):
    """The same stand-in offering the BACKWARD evaluation (gcmf_cheb_multi with GCMF_STEP_CLENSHAW) so
    that SlabFilter._apply_backward runs under gloo: b_n = p_n f, b_k = p_k f + 2A(b_{k+1}) - b_{k+2}, result
    p_0 f + A(b_1) - b_2, cut into launches of DEPTHS levels (tests set the class attribute)."""
    DEPTH = 4

    def clenshaw_cut(self, n_steps):
        if self.name in O.VECTOR:
            return []
        cut, left = [], n_steps
        while left > 0:
            cut.append(min(self.DEPTH, left))
            left -= cut[-1]
        return cut

    def has_land(self):
        return False          # land cells take part in the stand-in's state: no land_fix pass

    def _A(self, x, c, rows):
        """A(x) = -x - c L(x) on rows `rows` of the slab array x (nbatch, rows_alloc, nx); poison outside what is valid."""
        with np.errstate(all="ignore"):
            L = self.lap(self._embed(x))
        return -x[:, rows, :] - c * L[:, self.gidx, :][:, rows, :]

    def multi(self, u, v, uo, vo, fb_in, fb_out, pk, p0, c, mode, nbatch, row_lo, row_hi):
        if not (mode & STEP_CLENSHAW):
            return super().multi(u, v, uo, vo, fb_in, fb_out, pk, p0, c, mode, nbatch, row_lo, row_hi)
        S = len(pk)
        first, last = bool(mode & STEP_FIRST), bool(mode & STEP_LAST)
        f = fb_in[0].numpy().copy()
        if self.area is not None:
            f = f * self.area[self.gidx]
        if first:
            b1, b2 = p0 * f, np.zeros_like(f)
        else:
            b1, b2 = u[0].numpy().copy(), v[0].numpy().copy()
        for t in range(1, S + 1):
            lo, hi = max(row_lo - (S - t), 0), min(row_hi + (S - t), self.rows_alloc)
            rows = slice(lo, hi)
            new = np.full_like(b1, POISON)
            two = 1.0 if (last and t == S) else 2.0
            new[:, rows, :] = pk[t - 1] * f[:, rows, :] + two * self._A(b1, c, rows) - b2[:, rows, :]
            b1, b2 = new, b1
        rows = slice(row_lo, row_hi)
        if last:
            res = b1[:, rows, :]
            if self.area is not None:
                res = res / self.area[self.gidx[rows]]
            fb_out[0].numpy()[:, rows, :] = res
        else:
            uo[0].numpy()[:, rows, :] = b1[:, rows, :]
            vo[0].numpy()[:, rows, :] = b2[:, rows, :]
