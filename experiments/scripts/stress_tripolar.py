"""Stress for a once-seen fuzz failure (TRIPOLAR_POP_WITH_LAND (145, 352) f8, batch (2, 1), n_steps 10, err 0.59 in one run of
tools/fuzz_gpu.py 407 250 under GCMF_RESIDENT=1, not reproducible with the same seed): tripolar filters (k_ringc + k_fold_band on the side
stream) interleaved with filters of other small grids (on-chip kernel when GCMF_RESIDENT allows), against the oracle.

    python experiments/scripts/stress_tripolar.py [seed] [n]"""
import sys, warnings
import numpy as np
sys.path.insert(0, "/root/repo")
warnings.simplefilter("ignore")
from gcm_filters_amd import Filter, FilterShape, GridType, testing as T
from gcm_filters_amd.kernels import clear_plan_cache
from oracle import gcmf_oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for it in range(n):
    tri = it % 2 == 0
    grid = ("TRIPOLAR_POP_WITH_LAND", "TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED")[int(rng.integers(2))] if tri else \
           ("IRREGULAR_WITH_LAND", "REGULAR", "REGULAR_WITH_LAND", "MOM5U")[int(rng.integers(4))]
    ny, nx = int(rng.integers(20, 200)), int(rng.integers(16, 200)) * 2
    shape = (ny, nx)
    nb = [(), (2, 1), (2,), (3,)][int(rng.integers(4))]
    gv = T.scalar_grid_vars(grid, shape)
    f = rng.random(nb + shape)
    if "wet_mask" in gv and rng.random() < 0.5:
        f = np.where(gv["wet_mask"] == 0, np.nan, f)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    ns = int(rng.choice([10, 11, 13, 16, 24, 31]))
    flt = Filter(filter_scale=6 * dx, dx_min=dx, n_steps=ns, grid_type=GridType[grid], grid_vars=gv)
    got = flt.apply(f)
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
    ok = np.isfinite(want)
    err = float(np.abs(got[ok] - want[ok]).max() / np.abs(want[ok]).max()) if ok.any() else 0.0
    if not (np.array_equal(np.isnan(got), np.isnan(want)) and err <= 1e-9):
        bad += 1
        again = flt.apply(f)
        e2 = float(np.abs(again[ok] - want[ok]).max() / np.abs(want[ok]).max())
        w = np.argwhere(np.abs(np.nan_to_num(got) - np.nan_to_num(want)) > 1e-9)
        print("FAIL", it, grid, shape, nb, ns, f"err {err:.3e}; the same call again: {e2:.3e}; bad rows {w[:, -2].min()}..{w[:, -2].max()} cols {w[:, -1].min()}..{w[:, -1].max()} batch {sorted(set(map(tuple, w[:, :-2])))}", flush=True)
    if it % 40 == 39:
        clear_plan_cache()
print(f"{n} cases, {bad} bad")
