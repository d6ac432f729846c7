"""Odd input layouts and dtypes through Filter.apply on the GPU: non-contiguous / negatively strided / sliced numpy and
torch inputs, float16 / int / bool / big-endian fields, host and device; compared with the oracle on the converted values."""
import sys, warnings
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, GridType, testing as T
from oracle import gcmf_oracle as O
warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
ncase = 0
for grid in ("REGULAR", "REGULAR_WITH_LAND", "IRREGULAR_WITH_LAND", "TRIPOLAR_POP_WITH_LAND"):
    shape = (40, 64)
    _, gv = T.scalar_case(grid, shape)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    flt = Filter(filter_scale=4 * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gv)
    spec = O.make_spec(4 * dx, dx, "GAUSSIAN")
    base = rng.standard_normal((3, 2 * shape[0], 2 * shape[1]))
    variants = {
        "sliced": lambda: base[:, ::2, ::2],
        "negstride": lambda: base[:, : shape[0], : shape[1]][:, ::-1, ::-1],
        "transposed": lambda: np.ascontiguousarray(base[:, : shape[1], : shape[0]]).transpose(0, 2, 1),
        "f16": lambda: base[:, : shape[0], : shape[1]].astype(np.float16),
        "f32": lambda: base[:, : shape[0], : shape[1]].astype(np.float32),
        "int32": lambda: (base[:, : shape[0], : shape[1]] * 10).astype(np.int32),
        # (bool fields: the reference raises numpy's "boolean negative" TypeError; this build filters them as 0 / 1)
        "bigendian": lambda: base[:, : shape[0], : shape[1]].astype(">f8"),
        "fortran": lambda: np.asfortranarray(base[:, : shape[0], : shape[1]]),
        "2d": lambda: base[0, : shape[0], : shape[1]],
        "5d": lambda: base[:, : shape[0], : shape[1]].reshape(1, 3, 1, *shape),
    }
    for name, mk in variants.items():
        f = mk()
        for dev in ("host", "cuda"):
            ncase += 1
            try:
                if dev == "cuda":
                    if f.dtype.byteorder == ">" or f.dtype == np.bool_ and False:
                        continue
                    x = torch.from_numpy(np.ascontiguousarray(f)).cuda() if any(s < 0 for s in f.strides) else torch.from_numpy(f).cuda()
                    got = flt.apply(x).cpu().numpy()
                else:
                    got = flt.apply(f)
                with np.errstate(all="ignore"):
                    want = O.filter_func(spec, grid, np.asarray(f), gv)
                if got.shape != want.shape or got.dtype != want.dtype:
                    print("SHAPE/DTYPE", grid, name, dev, got.shape, got.dtype, want.shape, want.dtype); bad += 1; continue
                tol = 2e-3 if f.dtype == np.float16 else (2e-5 if f.dtype.itemsize <= 4 and f.dtype.kind == "f" else 1e-11)
                e = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))
                if not (e <= tol):
                    print("ERR", grid, name, dev, e); bad += 1
            except Exception as ex:
                print("EXC", grid, name, dev, repr(ex)[:160]); bad += 1
print(f"{ncase} input cases, {bad} bad")
