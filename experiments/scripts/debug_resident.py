import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from gcm_filters_amd import _lib, testing as T, GridType
from gcm_filters_amd.kernels import ALL_KERNELS
from test_gpu_resident import _levels_by_launches
grid, shape = sys.argv[1], (int(sys.argv[2]), int(sys.argv[3]))
f, gv = T.scalar_case(grid, shape)
plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
d = torch.from_numpy(f).cuda()
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
for n in (5, 6, 8, 10, 13, 16, 24, 32):
    p = rng.uniform(-0.3, 0.3, n + 1)
    cut = plan.clenshaw_cut(n)
    want, _, _ = _levels_by_launches(plan, d, p, 0.21, n, cut, torch)
    got = torch.zeros_like(d)
    plan.resident_levels(None, None, None, None, d.data_ptr(), got.data_ptr(), p[:n][::-1], p[n], 0.21, _lib.STEP_FIRST | _lib.STEP_LAST, 0, shape[0], stream=s)
    torch.cuda.synchronize()
    diff = (got - want).abs().cpu().numpy()
    bad = np.argwhere(diff > 0)
    print(n, cut, plan.last_kernel_geometry(), "max diff", diff.max(), "n bad", len(bad), "rows", (bad[:, 0].min(), bad[:, 0].max()) if len(bad) else None,
          "cols", (bad[:, 1].min(), bad[:, 1].max()) if len(bad) else None, flush=True)
