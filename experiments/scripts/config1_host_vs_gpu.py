"""BASELINE config 1 (REGULAR 512 x 512, n_steps 16): is a back-to-back sequence of applications bound by the host (Python + launches) or by the GPU?"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
shape = (512, 512)
flt = Filter(filter_scale=4, dx_min=1, n_steps=16, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType.REGULAR)
d = torch.from_numpy(T.random_field(shape, 1)).cuda()
for _ in range(50): flt.apply(d)
torch.cuda.synchronize()
N = 2000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(N): flt.apply(d)
t_host = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{N} applications: host loop {1e6*t_host/N:.1f} us each, until the GPU is done {1e6*t_all/N:.1f} us each, GPU span (events) {1e3*e0.elapsed_time(e1)/N:.1f} us each")
plan = ALL_KERNELS[GridType.REGULAR]()._plan(_lib.F64, shape)
print("kernel of the loop above:", plan.last_kernel())
for reps in (200, 200, 2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): flt.apply(d)
    th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"{reps} applications: host loop {1e6*th/reps:.1f} us each, until the GPU is done {1e6*ta/reps:.1f} us each")
plan.set_timing(2); flt.apply(d); print("kernel time of one application (events around the launches):", plan.last_timing(), plan.last_kernel()); plan.set_timing(False)
# the pieces of the host side
import timeit
lap = ALL_KERNELS[GridType.REGULAR]()
print("Laplacian._plan lookup:", 1e6 * timeit.timeit(lambda: lap._plan(_lib.F64, shape, 0), number=2000) / 2000, "us")
print("torch.empty:", 1e6 * timeit.timeit(lambda: torch.empty(shape, dtype=torch.float64, device=d.device), number=2000) / 2000, "us")
print("current_stream:", 1e6 * timeit.timeit(lambda: torch.cuda.current_stream(0).cuda_stream, number=2000) / 2000, "us")
