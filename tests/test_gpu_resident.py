"""The on-chip (resident) kernel, csrc/gcmf_resident.hip: up to 64 levels of the backward (Clenshaw) evaluation in ONE launch on a field
that lives in the register files + LDS of the chip -- against the strip-marching launches of 5..8 levels (bit for bit: the arithmetic of
a level is the same, operand for operand), against the oracle, and on row slabs (the 8-GPU geometry of BASELINE configs 3 / 4)."""
import os

import numpy as np
import pytest

from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
from oracle import gcmf_oracle as O

pytestmark = pytest.mark.gpu


def _levels_by_launches(plan, f, p, c, n, cut, torch):
    """The n levels as launches of 5..8 (gcmf_cheb_multi, GCMF_STEP_CLENSHAW) over the whole domain."""
    rows = f.shape[-2]
    pool = [torch.zeros_like(f) for _ in range(4)]
    out = torch.zeros_like(f)
    u = v = None
    lvl = 1
    for q, S in enumerate(cut):
        free = [b for b in pool if b is not u and b is not v]
        mode = _lib.STEP_CLENSHAW | (_lib.STEP_FIRST if q == 0 else 0) | (_lib.STEP_LAST if q == len(cut) - 1 else 0)
        pk = p[n - lvl - S + 1: n - lvl + 1][::-1]
        plan.cheb_multi_vec(None if u is None else [u.data_ptr()], None if v is None else [v.data_ptr()], [free[0].data_ptr()],
                            [free[1].data_ptr()], [f.data_ptr()], [out.data_ptr()], pk, p[n], c, mode, 1, 0, rows,
                            stream=torch.cuda.current_stream().cuda_stream)
        u, v = free[0], free[1]
        lvl += S
    return out, u, v


@pytest.mark.parametrize("grid,shape,n", [
    ("IRREGULAR_WITH_LAND", (96, 160), 16), ("IRREGULAR_WITH_LAND", (300, 364), 63), ("IRREGULAR_WITH_LAND", (37, 52), 11),
    ("MOM5U", (64, 96), 21), ("MOM5T", (100, 72), 13), ("REGULAR", (128, 128), 16), ("REGULAR_AREA_WEIGHTED", (90, 150), 24),
    ("REGULAR_WITH_LAND", (96, 160), 21), ("REGULAR_WITH_LAND_AREA_WEIGHTED", (150, 100), 56), ("REGULAR", (512, 512), 16),
    ("IRREGULAR_WITH_LAND", (600, 24), 13), ("REGULAR", (24, 600), 16), ("REGULAR_WITH_LAND", (1000, 32), 24),   # tall / flat narrow grids
    ("IRREGULAR_WITH_LAND", (364, 3600), 32),       # the slab of one of eight ranks of BASELINE config 3 with its 2 x 32 ghost rows
])
@pytest.mark.parametrize("nan", ["", "land", "wet"])
def test_resident_levels_equal_the_strip_marching_launches_bit_for_bit(grid, shape, n, nan):
    import torch
    if nan == "land" and grid in ("REGULAR", "REGULAR_AREA_WEIGHTED"):
        pytest.skip("no land")
    f, gv = T.scalar_case(grid, shape)
    if nan == "land":
        f = np.where(gv["wet_mask"] == 0, np.nan, f)
    if nan == "wet":
        wet = np.argwhere(gv["wet_mask"] != 0) if "wet_mask" in gv else np.argwhere(np.ones(shape, bool))
        j, i = wet[len(wet) // 3]
        f = f.copy()
        f[j, i] = np.nan
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=6.0 * dx, dx_min=dx, n_steps=n, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv)
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    spec = flt.filter_spec
    p = np.asarray(spec.p, dtype=np.float64)
    c = 2 / spec.s_max if ALL_KERNELS[GridType[grid]].is_dimensional else 2 / (spec.s_max * spec.dx_min_sq)
    cut = plan.clenshaw_cut(n)
    assert cut and sum(cut) == n
    L = min(n, 64)
    assert plan.resident_supported(0, shape[0], L)
    d = torch.from_numpy(f).cuda()
    want, wu, wv = _levels_by_launches(plan, d, p, c, n, cut, torch)
    s = torch.cuda.current_stream().cuda_stream
    # the whole polynomial in one launch
    got = torch.zeros_like(d)
    plan.resident_levels(None, None, None, None, d.data_ptr(), got.data_ptr(), p[:n][::-1], p[n], c, _lib.STEP_FIRST | _lib.STEP_LAST,
                         0, shape[0], stream=s)
    assert "k_resident<" in plan.last_kernel(), plan.last_kernel()
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0)), float((got - want).abs().nan_to_num().max())
    # ... and cut in two at an arbitrary level: states out, states in
    if n >= 10:
        a = n // 2 + 1
        u1, v1, got2 = torch.zeros_like(d), torch.zeros_like(d), torch.zeros_like(d)
        plan.resident_levels(None, None, u1.data_ptr(), v1.data_ptr(), d.data_ptr(), None, p[n - a: n][::-1], p[n], c, _lib.STEP_FIRST,
                             0, shape[0], stream=s)
        plan.resident_levels(u1.data_ptr(), v1.data_ptr(), None, None, d.data_ptr(), got2.data_ptr(), p[: n - a][::-1], p[n], c,
                             _lib.STEP_LAST, 0, shape[0], stream=s)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(got2, nan=-7.0), torch.nan_to_num(want, nan=-7.0))


@pytest.mark.parametrize("grid,shape,scale", [("REGULAR", (512, 512), 24.0), ("IRREGULAR_WITH_LAND", (256, 384), 12.0),
                                              ("REGULAR_WITH_LAND", (200, 300), 40.0), ("MOM5T", (128, 192), 70.0), ("MOM5U", (96, 160), 9.0),
                                              ("REGULAR_AREA_WEIGHTED", (128, 256), 24.0), ("REGULAR_WITH_LAND_AREA_WEIGHTED", (300, 400), 30.0),
                                              ("IRREGULAR_WITH_LAND", (512, 512), 16.0)])
def test_small_grids_run_the_whole_polynomial_in_one_launch(grid, shape, scale, monkeypatch):
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)    # the DEFAULT policy: whole grids of up to ~420 k cells run on the chip
    """north star: "the whole n_steps polynomial fused into a single launch" -- gcmf_apply does that for fields that fit on the chip
    (64 levels per launch).  Against the oracle, against the strip-marching path (GCMF_RESIDENT=0: same bits), NaN on land."""
    f, gv = T.scalar_case(grid, shape)
    if "wet_mask" in gv:
        f = np.where(gv["wet_mask"] == 0, np.nan, f)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    kw = dict(filter_scale=scale * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gv)
    flt = Filter(**kw)
    plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
    got = flt.apply(f)
    assert "k_resident<" in plan.last_kernel(), plan.last_kernel()
    fs = flt.filter_spec
    with np.errstate(all="ignore"):
        want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-12 * np.nanmax(np.abs(want))
    monkeypatch.setenv("GCMF_RESIDENT", "0")
    again = flt.apply(f)
    assert "k_resident<" not in plan.last_kernel()
    assert np.array_equal(got, again, equal_nan=True)
    # a tripolar grid (the seam is k_fold_band's job) and a grid beyond ~420 k cells stay on the strip-marching launches by default
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)


def test_default_policy_leaves_tripolar_and_larger_grids_on_the_strip_marching_launches(monkeypatch):
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)
    # (... and BASELINE config 1 -- REGULAR 512 x 512, 16 levels: two strip launches take 24-27 us, the on-chip launch 25-29 us; the cheap
    # REGULAR / land-mask levels go on the chip from 24 levels on)
    for grid, shape in (("TRIPOLAR_POP_WITH_LAND", (128, 192)), ("IRREGULAR_WITH_LAND", (720, 1440)), ("REGULAR", (512, 512))):
        f, gv = T.scalar_case(grid, shape)
        dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
        flt = Filter(filter_scale=12.0 * dx, dx_min=dx, grid_type=GridType[grid], grid_vars=gv, **({"n_steps": 16} if grid == "REGULAR" else {}))
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, shape)
        got = flt.apply(f)
        assert "k_ringc" in plan.last_kernel(), (grid, plan.last_kernel())
        fs = flt.filter_spec
        with np.errstate(all="ignore"):
            want = O.filter_func(O.FilterSpec(fs.n_steps, fs.s_max, np.asarray(fs.p), fs.dx_min_sq), grid, f, gv)
        assert np.nanmax(np.abs(got - want)) <= 1e-12 * np.nanmax(np.abs(want))


@pytest.mark.parametrize("grid,shape,halo,exchange", [("IRREGULAR_WITH_LAND", (300, 360), 32, "native"), ("IRREGULAR_WITH_LAND", (130, 132), 8, "p2p"),
                                                      ("REGULAR_WITH_LAND", (120, 128), 16, "p2p"), ("REGULAR", (96, 128), 12, "native"),
                                                      ("MOM5U", (96, 64), 10, "p2p")])
def test_slab_driver_runs_resident_between_exchanges(grid, shape, halo, exchange, monkeypatch):
    monkeypatch.setenv("GCMF_RESIDENT", "1")
    """A ring of one rank (ghost rows exchanged with itself: the 8-GPU choreography on one GPU): gcmf_slab_apply_backward runs every
    stretch between two exchanges as ONE resident launch; same bits as its strip-marching launches and as the single-domain filter."""
    from gcm_filters_amd.distributed import SlabFilter
    f, gv = T.scalar_case(grid, shape)
    if "wet_mask" in gv:
        f = np.where(gv["wet_mask"] == 0, np.nan, f)
    dx = T.grid_dx_min(grid, gv) if O.DIMENSIONAL[grid] else 1.0
    fk = dict(filter_scale=14.0 * dx, dx_min=dx, filter_shape="TAPER")
    sf = SlabFilter(grid, gv, fk, shape[0], shape[1], halo=halo, dtype=np.float64, device=0, rank=0, world=1, self_ring=True, exchange=exchange)
    assert sf.backward_cut and sf.native_driver
    got = sf.apply_local(sf.scatter_from_global([f[None]]))[0].cpu().numpy()
    assert "k_resident<" in sf.engine.plan.last_kernel(), sf.engine.plan.last_kernel()
    n_ex = sf.exchanges
    sf.resident = False
    again = sf.apply_local(sf.scatter_from_global([f[None]]))[0].cpu().numpy()
    assert "k_resident<" not in sf.engine.plan.last_kernel()
    assert np.array_equal(got, again, equal_nan=True)
    assert n_ex <= sf.exchanges - n_ex                                   # one exchange per `halo` levels: never more than the launches of 5..8 need
    one = Filter(filter_scale=fk["filter_scale"], dx_min=dx, filter_shape=FilterShape.TAPER, grid_type=GridType[grid], grid_vars=gv).apply(f)
    assert np.array_equal(np.isnan(got[0]), np.isnan(one))
    assert np.nanmax(np.abs(got[0] - one)) <= 1e-13 * np.nanmax(np.abs(one))


def _filter_cases():
    import make_golden as MG
    return [n for n in MG.case_names() if n != "REGULAR/config1" and "/lap" not in n]


@pytest.mark.parametrize("name", _filter_cases())
def test_reference_vectors_under_the_default_policy(name, golden_generated, monkeypatch):
    """The 93 vectors captured from the imported reference (tests/golden/reference_generated.npz) once more, with GCMF_RESIDENT unset -- the
    policy a user gets: the small scalar grids of these cases run on the chip wherever the policy says so (flux-form kinds with a polynomial
    that can be evaluated backwards; REGULAR / land-mask kinds from 24 levels on).  The rest of the GPU suite pins GCMF_RESIDENT=0 because it
    asserts which strip-marching kernel ran; here the results are held to the same gate and the kernel that ran is checked against the
    policy."""
    import make_golden as MG
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)
    grid, fields, gv, fk = MG.build_case(name)
    if fk is None:
        pytest.skip("a Laplacian vector, not a filter")
    want = golden_generated[name]
    flt = Filter(filter_scale=fk["filter_scale"], dx_min=fk["dx_min"], filter_shape=FilterShape[fk["filter_shape"]],
                 n_steps=fk.get("n_steps", 0), grid_type=GridType[grid], grid_vars=gv)
    vec = len(fields) == 2
    res = np.stack(flt.apply_to_vector(*fields)) if vec else flt.apply(fields[0])
    assert res.shape == want.shape and res.dtype == want.dtype
    all_f32 = all(f.dtype == np.float32 for f in fields) and all(v.dtype == np.float32 for v in gv.values())
    assert np.array_equal(np.isnan(res), np.isnan(want)), name
    ok = np.isfinite(want)
    err = float(np.abs(res[ok] - want[ok]).max() / np.abs(want[ok]).max())
    assert err <= (1e-4 if (all_f32 or name.endswith("/f32")) else 1e-11), (name, err)
    if not vec and fields[0].ndim == 2 and fields[0].dtype == np.float64 and all(v.dtype == np.float64 for v in gv.values()):
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, fields[0].shape)
        n = int(flt.n_steps)
        flux = grid in ("IRREGULAR_WITH_LAND", "MOM5U", "MOM5T")
        expect = bool(plan.clenshaw_cut(n)) and not grid.startswith("TRIPOLAR") and (flux or n >= 24)
        assert ("k_resident<" in plan.last_kernel()) == expect, (name, n, plan.last_kernel())


@pytest.mark.parametrize("grid", T.REFERENCE_TESTED_GRIDS)
def test_reference_zarr_goldens_under_the_default_policy(grid, golden_zarr, monkeypatch):
    """The reference's OWN nine filter goldens (upstream tests/test_data_filter/*.zarr, 128 x 256, Gaussian scale 8 = 9 levels) once more
    with GCMF_RESIDENT unset: what a user gets by default for grids of this size -- the flux-form kinds run their nine
    levels in one on-chip launch, the others the strip-marching / vector kernels."""
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)
    if grid in T.VECTOR_GRIDS:
        fields, gv = T.vector_case(grid)
    else:
        f, gv = T.scalar_case(grid)
        fields = (f,)
    flt = Filter(filter_scale=8.0, dx_min=1.0, filter_shape=FilterShape.GAUSSIAN, grid_type=GridType[grid], grid_vars=gv)
    res = np.stack(flt.apply_to_vector(*fields)) if len(fields) == 2 else flt.apply(fields[0])
    want = golden_zarr[f"test_data_filter/{grid}"]
    np.testing.assert_allclose(want, res.astype("f4"), rtol=2e-7, atol=1e-30)
    if grid == "IRREGULAR_WITH_LAND":
        plan = ALL_KERNELS[GridType[grid]](**gv)._plan(_lib.F64, fields[0].shape)
        assert "k_resident<" in plan.last_kernel(), plan.last_kernel()


def _run_two_workers(tmp_path, seconds, extra_env, nproc=2, extra_args=None):
    import json
    import subprocess
    import sys
    env = dict(os.environ, GCMF_RESIDENT_LOCK_DIR=str(tmp_path), **extra_env)
    env.pop("GCMF_RESIDENT", None)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resident_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(seconds), str(k)] + [str(x) for x in (extra_args[k] if extra_args else [])], env=env,
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(nproc)]
    try:
        for p in procs:
            assert p.stdout.readline().strip() == "READY", p.stderr.read()[-2000:]
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        outs = []
        for p in procs:
            out, err = p.communicate(timeout=180)
            assert p.returncode == 0, err[-2000:]
            outs.append(json.loads(out.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


def test_two_processes_share_a_gpu_without_clashing(tmp_path):
    """VERDICT r4 item 4 / ADVICE r4: k_resident is a persistent kernel with inter-workgroup waits; two PROCESSES running it on one GPU at
    the same time could each hold CUs the other waits for.  A process takes a per-GPU lock file before it runs resident kernels; the one
    that does not get it runs the strip-marching launches (same bits).  Two processes filter 512 x 512 grids side by side for 5 s: no NaN,
    no error, every result bit-equal to GCMF_RESIDENT=0."""
    outs = _run_two_workers(tmp_path, 5.0, {})
    for o in outs:
        assert o["n"] > 20 and o["wrong"] == 0 and o["nan_results"] == 0 and o["errors"] == [], outs
    assert any("gcmf::k_resident" in o["kernels"] for o in outs), outs          # somebody did run on the chip ...
    assert any(k.startswith("gcmf::k_ringc") for o in outs for k in o["kernels"]), outs   # ... and somebody stepped aside
    # ... and says so: Filter.last_path names the path, one RuntimeWarning per process names the reason
    stepped = [o for o in outs if "resident-lock-busy" in o["paths"]]
    assert stepped and all(len(o["lock_warnings"]) == 1 and "another process holds" in o["lock_warnings"][0] for o in stepped), outs
    assert all(o["path_counts"]["resident-lock-busy"] > 0 for o in stepped), outs
    assert any(o["paths"] == ["resident"] and o["lock_warnings"] == [] and o["status"]["state"] == "ok" for o in outs), outs


def test_an_idle_process_gives_the_lock_back(tmp_path):
    """a process that once filtered a small grid and then sits idle (a notebook) must not keep every other process of
    the GPU off the on-chip path until it exits: GCMF_RESIDENT_LOCK_IDLE_S seconds (default 5; 1 here) after its last on-chip launch has
    finished the lock is given back.  Process 0 filters for a second and then idles for six; process 1 starts three seconds in and must find
    the on-chip path free."""
    outs = _run_two_workers(tmp_path, 1.0, {"GCMF_RESIDENT_LOCK_IDLE_S": "1"}, extra_args=[[0.0, 6.0], [3.0, 0.0]])
    assert outs[0]["paths"] == ["resident"] and outs[1]["paths"] == ["resident"], outs
    assert outs[0]["status"]["state"] == "off" and outs[1]["status"]["state"] == "ok", outs     # (0 gave it back; 1 holds it)
    for o in outs:
        assert o["wrong"] == 0 and o["nan_results"] == 0 and o["errors"] == [] and o["lock_warnings"] == [], outs


def test_the_lock_file_is_not_followed_through_a_symlink(tmp_path):
    """the lock file's name is predictable and its directory shared, so somebody may have put a symbolic link there
    that points at a file of the victim's.  libgcmf opens it O_NOFOLLOW and sets a mode only on a file it created itself: the target keeps
    its mode and its content, and the process still filters (it falls back to the next directory, or to no lock at all)."""
    import glob
    import stat
    victim = tmp_path / "victim.txt"
    victim.write_text("precious")
    os.chmod(victim, 0o600)
    lockdir = tmp_path / "locks"
    lockdir.mkdir()
    # the name depends on the PCI bus id: let one run create it, then replace it by a link and run again
    (o,) = _run_two_workers(lockdir, 0.3, {}, nproc=1)
    names = glob.glob(str(lockdir / "gcmf_resident_*.lock"))
    assert len(names) == 1 and o["errors"] == [], (names, o)
    os.remove(names[0])
    os.symlink(victim, names[0])
    (o,) = _run_two_workers(lockdir, 0.3, {}, nproc=1)
    assert o["n"] > 0 and o["wrong"] == 0 and o["errors"] == [], o
    assert stat.S_IMODE(os.stat(victim).st_mode) == 0o600 and victim.read_text() == "precious"
    assert os.path.islink(names[0])                                       # (left alone)


def test_last_path_and_counters(monkeypatch):
    """Filter.last_path / Plan.path_counts (include/gcmf.h: gcmf_plan_last_path): which of the two bit-identical paths ran."""
    shape = (256, 256)
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    flt = Filter(filter_scale=8.0 * dx, dx_min=dx, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    assert flt.last_path is None
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)
    a = flt.apply(f)
    assert flt.last_path == "resident"
    monkeypatch.setenv("GCMF_RESIDENT", "0")
    b = flt.apply(f)
    assert flt.last_path == "strips" and np.array_equal(a, b, equal_nan=True)
    plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, shape)
    counts = plan.path_counts()
    assert counts["resident"] >= 1 and counts["strips"] >= 1 and counts["resident-lock-busy"] == 0
    assert _lib.resident_status(0)["state"] in ("ok", "off") and _lib.resident_status(0)["failures"] == 0
    big = Filter(filter_scale=4.0, dx_min=1.0, grid_type=GridType.REGULAR)      # too few levels for the on-chip policy: strips
    monkeypatch.delenv("GCMF_RESIDENT", raising=False)
    big.apply(T.random_field((64, 64), 3))
    assert big.last_path == "strips"


def test_a_clash_outside_the_lock_is_loud_and_then_falls_back(tmp_path):
    """The lock switched off (two containers that do not share /dev/shm): a clash ends in the bounded time-out -- the result of that
    application is NaN everywhere (never plausible-but-wrong), the next call reports it ONCE, and the process runs the strip-marching
    launches from then on instead of failing for ever."""
    outs = _run_two_workers(tmp_path, 4.0, {"GCMF_RESIDENT_LOCK": "0", "GCMF_RESIDENT_TIMEOUT_MS": "100"})
    for o in outs:
        assert o["wrong"] == 0, outs                       # right or NaN, nothing in between
        assert len(o["errors"]) <= 1, outs                 # told once -- to the plan whose application it poisoned
        assert (o["status"]["failures"] > 0) == (o["status"]["state"] == "disabled"), outs
        if o["errors"]:
            assert o["status"]["state"] == "disabled" and "resident-disabled" in o["paths"], outs
        assert o["n"] > 10, outs                           # and the work went on
        if o["errors"] or o["nan_results"]:
            assert any(k.startswith("gcmf::k_ringc") for k in o["kernels"]), outs


def test_a_process_with_a_cu_mask_stays_off_the_resident_kernel(tmp_path):
    """with HSA_CU_MASK / ROC_GLOBAL_CU_MASK the runtime still reports every compute unit while fewer can run workgroups, so
    "one workgroup per CU, all resident" cannot be promised: such a process runs the strip-marching launches (same bits).  The mask used
    here names every CU of the device (no actual restriction): only the decision is under test."""
    (o,) = _run_two_workers(tmp_path, 1.0, {"HSA_CU_MASK": "0:0-255"}, nproc=1)
    assert o["n"] > 5 and o["wrong"] == 0 and o["nan_results"] == 0 and o["errors"] == [], o
    assert o["kernels"] and all(k.startswith("gcmf::k_ringc") or k.startswith("gcmf::k_land_fix") for k in o["kernels"]), o


# ---- k_ringc_one (csrc/gcmf_ringc_one.hip): the whole polynomial of a BASELINE-size grid in ONE persistent launch (opt-in) ----------------
@pytest.mark.parametrize("shape,n_steps", [((128, 256), 18), ((300, 520), 27), ((260, 1100), 16), ((700, 1100), 63), ((2400, 3600), 63), ((2400, 3600), 56)])
def test_single_launch_gives_the_bits_of_the_back_to_back_launches(shape, n_steps, monkeypatch):
    """The north star's literal form, "the whole n_steps polynomial fused into a single launch", for whole f64 flux-form grids that do NOT
    fit the chip: the same strip marches, the passes separated by grid-wide barriers inside one persistent launch instead of by launch
    boundaries.  Opt-in (plan option "single_launch" / GCMF_SINGLE_LAUNCH=1: it measures 7 % slower at 2400 x 3600); bit-identical."""
    import warnings
    monkeypatch.setenv("GCMF_RESIDENT", "0")          # (not the on-chip tile kernel: this is about grids of any size)
    f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
    f = np.where(gv["wet_mask"] == 0, np.nan, f)
    dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flt = Filter(filter_scale=6 * dx, dx_min=dx, n_steps=n_steps, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
    plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, shape)
    try:
        plan.set_option("single_launch", 0)
        plan.last_kernel()
        want = flt.apply(f)
        assert "k_ringc" in plan.last_kernel()          # (k_ringc, or its early-exit form k_ringcs on short strips)
        plan.set_option("single_launch", 1)
        got = flt.apply(f)
        assert "k_ringc_one<" in plan.last_kernel(), plan.last_kernel()
        plan.set_timing(True)
        flt.apply(f)
        assert plan.last_timing()[1] == 1               # ONE recurrence launch (k_land_fix, the isolated cells, is not counted)
    finally:
        plan.set_timing(False)
        plan.set_option("single_launch", 0)
    assert np.array_equal(got, want, equal_nan=True)
    assert _lib.resident_status(0)["failures"] == 0


def test_single_launch_times_out_loudly(tmp_path):
    """A workgroup that never arrives at a barrier (debug switch): every wait is bounded, the result of that application is NaN
    everywhere, the NEXT call of the plan says so once, and the process goes on with the back-to-back launches."""
    import json
    import subprocess
    import sys
    code = r"""
import json, os, sys, warnings
sys.path.insert(0, %r)
import numpy as np
from gcm_filters_amd import Filter, FilterShape, GridType, _lib, testing as T
from gcm_filters_amd.kernels import ALL_KERNELS
shape = (300, 520)
f, gv = T.scalar_case("IRREGULAR_WITH_LAND", shape)
dx = T.grid_dx_min("IRREGULAR_WITH_LAND", gv)
warnings.simplefilter("ignore")
flt = Filter(filter_scale=6 * dx, dx_min=dx, n_steps=27, filter_shape=FilterShape.TAPER, grid_type=GridType.IRREGULAR_WITH_LAND, grid_vars=gv)
plan = ALL_KERNELS[GridType.IRREGULAR_WITH_LAND](**gv)._plan(_lib.F64, shape)
want = flt.apply(f)
plan.set_option("single_launch", 2)
import torch
d = torch.from_numpy(f).cuda()
bad = flt.apply(d).cpu().numpy()          # device pointers: the call returns before the kernel has given up
errors = []
for _ in range(3):
    try:
        plan.last_kernel()
        again = flt.apply(d).cpu().numpy()
    except Exception as e:
        errors.append(str(e)[:120])
wet = gv["wet_mask"] == 1                  # (land cells get their own polynomial from k_land_fix afterwards)
print(json.dumps({"all_nan": bool(np.isnan(bad[wet]).all()), "errors": errors, "recovered": bool(np.array_equal(again, want)),
                  "kernel": plan.last_kernel(), "status": _lib.resident_status(0)}))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GCMF_RESIDENT="0", GCMF_RESIDENT_TIMEOUT_MS="50", GCMF_RESIDENT_LOCK_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    o = json.loads(out.stdout.strip().splitlines()[-1])
    assert o["all_nan"] and len(o["errors"]) == 1 and "timed out" in o["errors"][0], o
    assert o["recovered"] and o["kernel"].startswith("gcmf::k_ringc") and o["status"] == {"state": "disabled", "failures": 1}, o
