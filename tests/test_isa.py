"""ISA-level guards (tools/check_isa.py) in the CPU suite: the gfx950 code inside the in-tree libgcmf.so keeps the properties the
kernels' performance and -- twice already (compiler notes) -- their correctness rest on.  No GPU needed: the code
objects are read out of the library's offload bundles with llvm-readelf / llvm-objdump."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import check_isa  # noqa: E402

needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(check_isa.LLVM, "llvm-readelf")), reason="no ROCm llvm tools")


def test_vmcnt_walker_sees_a_read_before_its_wait():
    """The checker itself: the vector-memory counter retires in order, `vmcnt(N)` leaves the youngest N outstanding."""
    ok = """
        global_load_dwordx4 v[10:13], v[2:3], off
        global_load_dwordx2 a[4:5], v[2:3], off offset:16
        v_add_f64 v[20:21], v[30:31], v[32:33]
        s_waitcnt vmcnt(1)
        v_add_f64 v[20:21], v[10:11], v[12:13]
        s_waitcnt vmcnt(0)
        v_accvgpr_read_b32 v40, a4
    """.splitlines()
    assert check_isa.walk_vmcnt(ok) == ([], 2, 2)
    early = """
        global_load_dwordx4 v[10:13], v[2:3], off
        global_load_dwordx2 a[4:5], v[2:3], off offset:16
        s_waitcnt vmcnt(1)
        v_accvgpr_read_b32 v40, a5
    """.splitlines()
    bad, nl, nw = check_isa.walk_vmcnt(early)
    assert len(bad) == 1 and "a5" in bad[0] and nl == 2
    overwritten = """
        global_load_dwordx2 v[6:7], v[2:3], off
        v_mov_b32 v7, 0
    """.splitlines()
    assert len(check_isa.walk_vmcnt(overwritten)[0]) == 1
    stores_count = """
        global_load_dwordx2 v[6:7], v[2:3], off
        global_store_dwordx2 v[2:3], v[8:9], off
        s_waitcnt vmcnt(1)
        v_add_f64 v[20:21], v[6:7], v[6:7]
    """.splitlines()
    assert check_isa.walk_vmcnt(stores_count)[0] == []      # the store is the younger one: the load has retired


@needs_llvm
def test_no_scratch_and_register_budgets_of_the_marching_kernels(capsys):
    """k_ring / k_ringc / k_fold_band: no scratch, one wave per SIMD fits, and k_fold_band (<= 48 registers) fits NEXT to every
    flux-kind k_ringc wave -- the co-residency config 4's rate depends on (gcmf_foldband.hip)."""
    old = sys.argv
    sys.argv = ["check_isa.py"]
    try:
        rc = check_isa.main()
    finally:
        sys.argv = old
    out = capsys.readouterr().out
    assert rc == 0, out


@needs_llvm
def test_vmcnt_discipline_of_the_default_kernels(capsys):
    """Every global load of the kernels the BASELINE configs run is waited for before its destination is touched."""
    for pat in ("k_ringc<double, 2, 8", "k_ring<double, double, 5, 8", "k_fold_band<double, double, 2"):
        old = sys.argv
        sys.argv = ["check_isa.py", "--waitcnt-only", pat]
        try:
            rc = check_isa.main()
        finally:
            sys.argv = old
        out = capsys.readouterr().out
        assert rc == 0 and "vmcnt discipline walked for 0 kernels" not in out, out
