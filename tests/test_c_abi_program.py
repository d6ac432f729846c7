"""The C ABI is usable from plain C: compile tests/c_abi/cabi_smoke.c with gcc against include/gcmf.h and
gcm_filters_amd/csrc/libgcmf.so.  Without a GPU it must fail loudly with GCMF_ERR_NO_DEVICE (exit code 3);
on the MI355X it must reproduce an inline C restatement of the filter bit for bit (-ffp-contract=off both sides)."""
import os
import subprocess

import pytest

from gcm_filters_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    _lib.load()
    exe = str(tmp_path / "cabi_smoke")
    libdir = os.path.join(REPO, "gcm_filters_amd", "csrc")
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "c_abi", "cabi_smoke.c"),
           "-o", exe, "-L", libdir, "-lgcmf", f"-Wl,-rpath,{libdir}", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_c_program_links_and_fails_loudly_without_gpu(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == _lib.ERR_NO_DEVICE, (r.returncode, r.stdout, r.stderr)
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_program_matches_inline_reference(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "0 of 3072 cells differ" in r.stdout
