"""Build libgcmf.so (HIP, gfx950 only) in-tree with hipcc.  No torch, no cmake: the translation units compile in parallel."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libgcmf.so")
SOURCES = ["gcmf_api.hip", "gcmf_precompute.hip", "gcmf_scalar.hip", "gcmf_scalar_multi.hip", "gcmf_scalar_multi_reg.hip", "gcmf_scalar_multi_mask.hip", "gcmf_scalar_multi_maskz.hip", "gcmf_scalar_multi_flux.hip", "gcmf_flux_multi2.hip", "gcmf_vector.hip", "gcmf_cgrid_stream.hip", "gcmf_cgrid_stream2.hip", "gcmf_bgrid_stream.hip", "gcmf_bgrid_stream2.hip", "gcmf_landfix.hip", "gcmf_exchange.hip", "gcmf_ring_flux.hip", "gcmf_ring_maskz.hip", "gcmf_ring_reg.hip", "gcmf_ringc_flux.hip", "gcmf_ringc_maskz.hip", "gcmf_ringc_reg.hip"]
HEADERS = [os.path.join(CSRC, "gcmf_internal.hpp"), os.path.join(CSRC, "gcmf_multi_common.hpp"), os.path.join(CSRC, "gcmf_scalar_multi_impl.hpp"), os.path.join(CSRC, "gcmf_recurrence.hpp"), os.path.join(CSRC, "gcmf_flux_multi2_body.hpp"), os.path.join(CSRC, "gcmf_ring_impl.hpp"), os.path.join(CSRC, "gcmf_ringc_impl.hpp"), os.path.join(INCLUDE, "gcmf.h")]
# -ffp-contract=off: no FMA contraction, so the REGULAR / land-mask / B-grid kernels reproduce the
# reference's (numpy's) rounding exactly; the kernels are HBM-bound, the extra VALU ops are free.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC]


FLAGS += os.environ.get("GCMF_EXTRA_HIPCC_FLAGS", "").split()  # experiments (A/B builds on the GPU box)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgcmf cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source to an object (in parallel) and link csrc/libgcmf.so.  Returns its path."""
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc(), *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"--- {src} ---\n{out}")
        elif verbose and out.strip():
            print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
