"""Build libgcmf.so (HIP, gfx950 only) in-tree with hipcc.  No torch, no cmake: the translation units compile in parallel."""
from __future__ import annotations

import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libgcmf.so")
SOURCES = ["gcmf_api.hip", "gcmf_api_blocks.hip", "gcmf_api_options.hip", "gcmf_precompute.hip", "gcmf_scalar.hip", "gcmf_scalar_multi.hip", "gcmf_scalar_multi_reg.hip", "gcmf_scalar_multi_mask.hip", "gcmf_scalar_multi_maskz.hip", "gcmf_scalar_multi_flux.hip", "gcmf_flux_multi2.hip", "gcmf_vector.hip", "gcmf_cgrid_stream.hip", "gcmf_cgrid_stream2.hip", "gcmf_cgrid_ring.hip", "gcmf_cgrid_ringf.hip", "gcmf_bgrid_stream.hip", "gcmf_bgrid_stream2.hip", "gcmf_landfix.hip", "gcmf_exchange.hip", "gcmf_ring_flux.hip", "gcmf_ring_flux_f32.hip", "gcmf_ring_maskz.hip", "gcmf_ring_reg.hip", "gcmf_ringc_flux.hip", "gcmf_ringc_flux_f32.hip", "gcmf_ringc_flux9.hip", "gcmf_ringc_one.hip", "gcmf_ringc_zip.hip", "gcmf_ringc_zip_b.hip", "gcmf_ringc_zip_c.hip", "gcmf_ringc_zip_d.hip", "gcmf_ringc_maskz.hip", "gcmf_ringc_maskz_f32.hip", "gcmf_ringc_reg.hip", "gcmf_ringc_reg_f32.hip", "gcmf_foldband.hip", "gcmf_p2p.hip", "gcmf_ringc_flux_slab.hip", "gcmf_ringc_flux_slab_b.hip", "gcmf_ringc_flux_slab_f32.hip", "gcmf_ringc_flux_slab_f32b.hip", "gcmf_resident.hip"]
BUILD_ID_SOURCE = "gcmf_buildid.hip"   # compiled on every link with -DGCMF_BUILD_ID=<source_build_id()>
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INCLUDE, "gcmf.h")]
# -ffp-contract=off: no FMA contraction, so the REGULAR / land-mask / B-grid kernels reproduce the
# reference's (numpy's) rounding exactly; the kernels are HBM-bound, the extra VALU ops are free.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC]


FLAGS += os.environ.get("GCMF_EXTRA_HIPCC_FLAGS", "").split()  # experiments (A/B builds on the GPU box)


_ID_MARK = b"GCMF_BUILD_ID="


def source_build_id() -> str:
    """sha256 over every file under csrc/ that is source (*.hip, *.hpp, *.h), include/gcmf.h and the compiler flags:
    the identity of the sources a binary has to come from.  libgcmf.so carries the value it was built with
    (gcmf_build_id(), and a marker string in its data segment that binary_build_id() reads without dlopen)."""
    h = hashlib.sha256()
    files = sorted(p for ext in ("*.hip", "*.hpp", "*.h") for p in glob.glob(os.path.join(CSRC, ext)))
    files.append(os.path.join(INCLUDE, "gcmf.h"))
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(a for a in FLAGS if a not in (INCLUDE, CSRC)).encode())
    return h.hexdigest()


def binary_build_id(path: str = None):
    """The build id baked into a libgcmf.so (None when the file is missing or carries none)."""
    path = path or LIB
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(re.escape(_ID_MARK) + rb"([0-9a-f]{64})", blob)
    return m.group(1).decode() if m else None


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
    """Compile every .hip source to an object (in parallel) and link csrc/libgcmf.so.  Returns its path.

    Serialised between processes by a lock file (the ranks of a torchrun / `bench.py --gpus N` job all find a stale binary at
    once): whoever gets the lock builds, the others wait and then find the binary fresh.  The link goes to a temporary name and
    is moved into place, so no process ever dlopens a half-written library."""
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and binary_build_id() == source_build_id() and not any(
                    _stale(os.path.join(CSRC, s.replace(".hip", ".o")), [os.path.join(CSRC, s)] + HEADERS) for s in SOURCES):
                return LIB          # another process built it while this one waited for the lock
            return _build_library_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_library_locked(force: bool, verbose: bool) -> str:
    objs, procs = [], []
    want_id = source_build_id()
    if not force and os.path.exists(LIB) and binary_build_id() != want_id and not any(
            _stale(os.path.join(CSRC, s.replace(".hip", ".o")), [os.path.join(CSRC, s)] + HEADERS) for s in SOURCES):
        force = True    # the objects look fresh by mtime but the binary was made from other sources: trust the hash
    todo = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            todo.append((src, [hipcc(), *FLAGS, "-c", s, "-o", o]))
    # longest first, one compiler per core: the static-ring units (dozens of 400-register kernels each) take two to three minutes of one
    # core, the others seconds -- started all at once they share the cores with everything else until the end
    heavy = ("gcmf_ringc", "gcmf_ring_", "gcmf_cgrid_ring", "gcmf_resident", "gcmf_scalar_multi", "gcmf_bgrid_stream2", "gcmf_cgrid_stream2")
    todo.sort(key=lambda t: (0 if t[0].startswith(heavy) else 1, t[0]))
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(job):
        src, cmd = job
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return src, r.returncode, r.stdout
    procs = todo
    failed = []
    with ThreadPoolExecutor(max_workers=max(2, os.cpu_count() or 2)) as pool:
        for src, rc, out in pool.map(compile_one, todo):
            if rc != 0:
                failed.append(f"--- {src} ---\n{out}")
            elif verbose and out.strip():
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    if force or procs or _stale(LIB, objs) or binary_build_id() != want_id:
        ido = os.path.join(CSRC, BUILD_ID_SOURCE.replace(".hip", ".o"))
        r = subprocess.run([hipcc(), *FLAGS, f'-DGCMF_BUILD_ID="{want_id}"', "-c", os.path.join(CSRC, BUILD_ID_SOURCE), "-o", ido],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        tmp = os.path.join(CSRC, f"libgcmf.tmp{os.getpid()}.so")
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs, ido]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        for junk in glob.glob(tmp + ".*") + glob.glob(LIB + ".*"):      # offload-bundler by-products of the link
            try:
                os.remove(junk)
            except FileNotFoundError:
                pass
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n" + r.stdout)
        if binary_build_id(tmp) != want_id:
            os.remove(tmp)
            raise RuntimeError("libgcmf.so does not carry the build id it was linked with")
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
