#!/usr/bin/env python3
"""ISA-level checks of the gfx950 code inside libgcmf.so: what the test suite cannot see but the kernels depend on.

    python tools/check_isa.py [--lib PATH] [--waitcnt] [-v]

Reads the gfx950 code objects out of the library's offload bundles (no GPU needed; llvm-readelf / llvm-objdump from /opt/rocm) and checks

  1. no scratch: the strip-marching kernels (k_ring, k_ringc), k_fold_band and the C-/B-grid streaming kernels keep everything in
     registers / LDS (private_segment_fixed_size == 0); a hipcc upgrade that starts spilling would cost bandwidth silently;
  2. register budgets: k_fold_band <= 48 registers and every k_ringc instantiation of the flux kinds <= 512 - 48 - 8, so that the
     band's waves fit on the SIMDs NEXT to the blocked launch of a tripolar plan (gcmf_foldband.hip; config 4 loses 10 % otherwise);
     one wave per SIMD budget (<= 512) for all k_ring / k_ringc;
  3. (--waitcnt) memory-wait discipline of k_ring / k_ringc / k_fold_band: walking every basic block with the hardware's in-order
     vmcnt counter, no instruction may read (or overwrite) the destination of a global load that an `s_waitcnt vmcnt(N)` has not
     yet retired.  Two compiler mis-schedules of exactly this kind were met in round 2 and caught only by chance.

Exit status 0 = all checks pass.  `python -m pytest tests/test_isa.py` runs 1 + 2 (and 3 on the default flux kernels) in the CPU suite.
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "gcm_filters_amd", "csrc", "libgcmf.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BAND_BUDGET = 48            # registers a k_fold_band wave may use
RING_FLUX_BUDGET = 512 - BAND_BUDGET - 8   # what a flux-kind k_ringc wave may allocate next to it (allocation granule 8)


def code_objects(lib_path):
    """The gfx950 ELF images of every offload bundle in the library (one per translation unit)."""
    blob = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24: pos + 24 + tl].decode()
            pos += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[base + off: base + off + size])
    return out


def kernel_metadata(elf_bytes, tmpdir, idx):
    """[{name, vgpr, agpr, sgpr, scratch, lds}] from the AMDGPU metadata note of one code object."""
    path = os.path.join(tmpdir, f"co{idx}.elf")
    open(path, "wb").write(elf_bytes)
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
    import yaml
    body = txt[txt.index("---"):]
    if "\n..." in body:
        body = body[: body.index("\n...")]
    meta = yaml.safe_load(body) or {}
    res = []
    for k in meta.get("amdhsa.kernels", []):
        res.append(dict(name=k[".name"], vgpr=int(k[".vgpr_count"]), agpr=int(k.get(".agpr_count", 0)), sgpr=int(k.get(".sgpr_count", 0)),
                        scratch=int(k.get(".private_segment_fixed_size", 0)), lds=int(k.get(".group_segment_fixed_size", 0)), path=path))
    return res


def demangle(names):
    r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],
                       input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines()
    return out if len(out) == len(names) else names


# ------------------------------------------------------------------------------------------------------------------------------
# vmcnt discipline
# ------------------------------------------------------------------------------------------------------------------------------
_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def check_waitcnt(path, mangled):
    """Walk the kernel's disassembly.  The vector-memory counter retires loads and stores IN ORDER; `s_waitcnt vmcnt(N)` returns
    when at most N are outstanding.  Returns the list of violations: instructions that touch the destination registers of a load
    still outstanding.  State is dropped at labels (a join of paths whose outstanding sets are unknown) -- the unrolled march
    bodies these kernels spend their time in are straight-line code thousands of instructions long."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={mangled}", path],
                         capture_output=True, text=True).stdout
    return walk_vmcnt(txt.splitlines())


def walk_vmcnt(lines):
    """(violations, loads seen, vmcnt waits seen) of a disassembly listing; see check_waitcnt."""
    outstanding = []   # [(dest regs or empty set, text)]
    bad, n_loads, n_waits = [], 0, 0
    for line in lines:
        s = line.strip()
        if not s or s.startswith(("Disassembly", "/")) or "file format" in s:
            continue
        if re.match(r"^[0-9a-f]+ <.*>:$", s) or s.endswith(":"):
            outstanding = []
            continue
        s = s.split("//")[0].strip()
        if not s:
            continue
        op, _, rest = s.partition(" ")
        if op.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                n_waits += 1
                keep = int(m.group(1))
                outstanding = outstanding[len(outstanding) - keep:] if keep else []
            elif "vmcnt" not in rest and re.fullmatch(r"\s*(0x[0-9a-f]+|\d+)\s*", rest or ""):   # raw immediate: assume it waits for all
                outstanding = []
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            outstanding = []
            continue
        touched = _regs(rest)
        if outstanding and touched:
            for dest, what in outstanding:
                hit = dest & touched
                if hit:
                    bad.append(f"{s}   <- touches {sorted(hit)[:4]} of outstanding `{what}`")
                    break
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            n_loads += 1
            first = rest.split(",")[0]
            outstanding.append((_regs(first), s[:60]))
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
            outstanding.append((set(), s[:60]))
    return bad, n_loads, n_waits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--waitcnt", action="store_true", help="also walk the vmcnt discipline of every k_ring / k_ringc / k_fold_band kernel (slow)")
    ap.add_argument("--waitcnt-only", default="", help="substring of the demangled kernel names to walk (implies --waitcnt)")
    ap.add_argument("-v", action="store_true")
    args = ap.parse_args()
    failures = []
    with tempfile.TemporaryDirectory() as tmp:
        kernels = []
        for i, co in enumerate(code_objects(args.lib)):
            kernels += kernel_metadata(co, tmp, i)
        if not kernels:
            print("no gfx950 kernels found in", args.lib)
            return 2
        names = demangle([k["name"] for k in kernels])
        for k, d in zip(kernels, names):
            k["dname"] = d.replace("void ", "").split("(")[0]
        watched = [k for k in kernels if re.search(r"gcmf::k_(ring|ringc|ringcs|ringcp|ringcz|ringc_one|fold_band|resident|cgrid_stream2c?|cgrid_ringf?|bgrid_stream2c?)<", k["dname"])]
        for k in sorted(watched, key=lambda k: k["dname"]):
            total = k["vgpr"]   # gfx90a and later: .vgpr_count is the unified total (architected + accumulation registers)
            alloc = (total + 7) // 8 * 8
            if args.v:
                print(f"{k['dname']:70s} registers {k['vgpr']:3d} (of them accumulation {k['agpr']:3d}) allocated {alloc:3d} scratch {k['scratch']:4d} lds {k['lds']}")
            if k["scratch"] and not re.search(r"k_[cb]grid_stream2|k_resident<", k["dname"]):
                failures.append(f"{k['dname']}: {k['scratch']} bytes of scratch per lane")
            # k_resident's deepest instantiations park a dozen loop-invariant index words (the flat band / halo list bases) in scratch: written
            # once before the level loop, read back in the tile exchange, nothing of it inside a level (checked in the ISA, round 4)
            if "k_resident<" in k["dname"] and k["scratch"] > 64:
                failures.append(f"{k['dname']}: {k['scratch']} bytes of scratch per lane (> 64: the level loop is probably spilling)")
            if "k_fold_band<" in k["dname"] and alloc > BAND_BUDGET:
                failures.append(f"{k['dname']}: {alloc} registers > {BAND_BUDGET}: its waves no longer fit beside a k_ringc wave")
            if re.search(r"k_ringcp?<(double|float), 2,", k["dname"]) and not k["dname"].rstrip(">").endswith(", true") and alloc > RING_FLUX_BUDGET:   # (k_ringcp<..., XE = true>: slabs without a seam)
                failures.append(f"{k['dname']}: {alloc} registers > {RING_FLUX_BUDGET}: no room for k_fold_band's waves on its SIMD (tripolar plans)")
            if re.search(r"k_ringc?[spz]?<", k["dname"]) and alloc > 512:
                failures.append(f"{k['dname']}: {alloc} registers > 512")
            # the on-chip kernel: 512 threads = two waves per SIMD, so 256 registers and not a byte of scratch (its cells LIVE in registers)
            if "k_resident<" in k["dname"] and alloc > 256:
                failures.append(f"{k['dname']}: {alloc} registers > 256: a 512-thread workgroup no longer fits a CU")
        n_ring = sum(1 for k in watched if re.search(r"k_ringc?[spz]?<", k["dname"]))
        n_band = sum(1 for k in watched if "k_fold_band<" in k["dname"])
        n_res = sum(1 for k in watched if "k_resident<" in k["dname"])
        if n_res < 11:
            failures.append(f"{n_res} k_resident instantiations found, 11 expected (three kinds x cells per thread)")
        print(f"{len(kernels)} gfx950 kernels, {n_ring} k_ring / k_ringc and {n_band} k_fold_band instantiations checked for scratch and register budgets")
        if n_ring < 20 or n_band < 6:
            failures.append("fewer ring / band kernels found than the library instantiates: the metadata parser is out of date")
        if args.waitcnt or args.waitcnt_only:
            todo = [k for k in watched if re.search(r"k_(ring|ringc|ringcs|fold_band)<", k["dname"]) and args.waitcnt_only in k["dname"]]
            for k in todo:
                bad, nl, nw = check_waitcnt(k["path"], k["name"])
                if args.v or bad:
                    print(f"{k['dname']}: {nl} loads, {nw} vmcnt waits, {len(bad)} violations")
                if nl == 0:
                    failures.append(f"{k['dname']}: no loads found in the disassembly (parser out of date?)")
                for b in bad[:5]:
                    failures.append(f"{k['dname']}: {b}")
            print(f"vmcnt discipline walked for {len(todo)} kernels")
    for f in failures:
        print("FAIL:", f)
    print("check_isa:", "FAILED" if failures else "ok")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
