#!/usr/bin/env python3
"""Fold a profile run (tools/profile_gpu.sh -> gpurun_out/prof_<tag>_cfg<N>/summary.json, optionally the SQ counter summary of
tools/sq_counters.sh) into profiles/hbm_traffic.json, the table bench.py reads `roofline.traffic` from.

    python tools/update_hbm_traffic.py <round> <config> <prof summary.json> <kernel substring> [<sq summary.txt>] [<dest name in profiles/>]

bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 64 B per 128-B request on gfx950 for the
16-byte-per-lane loads these kernels issue (MI355X_MICROARCH.md, calibrated in profiles/README.md)."""
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, cfg, summ, ksub = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
sq = sys.argv[5] if len(sys.argv) > 5 else None
s = json.load(open(summ))
kern = [k for k in s["pmc"] if ksub in k and "FETCH_SIZE" in s["pmc"][k] and "WRITE_SIZE" in s["pmc"][k]]
assert len(kern) == 1, kern
kern = kern[0]
p = s["pmc"][kern]
stat = [r for r in s.get("kernel_stats", []) if r["Name"].startswith(kern)]
dest_dir = os.path.join(REPO, "profiles", rnd)
os.makedirs(dest_dir, exist_ok=True)
base = f"cfg{cfg}_{re.sub(r'[^a-z0-9]+', '_', ksub.lower()).strip('_')}"
shutil.copy(os.path.join(os.path.dirname(summ), "summary.txt"), os.path.join(dest_dir, base + "_summary.txt"))
rec = {
    "round": rnd, "kernel": kern, "FETCH_SIZE_KB": p["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB": p["WRITE_SIZE"]["mean"],
    "bytes_per_launch": (2 * p["FETCH_SIZE"]["mean"] + p["WRITE_SIZE"]["mean"]) * 1024,
    "avg_launch_us_rocprof": float(stat[0]["AverageNs"]) / 1e3 if stat else None,
    "l2_hit_rate": (p["TCC_HIT_sum"]["mean"] / (p["TCC_HIT_sum"]["mean"] + p["TCC_MISS_sum"]["mean"])) if "TCC_HIT_sum" in p else None,
    "source": f"profiles/{rnd}/{base}_summary.txt",
}
if s.get("bench_kernel") and s["bench_kernel"] in kern.replace("void ", ""):
    rec["geometry"] = s["bench_geometry"]   # launch geometry at profile time: bench.py quotes the bytes only for the same one
else:
    print(f"WARNING: the traced bench line names {s.get('bench_kernel')!r}, not {kern!r}: no geometry recorded, bench.py will withhold the traffic")
if sq:
    txt = open(sq).read()
    blk = txt[txt.index(kern):]
    vals = {}
    for line in blk.split("\n")[1:]:
        m = re.match(r"\s+(\w+): n=\d+ mean=([0-9.e+-]+)", line)
        if not m:
            break
        vals[m.group(1)] = float(m.group(2))
    shutil.copy(sq, os.path.join(dest_dir, base + "_sq_counters.txt"))
    wc = vals["SQ_WAVE_CYCLES"]
    arith = sum(vals.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                         "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
    rec.update({"bound": "hbm: streams at the rate this access pattern reaches on the chip (experiments/stream_probe: 5.0-5.5 TB/s without halos or arithmetic); removing 20 % of its VALU work or adding rows in flight changes nothing", "valu_active_frac": vals["SQ_ACTIVE_INST_VALU"] / wc,
                "salu_active_frac": vals.get("SQ_ACTIVE_INST_SCA", 0) / wc, "wait_memory_frac": vals["SQ_WAIT_ANY"] / wc,
                "wait_issue_frac": vals["SQ_WAIT_INST_ANY"] / wc, "valu_arith_share": arith / vals["SQ_INSTS_VALU"],
                "valu_insts_per_launch": vals["SQ_INSTS_VALU"], "waves": vals["SQ_WAVES"],
                "counters_source": f"profiles/{rnd}/{base}_sq_counters.txt"})
tf = os.path.join(REPO, "profiles", "hbm_traffic.json")
tab = json.load(open(tf))
old = tab.get(f"config{cfg}")
if old and old.get("round") != rnd:
    tab[f"config{cfg}_{old.get('round', 'old')}"] = old
tab[f"config{cfg}"] = rec
json.dump(tab, open(tf, "w"), indent=1)
print(json.dumps(rec, indent=1))
