#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats CSV + PMC counter CSVs) into small text/JSON summaries."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
summary = {"tag": tag, "config": int(cfg)}
lines = []

stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines.append(f"== rocprofv3 --kernel-trace --stats ({os.path.basename(stats[0])}) ==")
    for r in rows[:12]:
        lines.append("  ".join(f"{k}={r[k]}" for k in r))
    summary["kernel_stats"] = rows[:12]


def pmc(sub, counters):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            c = r.get("Counter_Name")
            if c in counters:
                acc[name][c].append(float(r.get("Counter_Value", 0)))
    return acc


res = {}
for sub, cs in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    for kern, d in pmc(sub, cs).items():
        short = kern.split("(")[0][:90]
        for c, vals in d.items():
            res.setdefault(short, {})[c] = {"n": len(vals), "mean": sum(vals) / len(vals), "max": max(vals)}
summary["pmc"] = res
# the bench line of the traced run names the dominant kernel and the launch geometry it ran with: a PMC record is only
# valid for that geometry (bench.py load_traffic compares it with gcmf_last_kernel_geometry of the run it annotates)
for log in ("bench_trace.log", "bench_fetch.log"):
    try:
        line = [l for l in open(os.path.join(out, log)) if l.startswith("{")][-1]
        rf = json.loads(line)["roofline"]
        summary["bench_kernel"], summary["bench_geometry"] = rf["kernel"], rf["geometry"]
        lines.append(f"== bench.py ({log}) ==\n{rf['kernel']}  geometry {rf['geometry']}  avg launch {rf['avg_launch_ms'] * 1e3:.1f} us (HIP events)")
        break
    except Exception:
        continue
lines.append("== PMC (per dispatch means) ==")
for k, d in res.items():
    lines.append(k)
    for c, v in d.items():
        lines.append(f"    {c}: n={v['n']} mean={v['mean']:.6g} max={v['max']:.6g}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("\n".join(lines))
