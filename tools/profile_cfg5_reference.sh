#!/bin/bash
# rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of config 5 under evaluation="reference" (k_cgrid_ringf) -> gpurun_out/prof_cfg5_reference/
set -u
OUT=$PWD/gpurun_out/prof_cfg5_reference; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $REPO/tools/run_cfg5_reference.py 3 > "$OUT/run_trace.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 $REPO/tools/run_cfg5_reference.py 1 > "$OUT/run_fetch.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python3 $REPO/tools/run_cfg5_reference.py 1 > "$OUT/run_write.log" 2>&1
cd $REPO
python3 - <<'PY' > gpurun_out/prof_cfg5_reference/summary.txt
import csv, glob, collections
out = "gpurun_out/prof_cfg5_reference"
print("config 5, Filter(evaluation=\"reference\"): rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (tools/profile_cfg5_reference.sh)")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("stats:", r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
for sub in ("pmc_fetch", "pmc_write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cgrid" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(sub, k[0], k[1], "n", len(v), "mean KB", sum(v) / len(v))
PY
cat gpurun_out/prof_cfg5_reference/summary.txt
tail -2 $OUT/run_trace.log
