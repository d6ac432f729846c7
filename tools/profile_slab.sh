#!/bin/bash
# kernel trace of what ONE rank of an N-GPU slab run computes (tools/measure_slab_compute.py): profile_slab.sh <config> <N>
set -u
CFG=${1:-3}; N=${2:-8}
OUT=$PWD/gpurun_out/prof_slab_cfg${CFG}_n${N}
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $REPO/tools/measure_slab_compute.py $CFG $N > "$OUT/run.log" 2>&1
cd $REPO
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r["Name"].split("(")[0][:70], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
grep strong "$OUT/run.log"
