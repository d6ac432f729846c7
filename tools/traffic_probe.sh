#!/bin/bash
# usage (through gpurun): tools/traffic_probe.sh  -> gpurun_out/traffic_probe/summary.txt
# Builds experiments/cgrid_traffic_probe and runs it under rocprofv3 --pmc FETCH_SIZE (and the TCC request counters) for both access
# shapes: what FETCH_SIZE tallies per byte actually requested once from memory.
set -u
REPO=$PWD; OUT=$REPO/gpurun_out/traffic_probe; mkdir -p $OUT; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/traffic_probe $REPO/experiments/cgrid_traffic_probe/probe.hip 2> $OUT/build.log || { cat $OUT/build.log; exit 1; }
cd /tmp
for MODE in 0 1; do
  /tmp/traffic_probe $MODE 8 > $OUT/run_mode$MODE.log 2>&1
  for G in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $G | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $G -d $OUT/m${MODE}_$T -o pmc -- /tmp/traffic_probe $MODE 8 > $OUT/m${MODE}_$T.log 2>&1
  done
done
cd $REPO
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections, re
out = sys.argv[1]
for mode in (0, 1):
    known = int(re.search(r"KNOWN_BYTES (\d+)", open(f"{out}/run_mode{mode}.log").read()).group(1))
    print(open(f"{out}/run_mode{mode}.log").read().strip())
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/m{mode}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_probe" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        m = sum(v) / len(v)
        extra = ""
        if k == "FETCH_SIZE":
            extra = f"   -> x 1024 B = {m * 1024 / 1e6:.1f} MB = {m * 1024 / known:.4f} of the {known / 1e6:.1f} MB requested (bytes per unit of FETCH_SIZE: {known / m:.1f})"
        print(f"  mode {mode} {k}: n={len(v)} mean={m:.6g}{extra}")
PY
