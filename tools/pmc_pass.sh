#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<bench args>" "<counter group 1>" "<counter group 2>" ...
# One rocprofv3 --pmc pass per counter group; prints per-kernel means.  Run through gpurun.
set -u
TAG=$1; ARGS=$2; shift 2
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
i=0
for G in "$@"; do
  timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc $G -d "$OUT/g$i" -o pmc -- python3 $REPO/bench.py $ARGS > "$OUT/g$i.log" 2>&1
  i=$((i+1))
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "rocclr" in k or "k_pre" in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c}: n={len(v)} mean={sum(v)/len(v):.6g}")
PY
