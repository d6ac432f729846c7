#!/bin/bash
set -u
OUT=$PWD/gpurun_out/prof_cgrid
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
ARGS="--steps 1 --warmup 0 --no-cpu --no-extra --config 5 --nlev 12"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $REPO/bench.py $ARGS > "$OUT/b0.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 $REPO/bench.py $ARGS > "$OUT/b1.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python3 $REPO/bench.py $ARGS > "$OUT/b2.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq" -o pmc -- python3 $REPO/bench.py $ARGS > "$OUT/b3.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d "$OUT/pmc_sq2" -o pmc -- python3 $REPO/bench.py $ARGS > "$OUT/b4.log" 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
out="gpurun_out/prof_cgrid"
for sub in ("pmc_fetch","pmc_write","pmc_sq","pmc_sq2"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cgrid" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(sub, k, "n",len(v), "mean", sum(v)/len(v))
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
