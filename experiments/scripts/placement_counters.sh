#!/bin/bash
# experiments/scripts/placement_counters.sh [NPLANS]  (through gpurun): per-plan launch time and per-plan counter means of k_ringc<double, 2, 8, false>
set -u
N=${1:-8}
OUT=$PWD/gpurun_out/placement; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
i=0
for G in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
         "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCC_EA0_WRREQ_STALL_sum TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" \
         "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  # (every pass under its own timeout: a counter name this rocprofv3 does not know makes it abort and then sit in its finaliser --
  # round 4 lost 40 GPU-minutes to exactly that)
  timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc $G -d "$OUT/g$i" -o pmc -- python3 $REPO/experiments/scripts/placement_counters.py $N 2 > "$OUT/g$i.log" 2>&1
  i=$((i+1))
done
cd $REPO
python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys, re, collections
out, nplans = sys.argv[1], int(sys.argv[2])
for g in sorted(glob.glob(f"{out}/g[0-9]")):
    times = [float(m.group(1)) for m in re.finditer(r"PLAN \d+: ([0-9.]+) us", open(g + ".log").read())]
    rows = []
    for f in glob.glob(f"{g}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "k_ringc<double, 2, 8, false>" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    per = len(ids) // max(nplans, 1)
    print(f"== {g.split('/')[-1]}: {len(ids)} dispatches of k_ringc<double, 2, 8, false>, {per} per plan; launch time by HIP events (us) and counter means per plan")
    names = sorted({k for v in by.values() for k in v})
    print("plan  us/launch  " + "  ".join(names))
    for p in range(nplans):
        chunk = ids[p * per:(p + 1) * per]
        if not chunk:
            continue
        means = [sum(by[i].get(nm, 0.0) for i in chunk) / len(chunk) for nm in names]
        print(f"{p:4d}  {times[p] if p < len(times) else float('nan'):9.1f}  " + "  ".join(f"{m:.4g}" for m in means))
PY
