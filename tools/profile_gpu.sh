#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM PMC passes for the default bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
CFG=${2:-3}
OUT=$PWD/gpurun_out/prof_${TAG}_cfg${CFG}
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
# 1. per-kernel timing (no counters)
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-replan --config $CFG > "$OUT/bench_trace.log" 2>&1
# 2. HBM read / write bytes, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-extra --config $CFG > "$OUT/bench_fetch.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-extra --config $CFG > "$OUT/bench_write.log" 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-extra --config $CFG > "$OUT/bench_l2.log" 2>&1
cd $REPO
find "$OUT" -name '*.csv' | head -50
python3 tools/summarize_profile.py "$OUT" "$TAG" "$CFG"
