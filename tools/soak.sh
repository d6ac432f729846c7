#!/bin/bash
# Soak of the GPU suite: the whole `-m gpu` suite N times back to back and the two-rank p2p bench of config 4 (the
# test that failed on the driver's box in round 3) M times; one summary line per run.  Run through gpurun; copy the summary to profiles/.
N=${1:-5}; M=${2:-20}
OUT=$PWD/gpurun_out/soak; mkdir -p "$OUT"
: > "$OUT/summary.txt"
for i in $(seq 1 $N); do
  python -m pytest tests -x -q -m gpu -p no:cacheprovider > "$OUT/suite_$i.log" 2>&1
  echo "suite run $i: rc=$? $(grep -E 'passed|failed' "$OUT/suite_$i.log" | tail -1)" >> "$OUT/summary.txt"
done
for i in $(seq 1 $M); do
  s=$(date +%s%N)
  python -m pytest "tests/test_gpu_bench_cli.py::test_self_launch_two_ranks_sharing_the_gpu" "tests/test_gpu_bench_cli.py::test_two_ranks_with_skewed_clocks_run_matched_collectives" -x -q -p no:cacheprovider -k "4-extra4 or 4-p2p" > "$OUT/p2p_$i.log" 2>&1
  rc=$?
  e=$(date +%s%N)
  echo "config-4 p2p two-rank bench run $i: rc=$rc $(grep -E 'passed|failed' "$OUT/p2p_$i.log" | tail -1) wall $(( (e - s) / 1000000 )) ms" >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
