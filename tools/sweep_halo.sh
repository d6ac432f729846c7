#!/bin/bash
# ghost depth x batch x exchange for the emulated 8-way bound (configs 3 and 4)
out=gpurun_out/halo_sweep
mkdir -p $out
for cfg in 3 4; do
  for halo in 9 18 27 36 0; do
    timeout 600 python tools/measure_batched_scaling.py --config $cfg --batches 1,8,16 --halo $halo > $out/cfg${cfg}_halo${halo}.txt 2>&1
  done
done
tail -n 5 $out/*.txt
