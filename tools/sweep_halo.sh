#!/bin/bash
# ghost depth x batch x exchange x overlap for the emulated 8-way bound (configs 3 and 4)
out=gpurun_out/halo_sweep
mkdir -p $out
for cfg in ${CFGS:-3 4}; do
  for ov in ${OVERLAPS:-0 1}; do
    for halo in ${HALOS:-9 18 27 36 0}; do
      timeout 600 python tools/measure_batched_scaling.py --config $cfg --batches ${BATCHES:-1,8,16} --halo $halo --overlap $ov > $out/cfg${cfg}_halo${halo}_ov${ov}.txt 2>&1
    done
  done
done
grep -H "batch" $out/*_ov*.txt | sed 's/whole grid//; s/slab, exchange//g; s/gpurun_out.halo_sweep.//'
