#!/bin/bash
# One fuzz campaign over the GPU path against the oracle (through gpurun): every fuzzer in tools/ with fresh seeds; anything that prints
# FAIL / NANPATTERN / EXC / Traceback is a finding.  Output: gpurun_out/fuzz/summary.txt (copy to profiles/<round>/ if it is to be cited).
OUT=$PWD/gpurun_out/fuzz; mkdir -p "$OUT"; S=${1:-400}
( for seed in $S $((S+1)) $((S+2)); do timeout 600 python tools/fuzz_gpu.py $seed 250; done
  timeout 600 python tools/fuzz_gpu.py $((S+3)) 150 --tune
  timeout 600 python tools/fuzz_gpu.py $((S+20)) 60 --mid
  timeout 600 python tools/fuzz_gpu.py $((S+21)) 40 --mid --tune
  timeout 600 python tools/fuzz_gpu.py $((S+4)) 120 --cgrid
  timeout 600 python tools/fuzz_gpu.py $((S+5)) 120 --bgrid
  timeout 600 python tools/fuzz_gpu.py $((S+6)) 300 --tiny
  GCMF_RESIDENT=1 timeout 600 python tools/fuzz_gpu.py $((S+7)) 250
  GCMF_RESIDENT=0 timeout 600 python tools/fuzz_gpu.py $((S+8)) 150
  timeout 600 python tools/fuzz_gpu.py $((S+30)) 200 --eval backward
  timeout 600 python tools/fuzz_gpu.py $((S+31)) 200 --eval reference
  timeout 600 python tools/fuzz_gpu.py $((S+32)) 80 --eval backward --bgrid
  timeout 600 python tools/fuzz_gpu.py $((S+33)) 120 --eval reference --cgrid
  timeout 600 python tools/fuzz_nsteps.py
  timeout 600 python tools/fuzz_inputs.py $((S+9))
  for w in 2 3; do for ex in auto p2p; do timeout 900 python tools/fuzz_slabs.py $((S+10+w)) 30 $w $ex; done; done
) > "$OUT/all.log" 2>&1
grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)|amdgpu.ids|UserWarning|warnings.warn|^  flt|^  fv" "$OUT/all.log" > "$OUT/summary.txt"
echo "findings: $(grep -cE 'FAIL|NANPATTERN|EXC|Traceback|Error' "$OUT/summary.txt")" >> "$OUT/summary.txt"
tail -60 "$OUT/summary.txt"
