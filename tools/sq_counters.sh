#!/bin/bash
# usage: tools/sq_counters.sh <tag> <config>   (run through gpurun)
# SQ / GRBM counter passes of the default bench workload of one BASELINE config -> gpurun_out/pmc_<tag>/summary.txt
# (copy into profiles/<round>/).  Counters only with --kernel-trace, one group per rocprofv3 run.
set -u
TAG=$1; CFG=$2
ARGS="--steps 2 --warmup 1 --no-cpu --no-extra --config $CFG"
mkdir -p gpurun_out/pmc_$TAG
bash tools/pmc_pass.sh "$TAG" "$ARGS" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "GRBM_GUI_ACTIVE" > gpurun_out/pmc_$TAG/summary.txt 2>&1
echo "rocprofv3 --pmc passes (tools/sq_counters.sh $TAG $CFG: bench.py $ARGS, one counter group per run), per-dispatch means" | cat - gpurun_out/pmc_$TAG/summary.txt > gpurun_out/pmc_$TAG/s.tmp && mv gpurun_out/pmc_$TAG/s.tmp gpurun_out/pmc_$TAG/summary.txt
