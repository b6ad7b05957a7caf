#!/bin/bash
# SQ counter passes over one train step of each storage mode (counters only, separate runs) -> gpurun_out/$1/sq_<pass>/ ; summary by
# kernel family: tools/pmc_sq_summary.py
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 1 --train-net x2"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ( timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $out/sq_$name -o $name -- $CMD ) > $out/sq_$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
cd $R
python tools/pmc_sq_summary.py $out | tee $out/sq_summary.txt | head -60
