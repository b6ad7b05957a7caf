#!/bin/bash
# SQ / SQC counter passes of the eval forward (separate runs, counters only) -> gpurun_out/$1/pmc_<pass>/
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ( timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 0 --csf-batch 0 --event-steps 0 ) > $out/pmc_$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE
run c SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
cd $R
python tools/parse_pmc_ilb.py $out > $out/pmc_summary.txt 2>&1; cat $out/pmc_summary.txt
