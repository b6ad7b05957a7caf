#!/bin/bash
# SQ counter passes of the eval forward (separate runs, counters only) -> gpurun_out/$1/pmc_<pass>/ ; per-dispatch summary
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ( timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o $name -- python $R/tools/unit_table.py --steps 2 --iters 1 --quiet ) > $out/pmc_$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
cd $R
python tools/pmc_forward.py $out > $out/pmc_forward.txt 2>&1; cat $out/pmc_forward.txt
python tools/trace_forward.py $out 2>/dev/null | tail -3
