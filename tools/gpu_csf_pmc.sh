#!/bin/bash
# One PMC pass over the CSF head (separate from any trace): matrix-pipe busy cycles, wave-cycle split, LDS conflicts.
mkdir -p gpurun_out/csf
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/csf/pmc
( timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/csf/pmc -o pmc -- python $R/bench.py --steps 2 --warmup 1 --train-steps 0 --no-cpu-baseline --profile-iters 1 --csf-steps 2 ) > $R/gpurun_out/csf/pmc.log 2>&1
cd $R
tail -3 gpurun_out/csf/pmc.log | cut -c1-300
ls gpurun_out/csf/pmc
