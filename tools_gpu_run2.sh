#!/bin/bash
mkdir -p gpurun_out/pmc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
run() { # name, counters...
  name=$1; shift
  ( timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 ) > $R/gpurun_out/pmc/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $R
ls -R gpurun_out/pmc | head -40
