#!/bin/bash
mkdir -p gpurun_out/pmc2
R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ( timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc2/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 ) > $R/gpurun_out/pmc2/$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run c SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R; ls gpurun_out/pmc2
