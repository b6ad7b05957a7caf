#!/bin/bash
# kernel trace of the train step at the recipe's batch size (24)
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/profs
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profs -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-batch 24 --train-steps 20 ) > $R/gpurun_out/rocprof_s.log 2>&1
cd $R
tail -c 300 gpurun_out/rocprof_s.log
