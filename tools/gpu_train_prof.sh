#!/bin/bash
# train-step bench + kernel trace of the same command
mkdir -p gpurun_out
R=$PWD
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 10 --train-batch 256 ) > gpurun_out/bench_t.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/proft
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/proft -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 5 --train-batch 256 ) > $R/gpurun_out/rocprof_t.log 2>&1
cd $R
tail -c 900 gpurun_out/bench_t.log
find gpurun_out/proft -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -d, -f1-4,6 {} | head -40"
