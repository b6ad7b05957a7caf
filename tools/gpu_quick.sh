#!/bin/bash
# quick GPU iteration: gpu parity tests (short), bench, kernel trace
mkdir -p gpurun_out
R=$PWD
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_q.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/profq
( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profq -o q -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-iters 1 ) > $R/gpurun_out/rocprof_q.log 2>&1
cd $R
tail -3 gpurun_out/pytest_gpu.log
tail -c 1500 gpurun_out/bench_q.log
