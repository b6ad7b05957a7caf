#!/bin/bash
# CSF+Res2Net on the MI355X box: GPU parity tests, then the config-5 data point of bench.py under a kernel trace.
mkdir -p gpurun_out/csf
R=$PWD
( timeout 600 python -m pytest tests/test_gpu_csf.py -x -q -s ) > gpurun_out/csf/tests.log 2>&1
tail -25 gpurun_out/csf/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/csf/trace
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/csf/trace -o trace -- python $R/bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --profile-iters 1 ) > $R/gpurun_out/csf/bench.log 2>&1
cd $R
tail -c 1500 gpurun_out/csf/bench.log
f=$(find gpurun_out/csf/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200
