#!/bin/bash
# Kernel trace of the train step (batch 256, both storage modes) -> gpurun_out/<out>/train_step_kernels[_<tag>].md (per kernel family:
# launches, ms; tools/train_step_breakdown.py).  usage (inside a lease): bash tools/gpu_train_trace.sh <out> [tag] [net: x2 | unpruned]
O=$PWD/gpurun_out/$1; mkdir -p $O; R=$PWD; TAG=${2:-new}; NET=${3:-x2}
cd /tmp && export TMPDIR=/tmp
TR="--steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3"
rm -rf $O/trace_$TAG
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -o t -- python $R/bench.py $TR --train-net $NET ) > $O/trace_$TAG.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace_$TAG -name "*kernel_trace.csv" | head -1) 0 $O/train_step_launches_$TAG.txt > $O/train_step_kernels_$TAG.md 2>&1
rm -rf $O/trace_$TAG
grep -A16 "## bf16" $O/train_step_kernels_$TAG.md | head -24
