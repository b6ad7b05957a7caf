#!/bin/bash
# end-of-round collection on one lease: counter passes (eval forward, train step) first -- bench.py then reads the refreshed
# profiles/pmc_*latest.json of THIS tree --, full GPU suite + bench + eval-only traces (tools/gpu_round3.sh), train-step trace
bash tools/gpu_pmc_hbm.sh r3z r3 > /dev/null 2>&1; tail -3 gpurun_out/r3z/pmc_hbm.txt
bash tools/gpu_pmc_train.sh r3z r3 > /dev/null 2>&1; grep "^##" gpurun_out/r3z/pmc_train.txt
bash tools/gpu_round3.sh
O=$PWD/gpurun_out/r3z; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 ) > $O/trace.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/train_step_kernels.md 2>&1
rm -rf $O/trace
head -12 $O/train_step_kernels.md
