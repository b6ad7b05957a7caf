#!/bin/bash
# end-of-round-4 collection on ONE lease, in this order: HBM counter passes of the final tree (eval forward, train step: bench.py then
# reads the refreshed profiles/pmc_*latest.json of THIS tree), the whole GPU suite, the default bench line, kernel-trace stats of the
# eval-only command (with and without stream lanes), a kernel trace of the train step, the per-unit table of the eval forward.
# Everything lands under gpurun_out/r4/ (merged back); profiles/r4_* are written from it (tools/r4_tables.py, here and again locally).
O=$PWD/gpurun_out/r4; mkdir -p $O; R=$PWD
bash tools/gpu_pmc_hbm.sh r4z r4 > /dev/null 2>&1; tail -3 gpurun_out/r4z/pmc_hbm.txt
bash tools/gpu_pmc_train.sh r4z r4 > /dev/null 2>&1; grep "^##" gpurun_out/r4z/pmc_train.txt
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
( timeout 300 python tools/unit_table.py --json $O/unit_table.json ) > $O/unit_table.txt 2>&1; tail -3 $O/unit_table.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_eval $O/trace_eval_nolanes $O/trace_train
EV="--train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval -o t -- python $R/bench.py $EV ) > $O/trace_eval.json 2> $O/trace_eval.err
( CSN_OVERLAP=0 CSN_SLICE_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval_nolanes -o t -- python $R/bench.py $EV ) > $O/trace_eval_nolanes.json 2> $O/trace_eval_nolanes.err
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_train -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 --train-net x2 ) > $O/trace_train.log 2>&1
cd $R
python tools/stats_md.py $O/trace_eval $O/kernel_stats_eval.md "rocprofv3 --kernel-trace --stats -- python bench.py $EV (r4, MI355X)" $O/trace_eval.json
python tools/stats_md.py $O/trace_eval_nolanes $O/kernel_stats_eval_nolanes.md "same command with CSN_OVERLAP=0 CSN_SLICE_LANES=0 (one stream, whole batch: no launch overlaps another)" $O/trace_eval_nolanes.json
python tools/train_step_breakdown.py $(find $O/trace_train -name "*kernel_trace.csv" | head -1) 0 > $O/train_step_kernels.md 2>&1
cp $(find $O/trace_train -name "*kernel_trace.csv" | head -1) $O/train_kernel_trace.csv 2>/dev/null; gzip -f $O/train_kernel_trace.csv
rm -rf $O/trace_train $O/trace_eval/*/*.db $O/trace_eval_nolanes/*/*.db
head -12 $O/kernel_stats_eval_nolanes.md; grep -A12 "## bf16" $O/train_step_kernels.md
# the whole GPU suite on the same tree, last (the collection above does not depend on it)
[ -n "$SKIP_PYTEST" ] || { ( timeout 600 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 ) > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log; }
# (SKIP_PYTEST=1: the re-collection after a change that the CPU suite covers; then only the train / path tests that touch it)
[ -z "$SKIP_PYTEST" ] || { ( timeout 200 python -m pytest tests -m gpu -q -x -k "(units_local and bf16 and 96) or train_step_bf16" 2>&1 | tail -3 ) > $O/pytest_gpu_recheck.log; tail -2 $O/pytest_gpu_recheck.log; }
