#!/bin/bash
# End-of-round collection on ONE lease (round given as $1, default r6), in this order: HBM counter passes of the final tree (eval forward,
# train step: bench.py then reads the refreshed profiles/pmc_*latest.json of THIS tree), the default bench line, the per-unit table,
# kernel-trace stats of the eval-only command (with and without stream lanes), a kernel trace of the train step of the shipped net and
# of the UN-PRUNED net, SQ / TCP counters of the eval forward's largest launches, the whole GPU suite LAST.
# Everything lands under gpurun_out/<round>/ (merged back); profiles/<round>_* are written from it (tools/round_tables.py, here and locally).
RND=${1:-r6}
O=$PWD/gpurun_out/$RND; mkdir -p $O; R=$PWD
bash tools/gpu_pmc_hbm.sh ${RND}z $RND > /dev/null 2>&1; tail -3 gpurun_out/${RND}z/pmc_hbm.txt
bash tools/gpu_pmc_train.sh ${RND}z $RND > /dev/null 2>&1; grep "^##" gpurun_out/${RND}z/pmc_train.txt
bash tools/gpu_pmc_train.sh ${RND}y $RND unpruned > /dev/null 2>&1; grep "^##" gpurun_out/${RND}y/pmc_train.txt
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; python tools/bench_line.py $O/bench.json
( timeout 300 python tools/unit_table.py --json $O/unit_table.json ) > $O/unit_table.txt 2>&1; tail -1 $O/unit_table.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_eval $O/trace_eval_nolanes $O/trace_train $O/trace_unpruned
EV="--train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval -o t -- python $R/bench.py $EV ) > $O/trace_eval.json 2> $O/trace_eval.err
( CSN_OVERLAP=0 CSN_SLICE_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval_nolanes -o t -- python $R/bench.py $EV ) > $O/trace_eval_nolanes.json 2> $O/trace_eval_nolanes.err
TR="--steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3"
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_train -o t -- python $R/bench.py $TR --train-net x2 ) > $O/trace_train.log 2>&1
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_unpruned -o t -- python $R/bench.py $TR --train-net unpruned ) > $O/trace_unpruned.log 2>&1
cd $R
python tools/stats_md.py $O/trace_eval $O/kernel_stats_eval.md "rocprofv3 --kernel-trace --stats -- python bench.py $EV ($RND, MI355X)" $O/trace_eval.json
python tools/stats_md.py $O/trace_eval_nolanes $O/kernel_stats_eval_nolanes.md "same command with CSN_OVERLAP=0 CSN_SLICE_LANES=0 (one stream, whole batch: no launch overlaps another -- the launches bench.py's per-kernel roofline is taken on)" $O/trace_eval_nolanes.json
python tools/train_step_breakdown.py $(find $O/trace_train -name "*kernel_trace.csv" | head -1) 0 > $O/train_step_kernels.md 2>&1
python tools/train_step_breakdown.py $(find $O/trace_unpruned -name "*kernel_trace.csv" | head -1) 0 > $O/train_step_kernels_unpruned.md 2>&1
cp $(find $O/trace_train -name "*kernel_trace.csv" | head -1) $O/train_kernel_trace.csv 2>/dev/null; gzip -f $O/train_kernel_trace.csv
rm -rf $O/trace_train $O/trace_unpruned $O/trace_eval/*/*.db $O/trace_eval_nolanes/*/*.db
head -12 $O/kernel_stats_eval_nolanes.md; grep -A12 "## bf16" $O/train_step_kernels.md
bash tools/gpu_pmc_eval.sh $RND > /dev/null 2>&1; grep -c "per wave" $O/pmc_eval.txt
python tools/round_tables.py $RND $O profiles/${RND}_pmc_train.json | tail -1
# the whole GPU suite on the same tree, last (the collection above does not depend on it)
[ -n "$SKIP_PYTEST" ] || { ( timeout 900 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 ) > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log; }
