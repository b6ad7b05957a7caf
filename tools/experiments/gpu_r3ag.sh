#!/bin/bash
# eval-only kernel traces of the final tree (the two profiles/r3_kernel_stats_eval*.md), nothing else
O=$PWD/gpurun_out/r3; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_eval $O/trace_eval_nolanes
( CSN_OVERLAP=0 CSN_SLICE_LANES=0 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval_nolanes -o t -- python $R/bench.py --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 ) > $O/trace_eval_nolanes.json 2> $O/trace_eval_nolanes.err
( timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval -o t -- python $R/bench.py --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 ) > $O/trace_eval.json 2> $O/trace_eval.err
cd $R
python tools/stats_md.py $O/trace_eval_nolanes $O/kernel_stats_eval_nolanes.md "same command with CSN_OVERLAP=0 CSN_SLICE_LANES=0 (one stream, whole batch: no launch overlaps another)" $O/trace_eval_nolanes.json
python tools/stats_md.py $O/trace_eval $O/kernel_stats_eval.md "rocprofv3 --kernel-trace --stats -- python bench.py --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 (r3, MI355X)" $O/trace_eval.json
sed -n 3,12p $O/kernel_stats_eval_nolanes.md
