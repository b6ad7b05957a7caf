#!/bin/bash
# round-3 checkpoint on the MI355X box: full GPU test suite, default bench line, kernel-trace stats of the eval-only command
# (default and without stream lanes / slices) -> gpurun_out/r3/
O=$PWD/gpurun_out/r3; mkdir -p $O
R=$PWD
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_eval $O/trace_eval_nolanes
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval -o t -- python $R/bench.py --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 ) > $O/trace_eval.json 2> $O/trace_eval.err
( CSN_OVERLAP=0 CSN_SLICE_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval_nolanes -o t -- python $R/bench.py --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 ) > $O/trace_eval_nolanes.json 2> $O/trace_eval_nolanes.err
cd $R
python tools/stats_md.py $O/trace_eval $O/kernel_stats_eval.md "rocprofv3 --kernel-trace --stats -- python bench.py --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 (r3, MI355X)" $O/trace_eval.json
python tools/stats_md.py $O/trace_eval_nolanes $O/kernel_stats_eval_nolanes.md "same command with CSN_OVERLAP=0 CSN_SLICE_LANES=0 (one stream, whole batch: no launch overlaps another)" $O/trace_eval_nolanes.json
