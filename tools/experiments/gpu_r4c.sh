#!/bin/bash
# round 4, lease C: train-step tests (deferred launches under graph replay, prefetching depthwise kernels), A/B of the prefetch
# (variant library without it), kernel trace of the train step
O=$PWD/gpurun_out/${1:-r4c}; mkdir -p $O; R=$PWD
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_replay.py -m gpu -q -x -k "units_local or train_step_bf16 or train_golden or overlap2 or loss_goes_down or train_replay or batch256" --durations=6 2>&1 | tail -22 ) > $O/pytest_sel.log; tail -12 $O/pytest_sel.log
B="python bench.py --steps 5 --warmup 2 --train-steps 6 --csf-batch 0 --no-cpu-baseline --no-latency-b1 --event-steps 0 --profile-iters 1"
for v in main nopf main nopf; do
  if [ $v = main ]; then unset SOD100K_HIP_LIB; else export SOD100K_HIP_LIB=$R/gpurun_variants/lib_$v.so; fi
  ( timeout 300 $B ) > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("$v", "eval", d["value"], "train fp32", d["train_step"]["ms_per_step"], "bf16", d["train_step_bf16"]["ms_per_step"])
PY
done
unset SOD100K_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 ) > $O/trace.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/train_step_kernels.md 2>&1
cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv 2>/dev/null; gzip -f $O/kernel_trace.csv
rm -rf $O/trace
grep -A45 "## bf16" $O/train_step_kernels.md | head -50
